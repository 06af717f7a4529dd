#!/usr/bin/env python3
"""tools/debug/side_compare.py '<feature dict>': CPU only.  A free-slices stream through the product's PARSER (parse-only hook, KVAZZUP_AMD_PROBE_DUMP) and the
checker's decoder; the per-4x4 side information (prediction mode, motion, QpY, intra mode) of every picture compared."""
import sys, os, ctypes as C, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, parser_probe as PP
from test_gpu_foreign import PLAIN

feature = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
cfg = dict(PLAIN); cfg.update(feature)
w, h, n = int(os.environ.get("W", 416)), int(os.environ.get("H", 240)), int(os.environ.get("N", 6))
g = orc.OracleGen(w, h, seed=int(os.environ.get("SEED", "11")), slices=int(os.environ.get("SLICES", "3")), **cfg)
od = orc.OracleDecoder()
L = orc.lib()
L.orc_dec_debug_side.argtypes = [C.c_void_p] + [C.c_void_p] * 5
dump = tempfile.mktemp()
os.environ["KVAZZUP_AMD_PROBE_DUMP"] = dump
nals, sides = [], []
nb = (w // 4) * (h // 4)
for t in range(n):
    au = g.picture()
    nals += list(orc.split_nals(au))
    od.decode_au(au, t)
    mv = np.zeros((nb, 2), np.int16); ref = np.zeros(nb, np.int8); pm = np.zeros(nb, np.uint8); im = np.zeros(nb, np.uint8); qp = np.zeros(nb, np.int8)
    assert L.orc_dec_debug_side(od.p, mv.ctypes.data, ref.ctypes.data, pm.ctypes.data, im.ctypes.data, qp.ctypes.data) == nb
    sides.append((mv, ref, pm, im, qp))
nals.append(bytes([0, 0, 0, 1, 36 << 1, 1]))
print(PP.probe(nals, 1))
raw = open(dump, "rb").read(); os.unlink(dump)
off = 0
ctb = 1 << g.config["ctb_log2"]
for t in range(n):
    pw4, ph4, w4, h4 = np.frombuffer(raw, np.int32, 4, off); off += 16
    rec = np.frombuffer(raw, np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("ref", "i1"), ("flags", "u1"), ("qp", "i1"), ("slot", "u1")]), pw4 * ph4, off).reshape(ph4, pw4)[:h4, :w4]; off += 8 * pw4 * ph4
    imode = np.frombuffer(raw, np.uint8, pw4 * ph4, off).reshape(ph4, pw4)[:h4, :w4]; off += pw4 * ph4
    mv, ref, pm, im, qp = sides[t]
    mv = mv.reshape(h4, w4, 2); ref = ref.reshape(h4, w4); pm = pm.reshape(h4, w4); im = im.reshape(h4, w4); qp = qp.reshape(h4, w4)
    intra_o = pm == 1                                           # MODE_INTRA
    bad_ref = (rec["ref"] < 0) != intra_o
    inter = ~intra_o & ~bad_ref
    bad_mv = inter & ((rec["mvx"] != mv[:, :, 0]) | (rec["mvy"] != mv[:, :, 1]) | (rec["ref"] != ref))
    bad_qp = rec["qp"] != qp
    bad_im = intra_o & ~bad_ref & (imode != im)
    print("picture %d: intra/inter %d, motion %d, qp %d, intra mode %d blocks differ" % (t, bad_ref.sum(), bad_mv.sum(), bad_qp.sum(), bad_im.sum()))
    for name, bad in (("motion", bad_mv), ("qp", bad_qp), ("intra mode", bad_im), ("intra/inter", bad_ref)):
        if bad.any():
            y, x = np.argwhere(bad)[0]
            print("   first %s difference at 4x4 (%d, %d) = luma (%d, %d), CTB (%d, %d): product mv (%d, %d) ref %d qp %d mode %d | checker mv (%d, %d) ref %d qp %d mode %d pm %d"
                  % (name, x, y, 4 * x, 4 * y, 4 * x // ctb, 4 * y // ctb, rec["mvx"][y, x], rec["mvy"][y, x], rec["ref"][y, x], rec["qp"][y, x], imode[y, x], mv[y, x, 0], mv[y, x, 1], ref[y, x], qp[y, x], im[y, x], pm[y, x]))
if os.environ.get("AT"):
    t, x, y = [int(v) for v in os.environ["AT"].split(",")]
    mv, ref, pm, im, qp = sides[t]
    w4 = w // 4
    for yy in range(y // 4 - 1, y // 4 + 3):
        print("  4x4 row %3d: " % (4 * yy) + " ".join("%s%2d" % ("I" if pm[yy * w4 + xx] == 1 else "p", im[yy * w4 + xx] if pm[yy * w4 + xx] == 1 else ref[yy * w4 + xx]) for xx in range(x // 4 - 2, x // 4 + 10)))
def bits_of(nal, n=12):
    b = nal[6 if nal[2] == 0 else 5:]
    out, z, k = [], 0, 0
    # (emulation prevention bytes out)
    raw = bytearray()
    for x in b[:n + 4]:
        if z >= 2 and x == 3: z = 0; continue
        raw.append(x); z = z + 1 if x == 0 else 0
    return "".join("{:08b}".format(x) for x in raw)
def ue(bits, p):
    z = 0
    while bits[p] == "0": z += 1; p += 1
    return (1 << z) - 1 + (int(bits[p + 1:p + 1 + z], 2) if z else 0), p + 1 + z
if os.environ.get("MAP"):
    want = int(os.environ["MAP"])
    wc, hc = (w + ctb - 1) // ctb, (h + ctb - 1) // ctb
    nbits = max(1, (wc * hc - 1).bit_length())
    dep_en, pic, segs = 0, -1, []
    for nal in nals:
        ty = (nal[4 if nal[2] == 0 else 3] >> 1) & 63
        bits = bits_of(nal)
        if ty == 34:
            _, p = ue(bits, 0); _, p = ue(bits, p); dep_en = bits[p] == "1"
        if ty < 32:
            p = 0; first = bits[p] == "1"; p += 1
            if 16 <= ty <= 23: p += 1
            _, p = ue(bits, p)
            dep, addr = False, 0
            if not first:
                if dep_en: dep = bits[p] == "1"; p += 1
                addr = int(bits[p:p + nbits], 2)
            if first: pic += 1
            if pic == want: segs.append((dep, addr))
    print("dependent_slice_segments_enabled", dep_en, "segments (dependent, address)", segs)
    lab = [" "] * (wc * hc); k = -1
    for i, (dep, a) in enumerate(segs):
        e = segs[i + 1][1] if i + 1 < len(segs) else wc * hc
        if not dep: k += 1
        for q in range(a, e): lab[q] = chr((97 if dep else 65) + k % 26)
    for cy in range(hc): print("   ", "".join(lab[cy * wc:(cy + 1) * wc]))
