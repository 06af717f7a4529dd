#!/usr/bin/env python3
"""tools/debug/slices_diff.py: a free-slices stream through the HIP decoder and the checker; for every picture that differs the map of differing coding tree blocks
beside the map of slices (letter = slice, lower case = dependent segment continues it)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import orc
from kvazzup_amd.codec import Decoder
from test_gpu_foreign import PLAIN

def seg_info(nal, nctb_bits, dep_enabled):
    # first_slice_segment_in_pic_flag, [no_output_of_prior_pics], pps id (ue), dependent flag, address
    b = nal[6 if nal[2] == 0 else 5:]
    t = (nal[4 if nal[2] == 0 else 3] >> 1) & 63
    bits = "".join("{:08b}".format(x) for x in b[:8])
    p = 0
    first = bits[p] == "1"; p += 1
    if 16 <= t <= 23: p += 1
    z = 0
    while bits[p] == "0": z += 1; p += 1
    p += 1 + z
    dep, addr = False, 0
    if not first:
        if dep_enabled: dep = bits[p] == "1"; p += 1
        addr = int(bits[p:p + nctb_bits], 2)
    return first, dep, addr

def main():
    feature = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
    cfg = dict(PLAIN); cfg.update(feature)
    w, h, n = 416, 240, 6
    g = orc.OracleGen(w, h, seed=int(os.environ.get("SEED", "11")), slices=int(os.environ.get("SLICES", "3")), **cfg)
    ctb = 1 << g.config["ctb_log2"]
    wc, hc = (w + ctb - 1) // ctb, (h + ctb - 1) // ctb
    nb = max(1, (wc * hc - 1).bit_length())
    od = orc.OracleDecoder(); gd = Decoder()
    refs, got, maps = [], [], []
    for t in range(n):
        au = g.picture()
        dep_enabled = None
        m = []
        for nal in orc.split_nals(au):
            ty = (nal[4 if nal[2] == 0 else 3] >> 1) & 63
            if ty < 32:
                # dependent_slice_segments_enabled_flag: try both readings, keep the one whose addresses increase
                m.append(nal)
        maps.append(m)
        refs += [f["i420"] for f in od.decode_au(au, t)]
        got += gd.decode_au(au, t)
    got += gd.drain()
    print("pictures", len(refs), len(got), "wpp", g.config["wpp"])
    for t in range(n):
        if np.array_equal(got[t]["i420"], refs[t]): continue
        d = (got[t]["i420"][:w * h] != refs[t][:w * h]).reshape(h, w)
        print("picture", t, "differs:", int(d.sum()), "luma samples")
        for dep_enabled in (True, False):
            try:
                segs = [seg_info(x, nb, dep_enabled) for x in maps[t]]
                addrs = [s[2] for s in segs]
                if addrs == sorted(addrs) and len(set(addrs)) == len(addrs): break
            except Exception:
                pass
        print("  segments (first, dependent, address):", segs)
        lab = [" "] * (wc * hc); k = -1
        for i, (first, dep, a) in enumerate(segs):
            e = segs[i + 1][2] if i + 1 < len(segs) else wc * hc
            if not dep: k += 1
            for q in range(a, e): lab[q] = chr((97 if dep else 65) + k % 26)
        for cy in range(hc):
            row = "".join(lab[cy * wc:(cy + 1) * wc])
            bad = "".join("X" if d[cy * ctb:(cy + 1) * ctb, cx * ctb:(cx + 1) * ctb].any() else "." for cx in range(wc))
            print("   ", row, "  ", bad)
        ys, xs = np.nonzero(d)
        for cy in range(hc):
            for cx in range(wc):
                blk = d[cy * ctb:(cy + 1) * ctb, cx * ctb:(cx + 1) * ctb]
                if not blk.any(): continue
                print("  CTB (%d, %d), 4x4 units that differ:" % (cx, cy))
                for y4 in range(0, blk.shape[0], 4):
                    print("     " + "".join("#" if blk[y4:y4 + 4, x4:x4 + 4].any() else "." for x4 in range(0, blk.shape[1], 4)))
        y0, x0 = max(0, ys[0] - 2), max(0, xs[0] - 6)
        G = got[t]["i420"][:w * h].reshape(h, w); R = refs[t][:w * h].reshape(h, w)
        for y in range(y0, min(h, y0 + 8)):
            print("   y=%3d got " % y + " ".join("%3d" % v for v in G[y, x0:x0 + 20]) + "\n        want " + " ".join("%3d" % v for v in R[y, x0:x0 + 20]))
        print("  first differing sample x=%d y=%d; got %d want %d" % (xs[0], ys[0], got[t]["i420"][ys[0] * w + xs[0]], refs[t][ys[0] * w + xs[0]]))
        break

main()
