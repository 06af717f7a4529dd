"""First-block times of every luma CTU of the encoder's intra wavefront and which neighbour each CTU followed (GPU box only; see tools/intra_timeline.py)."""
import os, sys
os.environ["KVAZZUP_AMD_INTRA_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
e = Encoder(w, h, options=(("qp", 32), ("period", 1), ("me-range", 16), ("intra-chain", os.environ.get("INTRA_CHAIN", "1"))))
for t in range(3):
    e.encode(synth.frame(synth.MOVING, 0x5EED0002, w, h, t))
wc, hc = (w + 63) // 64, (h + 63) // 64
buf = np.zeros(wc * hc * 72, dtype=np.uint64)
assert e.lib.kvzx_encoder_debug_copy(e.enc, b"trace", buf.ctypes.data, buf.nbytes)
tr = buf[:wc * hc * 24].reshape(hc, wc, 3, 8).astype(np.int64)
us = (tr - tr[..., 0].min()) / 100.0
np.set_printoptions(linewidth=250)
for c in (0,):
    st, fb, en = us[:, :, c, 0], us[:, :, c, 1], us[:, :, c, 2]
    print("first block times, plane", c); print(np.round(fb).astype(int))
    print("start times"); print(np.round(st).astype(int))
    print("end - first block"); print(np.round(en - fb).astype(int))
blk = buf[wc * hc * 56:].reshape(hc, wc, 16)
for (ry, rx) in ((8, 0), (9, 0), (10, 0), (10, 1), (10, 2), (11, 0), (11, 1)):
    ts = [(int(v & ((1 << 60) - 1)) - int(tr[ry, rx, 0, 1])) / 100.0 for v in blk[ry, rx] if v]
    sz = [1 << int(v >> 60) for v in blk[ry, rx] if v]
    print("  CTU (%d, %d) fb %.0f: " % (rx, ry, us[ry, rx, 0, 1]) + " ".join("%.1f[%d]" % (t, n) for t, n in zip(ts, sz)))
e.close()
