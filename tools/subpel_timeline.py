"""Where k_subpel's time goes (KVAZZUP_AMD_INTRA_TRACE=1): per searched 32x32 block eight 100 MHz stamps -- start, window staged,
then per step: planes filtered, candidates priced, decision made.  GPU box only:  python tools/subpel_timeline.py [w h subme early]"""
import os, sys
os.environ["KVAZZUP_AMD_INTRA_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
subme = int(sys.argv[3]) if len(sys.argv) > 3 else 4
early = sys.argv[4] if len(sys.argv) > 4 else "on"
e = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16), ("subme", subme), ("me-early-termination", early)))
for t in range(4):
    e.encode(synth.frame(synth.MOVING, 0x5EED0002, w, h, t))
wc, hc = (w + 63) // 64, (h + 63) // 64
buf = np.zeros(wc * hc * 40, dtype=np.uint64)
assert e.lib.kvzx_encoder_debug_copy(e.enc, b"trace", buf.ctypes.data, buf.nbytes)
tr = buf[:wc * hc * 4 * 8].reshape(-1, 8).astype(np.int64)
tr = tr[tr[:, 0] > 0]
tr = tr[tr[:, 0] > tr[:, 0].max() - 100000]       # the last picture's launch only (stamps of earlier pictures stay in the buffer)
us = (tr - tr[:, :1].min()) / 100.0
names = ["start", "window", "planes0", "priced0", "decided0", "planes1", "priced1", "decided1"]
print("searched blocks: %d; kernel span (first start -> last end) %.1f us" % (len(tr), us[:, 7].max() - us[:, 0].min()))
print("block start spread: %.1f us" % (us[:, 0].max() - us[:, 0].min()))
d = np.diff(us, axis=1)
for i in range(7):
    print("  %-9s -> %-9s mean %6.2f us  median %6.2f  max %6.2f" % (names[i], names[i + 1], d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
print("  per block total: mean %.2f us, max %.2f" % ((us[:, 7] - us[:, 0]).mean(), (us[:, 7] - us[:, 0]).max()))
