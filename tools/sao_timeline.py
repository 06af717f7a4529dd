"""Where k_sao's time goes (KVAZZUP_AMD_INTRA_TRACE=1): 100 MHz stamps per CTU -- start, window in LDS, statistics, offsets + candidates, decision, filtered.
GPU box only:  python tools/sao_timeline.py [w h [pictures]]"""
import os, sys
os.environ["KVAZZUP_AMD_INTRA_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
e = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16), ("sao", "full")))
npic = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for t in range(npic):
    e.encode(synth.frame(synth.MOVING, 0x5EED0002, w, h, t))
wc, hc = (w + 63) // 64, (h + 63) // 64
buf = np.zeros(wc * hc * 72, dtype=np.uint64)
assert e.lib.kvzx_encoder_debug_copy(e.enc, b"trace", buf.ctypes.data, buf.nbytes)
tr = buf[wc * hc * 56:].reshape(-1, 16)[:, :6].astype(np.int64)
c = (tr - tr[:, 0].min()) / 100.0
names = ["start", "window", "statistics", "candidates", "decision", "filtered"]
print("k_sao %dx%d: %d workgroups, span %.1f us, starts spread over %.1f us" % (w, h, len(c), c[:, 5].max(), c[:, 0].max()))
d = np.diff(c, axis=1)
for i in range(5):
    print("  %-10s -> %-10s mean %6.2f us  median %6.2f  max %6.2f" % (names[i], names[i + 1], d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
print("  per workgroup: mean %.2f us, max %.2f" % ((c[:, 5] - c[:, 0]).mean(), (c[:, 5] - c[:, 0]).max()))
