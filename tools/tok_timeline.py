"""Where k_tokenize's and k_tok_compact's time goes (KVAZZUP_AMD_INTRA_TRACE=1): 100 MHz stamps per CTU.  k_tok_compact: start, place in
the dense array known, table staged, order restored, tokens copied.  k_tokenize: start of the CTU's first unit, start and end of its last
unit (the one that closes the CTU).  GPU box only:  python tools/tok_timeline.py [w h [pictures]]   (pictures = 1: the stamps are the intra picture's)"""
import os, sys
os.environ["KVAZZUP_AMD_INTRA_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
e = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16)))
npic = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for t in range(npic):
    e.encode(synth.frame(synth.MOVING, 0x5EED0002, w, h, t))
wc, hc = (w + 63) // 64, (h + 63) // 64
buf = np.zeros(wc * hc * 56, dtype=np.uint64)
assert e.lib.kvzx_encoder_debug_copy(e.enc, b"trace", buf.ctypes.data, buf.nbytes)
tr = buf[wc * hc * 32:wc * hc * 40].reshape(-1, 8).astype(np.int64)
cen = buf[wc * hc * 40:].reshape(-1, 16).astype(np.int64).sum(axis=0)
c = (tr[:, :5] - tr[:, 0].min()) / 100.0
print("k_tok_compact: %d workgroups, span (first start -> last end) %.1f us, starts spread over %.1f us" % (len(c), c[:, 4].max(), c[:, 0].max()))
names = ["start", "placed", "table", "ordered", "copied"]
d = np.diff(c, axis=1)
for i in range(4):
    print("  %-8s -> %-8s mean %6.2f us  median %6.2f  max %6.2f" % (names[i], names[i + 1], d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
print("  per workgroup: mean %.2f us, max %.2f" % ((c[:, 4] - c[:, 0]).mean(), (c[:, 4] - c[:, 0]).max()))
srt = np.sort(c[:, 0])
print("  start times, deciles (us):", " ".join("%.1f" % srt[int(q * (len(srt) - 1) / 10)] for q in range(11)))
k = (tr[:, 5:8] - tr[:, 5:8].min()) / 100.0
print("k_tokenize: span %.1f us; first-unit starts spread over %.1f us, last-unit starts over %.1f us" % (k.max(), k[:, 0].max() - k[:, 0].min(), k[:, 1].max() - k[:, 1].min()))
print("  the closing unit's wave: mean %.2f us, median %.2f, max %.2f" % ((k[:, 2] - k[:, 1]).mean(), np.median(k[:, 2] - k[:, 1]), (k[:, 2] - k[:, 1]).max()))
srt = np.sort(k[:, 1])
print("  last-unit start times, deciles (us):", " ".join("%.1f" % srt[int(q * (len(srt) - 1) / 10)] for q in range(11)))
for name, o, npic in (("P pictures", 0, max(1, npic - 1)), ("the IDR picture", 6, 1)):
    print("k_tokenize waves, %s (per picture):" % name)
    for k, cls in enumerate(("left at once", "header only", "with residual")):
        n, ticks = cen[o + 2 * k], cen[o + 2 * k + 1]
        print("  %-14s %8.0f waves, mean %6.2f us, wave-time %8.1f us" % (cls, n / npic, ticks / max(1, n) / 100.0, ticks / 100.0 / npic))
