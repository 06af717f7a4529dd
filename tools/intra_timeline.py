"""Timeline of the encoder's intra reconstruction wavefront (KVAZZUP_AMD_INTRA_TRACE=1): when every (CTU, plane) workgroup of
k_intra_recon started, ran its first block and finished.  GPU box only:  python tools/intra_timeline.py [w h]"""
import os, sys
os.environ["KVAZZUP_AMD_INTRA_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
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
t0 = tr[..., 0].min()
us = (tr - t0) / 100.0
for c in range(3):
    st, fb, en = us[:, :, c, 0], us[:, :, c, 1], us[:, :, c, 2]
    print("plane %d: kernel span %.0f us; busy per CTU (first block -> end) mean %.1f us, median %.1f; start->first block (waiting) mean %.0f us" %
          (c, en.max() - st.min(), (en - fb).mean(), np.median(en - fb), (fb - st).mean()))
    print("  first-block time along row 0 (every 4th CTU):", np.round(fb[0, ::4]).astype(int).tolist())
    print("  first-block time down column 0:", np.round(fb[:, 0]).astype(int).tolist())
    nb = tr[:, :, c, 7].astype(float)
    p24, p44 = us[:, :, c, 3], us[:, :, c, 4]
    ok = (tr[:, :-1, c, 3] > 0)
    print("  per CTU: blocks %.1f; own first block -> publish(24): mean %.1f us, -> publish(44): mean %.1f us" % (nb.mean(), (p24 - fb)[tr[:, :, c, 3] > 0].mean(), (p44 - fb)[tr[:, :, c, 4] > 0].mean()))
    print("  hand-off: first block after the left neighbour's publish(24): mean %.1f us, median %.1f; after the upper-right neighbour's publish(44): mean %.1f us, median %.1f" %
          ((fb[:, 1:] - p24[:, :-1])[ok].mean(), np.median((fb[:, 1:] - p24[:, :-1])[ok]), (fb[1:, :-1] - p44[:-1, 1:]).mean(), np.median(fb[1:, :-1] - p44[:-1, 1:])))
    print("  lag to the left neighbour's first block, mean %.1f us; to the upper neighbour's, mean %.1f us" %
          ((fb[:, 1:] - fb[:, :-1]).mean(), (fb[1:, :] - fb[:-1, :]).mean()))
# when the first sixteen luma blocks of some CTUs in the middle of the picture were done (us after the CTU's first block started)
blk = buf[wc * hc * 56:].reshape(hc, wc, 16)
print("luma block completion times after the CTU's first block began (block size in brackets), three CTUs of the middle rows:")
for (ry, rx) in ((hc // 2, wc // 2), (hc // 2, wc // 2 + 3), (hc // 2 + 1, wc // 3)):
    ts = [(int(v & ((1 << 60) - 1)) - int(tr[ry, rx, 0, 1])) / 100.0 for v in blk[ry, rx] if v]
    sz = [1 << int(v >> 60) for v in blk[ry, rx] if v]
    print("  CTU (%d, %d): " % (rx, ry) + " ".join("%.1f[%d]" % (t, n) for t, n in zip(ts, sz)))
d = None
lib = e.lib
import ctypes
log2 = np.zeros((((h + 63) // 64) * 8) * (((w + 63) // 64) * 8), dtype=np.uint8)
lib.kvzx_encoder_debug_copy(e.enc, b"cu_log2", log2.ctypes.data, log2.nbytes)
print("CU sizes (8x8 cells): 8x8 %.0f%%, 16x16 %.0f%%, 32x32 %.0f%%" % tuple(100.0 * (log2 == k).mean() for k in (3, 4, 5)))
e.close()
