"""Cycle attribution inside k_intra_recon's per-block chain.  Needs the instrumented build
   make -C kvazzup_amd/csrc BUILD=build_prof TARGET=../libkvazzup_amd_prof.so EXTRA="-DKVZ_PROF [-DKVZ_INTRA_THREADS=64]"
GPU box only:  python tools/intra_prof.py [w h]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["KVAZZUP_AMD_INTRA_TRACE"] = "1"
os.environ.setdefault("KVAZZUP_AMD_LIBRARY", os.path.join(ROOT, "kvazzup_amd", "libkvazzup_amd_prof.so"))
sys.path.insert(0, ROOT)
import numpy as np
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
e = Encoder(w, h, options=(("qp", 32), ("period", 1), ("me-range", 16)))
for t in range(3):
    e.encode(synth.frame(synth.MOVING, 0x5EED0002, w, h, t))
wc, hc = (w + 63) // 64, (h + 63) // 64
buf = np.zeros(wc * hc * 40, dtype=np.uint64)
assert e.lib.kvzx_encoder_debug_copy(e.enc, b"trace", buf.ctypes.data, buf.nbytes)
tr = buf[:wc * hc * 24].reshape(hc, wc, 3, 8).astype(np.int64)
nblk = tr[:, :, 0, 7].astype(float)
prof = buf[wc * hc * 24:].reshape(hc, wc, 16).astype(float)      # g_prof[0..15] of the luma workgroup of every CTU
names = {1: "claim a block", 2: "wait for the units it reads (LDS mask)", 3: "neighbouring CTUs: waits + copies", 5: "reference samples + prediction + residual", 6: "forward rows + columns (matrix cores)",
         7: "quantiser, dequantiser", 8: "inverse transform + reconstruction", 9: "stores (LDS, edge blocks: picture)", 10: "mark / acknowledge / publish"}
w0 = prof[:, :, 14]                                     # blocks wave 0 of the luma workgroup did (the stamps are wave 0's)
print("luma workgroups: blocks per CTU %.1f, of which wave 0 did %.1f; shader cycles per block of wave 0 by phase (mean over CTUs):" % (nblk.mean(), w0.mean()))
tot = 0.0
for k in sorted(names):
    v = (prof[:, :, k] / np.maximum(w0, 1)).mean(); tot += v
    print("  %-44s %8.0f" % (names[k], v))
print("  %-44s %8.0f   (shader clock; 2.4 GHz: %.2f us per block)" % ("sum", tot, tot / 2400.0))
e.close()
