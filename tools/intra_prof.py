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
names = {1: "publish check", 2: "border waits/copies", 3: "reference samples", 4: "dc sum", 5: "prediction", 6: "residual + forward rows", 7: "forward columns + quantiser",
         8: "dequantised -> inverse columns", 9: "inverse rows + reconstruction", 10: "return", 11: "store to picture"}
print("luma workgroups: blocks per CTU %.1f; shader cycles per block by phase (mean over CTUs):" % nblk.mean())
tot = 0.0
for k in sorted(names):
    v = (prof[:, :, k] / np.maximum(nblk, 1)).mean(); tot += v
    print("  %-34s %8.0f" % (names[k], v))
print("  %-34s %8.0f" % ("sum", tot))
e.close()
