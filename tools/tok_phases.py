"""Where a k_tokenize wave's time goes (KVAZZUP_AMD_INTRA_TRACE=1): luma waves of P pictures that code ONE coding unit with residual; 100 MHz ticks per
phase, summed per CTU.  Needs a library built with the stamps:  touch kvazzup_amd/csrc/enc_kernels.hip; make -C kvazzup_amd/csrc EXTRA=-DKVZ_TOK_PHASES
(and a plain rebuild afterwards).  GPU box only:  python tools/tok_phases.py [w h [pictures]]"""
import os, sys
os.environ["KVAZZUP_AMD_INTRA_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kvazzup_amd import synth
from kvazzup_amd.codec import Encoder

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
e = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16)))
npic = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wc, hc = (w + 63) // 64, (h + 63) // 64
def read():
    buf = np.zeros(wc * hc * 72, dtype=np.uint64)
    assert e.lib.kvzx_encoder_debug_copy(e.enc, b"trace", buf.ctypes.data, buf.nbytes)
    return buf[wc * hc * 56:].reshape(-1, 16).copy()
for t in range(npic):
    e.encode(synth.frame(synth.MOVING, 0x5EED0002, w, h, t))
    if t == 1:
        base = read()                  # (the intra picture's chain leaves stamps in the same words: what the P pictures add is the difference)
for t in range(8):
    e.encode(None)
ph = (read() - base).astype(np.int64).sum(axis=0)
n = max(1, ph[15])
names = ["start -> has work (first CU record)", "tables + 3x3 records -> LDS", "header bins (lane 0)", "digest of the block's levels", "last position, greater1 carry, COUNT pass, offsets",
         "reserve (global atomic)", "EMIT pass + copy to the slot", "table entries out"]
print("%d luma waves with one coding unit and residual (P pictures, %dx%d)" % (ph[15], w, h))
for k in range(8):
    print("  %-52s %6.2f us" % (names[k], ph[k] / n / 100.0))
print("  %-52s %6.2f us" % ("sum", ph[:8].sum() / n / 100.0))
