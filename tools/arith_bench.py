"""ns per token of the host arithmetic coder's bin loop on a real picture's tokens (CPU only): python tools/arith_bench.py [w h]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, hc
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16)
for t in range(3):
    oe.encode(orc.synth_frame(0, 0x5EED0000, w, h, t))
f, hold = hc.make_frame(oe.debug_all() if hasattr(oe, "debug_all") else oe.debug(), w, h, 32)
L = hc.lib()
L.hc_picture_tokens.restype = C.c_long
L.hc_picture_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
L.hc_bench_play_tokens.restype = C.c_double
L.hc_bench_play_tokens.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int]
tok = np.zeros(4 << 20, dtype=np.uint16)
n = L.hc_picture_tokens(C.byref(f), tok.ctypes.data, len(tok))
print("tokens of picture 2 (P):", n)
for variant, name in ((0, "cabac_play_tokens (hevc_core.h, generic)"), (1, "cabac_play_tokens_host (entropy_host.h)")):
    print("%-45s %.2f ns per token" % (name, L.hc_bench_play_tokens(tok.ctypes.data, n, 30, variant)))
