"""A long run of uvgComm's default mode (preset veryfast: SAO, subme 2, intra units in P pictures; 1 Mbit/s with rate control v2; owf 6 -- the tokenizer launcher
thread, the side stream for intra pictures) against the checker, every access unit of N pictures.  GPU box only:  python tools/measure/soak_default_mode.py [N]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, orc
from kvazzup_amd.codec import Encoder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
w, h, owf, br = 640, 368, 6, 600000
ge = Encoder(w, h, options=(("preset", "veryfast"), ("qp", 32), ("period", 64), ("me-range", 16), ("owf", owf), ("bitrate", br), ("rc-algorithm", "lambda")), fields={"target_bitrate": br})      # (the field is what uvgComm writes, kvazaarfilter.cpp:223)
oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16, sao=1, subme=2, bitrate=br, rc_bands=4)
oe.set_option("intra-in-p", 1); oe.set_option("rc-delay", owf + 1); oe.set_option("me-source", 1)
got = []
for t in range(N + owf):
    out = ge.encode(orc.synth_frame(0 if (t // 200) % 2 == 0 else 2, 0x5EED0002 + t // 200, w, h, t) if t < N else None)
    if out[0] is not None:
        got.append(out[0])
assert len(got) == N, len(got)
bad = 0
for t in range(N):
    want = oe.encode(orc.synth_frame(0 if (t // 200) % 2 == 0 else 2, 0x5EED0002 + t // 200, w, h, t))
    if got[t] != want:
        bad += 1
        if bad < 4: print("picture", t, "differs:", len(got[t]), "/", len(want), "bytes")
print("%d pictures, %d differ" % (N, bad))
