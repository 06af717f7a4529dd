"""A long all-intra run through the filter graph (KvazaarFilter' -> WireAdapter -> OpenHEVCFilter', video/Intra = 1: the encoder's and the decoder's intra chains
alternate between two streams each, OWF 6, 16 decoder frame threads) against the checker: every decoded picture of N equals the checker encoder's reconstruction.
GPU box only:  python tools/measure/soak_all_intra.py [N]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, orc
from kvazzup_amd.pipeline import Pipeline
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
w, h = 640, 368
pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 1, "video/OWF": 6, "video/OPENHEVC_threads": 16, "video/OH_parallelization": "Frame"},
              custom=(("me-range", 16),), loopback=True, keep_outputs=True)
frame = lambda t: orc.synth_frame(0 if (t // 100) % 2 == 0 else 2, 0x5EED0003 + t // 100, w, h, t)
got = []
def drain():
    while True:
        d = pl.pop_decoded()
        if d is None: break
        got.append(d["i420"])
for t in range(N):
    assert pl.push_host_paced(frame(t), 6, 120000, borrow=False), pl.stats()       # (a uvgComm filter drops inputs at 10 buffered: the source waits instead)
    if t % 50 == 49: drain()
pl.flush(); assert pl.wait(N, 120000), pl.stats()
drain()
pl.close()
assert len(got) == N, len(got)
oe = orc.OracleEncoder(w, h, qp=32, period=1, me_range=16)
bad = 0
for t in range(N):
    oe.encode(frame(t))
    if not np.array_equal(np.asarray(got[t]).reshape(-1), np.asarray(oe.recon()).reshape(-1)):
        bad += 1
        if bad < 4: print("picture", t, "differs")
print("%d pictures, %d differ" % (N, bad))
