"""4K with SAO, subme, rate control and owf 3 (the tokenizer launcher thread) against the checker, five pictures.  GPU box only (the checker needs ~1 min)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, orc
from kvazzup_amd.codec import Encoder, Decoder
w, h = 3840, 2160
ge = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16), ("sao", "full"), ("subme", 2), ("bitrate", 8000000), ("rc-algorithm", "lambda"), ("owf", 3)))
oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16, sao=1, subme=2, bitrate=8000000)
oe.set_option("rc-delay", 4)
gd = Decoder()
outs = []
N = 5
for t in range(N + 3):
    fr = orc.synth_frame(0, 0x5EED0002, w, h, t) if t < N else None
    out = ge.encode(fr)
    if out[0] is not None: outs.append(out)
for t in range(N):
    want = oe.encode(orc.synth_frame(0, 0x5EED0002, w, h, t))
    assert outs[t][0] == want, (t, len(outs[t][0]), len(want))
    assert np.array_equal(outs[t][1], oe.recon()), t
    d = gd.decode_au(outs[t][0], t)
    assert len(d) == 1 and np.array_equal(d[0]["i420"], outs[t][1]), t
print("4K sao + rc + subme, owf 3 (tokenizer launcher thread): %d pictures equal to the checker" % N)
