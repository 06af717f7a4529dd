"""synchronous encoder / decoder at 1080p (owf 0): per-kernel times of P pictures with and without intra-in-p, nothing else on the GPU"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, orc
from kvazzup_amd.codec import Encoder, Decoder
w, h = 1920, 1080
import sys
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
frames = [orc.synth_frame(kind, 0x5EED0001, w, h, t if kind == 0 else 0) for t in range(12)]
for on in (0, 1):
    e = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16), ("intra-in-p", on), ("owf", 0)))
    d = Decoder()
    e.set_profiling(True); d.set_profiling(True)
    for t, f in enumerate(frames):
        au, rec = e.encode(f)
        d.decode_au(au, t)
        if t == 0:
            e.kernel_times(); d.kernel_times()          # drop the IDR picture
    ke, kd = e.kernel_times(), d.kernel_times()
    print("intra-in-p", on, "intra units in last picture", int(np.count_nonzero(e.debug_all()["cu_intra"])))
    print("  enc", {k: round(v[0] / max(1, v[1]) * 1e3, 1) for k, v in ke.items() if v[1]})
    print("  dec", {k: round(v[0] / max(1, v[1]) * 1e3, 1) for k, v in kd.items() if v[1]})
    e.close(); d.close()
