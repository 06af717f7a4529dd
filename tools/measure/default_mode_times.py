"""synchronous encoder / decoder at 1080p with uvgComm's default settings for the size (preset veryfast, 1 Mbit/s, rc lambda): per-kernel times of P pictures"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, orc
from kvazzup_amd.codec import Encoder, Decoder
w, h = 1920, 1080
frames = [orc.synth_frame(0, 0x5EED0001, w, h, t) for t in range(16)]
extra = tuple(tuple(a.split("=", 1)) for a in sys.argv[1:])      # e.g. intra-in-p=2
e = Encoder(w, h, options=(("preset", "veryfast"),) + extra + (("qp", 32), ("period", 64), ("me-range", 16), ("owf", 0), ("bitrate", 1000000), ("rc-algorithm", "lambda")), fields={"target_bitrate": 1000000})
d = Decoder()
e.set_profiling(True); d.set_profiling(True)
for t, f in enumerate(frames):
    au, rec = e.encode(f)
    d.decode_au(au, t)
    if t == 0:
        e.kernel_times(); d.kernel_times()
ke, kd = e.kernel_times(), d.kernel_times()
print("enc", {k: (round(v[0] / max(1, v[1]) * 1e3, 1), int(v[1])) for k, v in ke.items() if v[1]})
print("dec", {k: (round(v[0] / max(1, v[1]) * 1e3, 1), int(v[1])) for k, v in kd.items() if v[1]})
e.close(); d.close()
