# the receiving side alone (WireAdapter -> OpenHEVCFilter): access units made once, then decoded with pictures left in HBM / copied to host memory
import os, sys, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import orc
from kvazzup_amd.pipeline import Pipeline
w, h, nclip = 1920, 1080, 128
clip = [orc.synth_frame(0, 0x5EED0001, w, h, t) for t in range(nclip)]
enc = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/OWF": 0}, custom=(("me-range", 16),), loopback=False, keep_outputs=True)
aus = []
for t in range(nclip):
    enc.push(clip[t]); enc.flush(); assert enc.wait(t + 1, 60000)
    aus.append(enc.pop_encoded()[0])
enc.close()
print("access units:", len(aus), "bytes/picture", sum(len(a) for a in aus) / len(aus))
for download in (0, 1):
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/OWF": 0, "video/OPENHEVC_threads": 24, "video/OH_parallelization": "Frame", "uvgx/decoderDownload": download},
                  custom=(("me-range", 16),), loopback=True, keep_outputs=False)
    lib = pl.lib
    lib.uvgx_pipeline_push_encoded.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int64, C.c_uint32, C.c_int]
    n = 0
    for rep in range(4):
        t0 = time.perf_counter(); n0 = n
        for k in range(10 * nclip):
            a = aus[k % nclip]
            assert lib.uvgx_pipeline_push_encoded(pl.p, a, len(a), n, 40, 60000); n += 1
        lib.uvgx_pipeline_push_encoded(pl.p, None, 0, 0, 0, 0)
        assert lib.uvgx_pipeline_wait(pl.p, n, 60000)
        dt = time.perf_counter() - t0
        print("decoder only, %s rep %d: %.1f frames/s" % ("host output" if download else "resident output", rep, (n - n0) / dt), flush=True)
    print("busy ms/pic:", [round(b / n, 4) for b in pl.busy_ms()])
    pl.close()
