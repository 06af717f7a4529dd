# encoder alone, host pictures in (no decoder in the process): what the encoder side sustains
import os, sys, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import orc
from kvazzup_amd.pipeline import Pipeline
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
w, h = 1920, 1080
nclip, periods = 128, 10
clip = [orc.synth_frame(0, 0x5EED0002, w, h, t) for t in range(nclip)]
dclip = []
for f in clip:
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), f.nbytes) == 0; assert hip.hipMemcpy(p, f.ctypes.data, f.nbytes, 1) == 0; dclip.append(p.value)
for host in ((True,) if os.environ.get("HOST_ONLY") else (False, True)):
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/OWF": 6, "uvgx/copyThreads": os.environ.get("COPY_THREADS", "4")},
                  custom=(("me-range", 16),) + ((("recon-output", "0"), ("null-input", os.environ.get("NULL_INPUT", "poll"))) if host else (("input-hold", "1"),)), loopback=False, keep_outputs=False)
    for rep in range(3):
        t0 = time.perf_counter()
        n0 = pl.pushed
        for t in range(periods * 64):
            ok = pl.push_host_paced(clip[pl.pushed % nclip], 6, 60000) if host else pl.push_device_paced(dclip[pl.pushed % nclip], 6, 60000)
            assert ok
        pl.flush()
        assert pl.wait(pl.pushed, 60000)
        dt = time.perf_counter() - t0
        print("encoder only, %s rep %d: %.1f frames/s" % ("host" if host else "resident", rep, (pl.pushed - n0) / dt), flush=True)
    print("busy ms/pic:", [round(b / pl.pushed, 4) for b in pl.busy_ms()])
    pl.close()
