# an all-intra stream (video/Intra = 1), each side alone and both together: which of them sets the rate of the bench's `all_intra` leg
#   python tools/measure/all_intra_sides.py [threads]      (decoder frame threads, default 32)
import os, sys, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import orc
from kvazzup_amd.pipeline import Pipeline
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
w, h, nclip = 1920, 1080, 64
both_only = "--both-only" in sys.argv
args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
threads = int(args_[0]) if args_ else 32
clip = [orc.synth_frame(0, 0x5EED0002, w, h, t) for t in range(nclip)]
dclip = []
for f in clip:
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), f.nbytes) == 0; assert hip.hipMemcpy(p, f.ctypes.data, f.nbytes, 1) == 0; dclip.append(p.value)
ST = {"video/QP": 32, "video/Intra": 1, "video/OWF": 6, "video/OPENHEVC_threads": threads, "video/OH_parallelization": "Frame", "uvgx/decoderDownload": 0}
CU = (("me-range", 16), ("input-hold", "1"))
def encoder_alone():
    pl = Pipeline(w, h, settings=ST, custom=CU, loopback=False, keep_outputs=False)
    for rep in range(3):
        t0 = time.perf_counter(); n0 = pl.pushed
        for t in range(6 * nclip):
            assert pl.push_device_paced(dclip[pl.pushed % nclip], 6, 60000)
        pl.flush(); assert pl.wait(pl.pushed, 60000)
        print("encoder alone rep %d: %.1f frames/s" % (rep, (pl.pushed - n0) / (time.perf_counter() - t0)), flush=True)
    pl.close()


def decoder_alone():
    enc = Pipeline(w, h, settings=dict(ST, **{"video/OWF": 0}), custom=(("me-range", 16),), loopback=False, keep_outputs=True)
    aus = []
    for t in range(nclip):
        enc.push(clip[t]); enc.flush(); assert enc.wait(t + 1, 60000)
        aus.append(enc.pop_encoded()[0])
    enc.close()
    print("access units:", len(aus), "bits/picture %.0f" % (8 * sum(len(a) for a in aus) / len(aus)))
    pl = Pipeline(w, h, settings=ST, custom=(("me-range", 16),), loopback=True, keep_outputs=False)
    lib = pl.lib
    lib.uvgx_pipeline_push_encoded.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int64, C.c_uint32, C.c_int]
    n = 0
    for rep in range(3):
        t0 = time.perf_counter(); n0 = n
        for k in range(6 * nclip):
            a = aus[k % nclip]
            assert lib.uvgx_pipeline_push_encoded(pl.p, a, len(a), n, 40, 60000); n += 1
        lib.uvgx_pipeline_push_encoded(pl.p, None, 0, 0, 0, 0)
        assert lib.uvgx_pipeline_wait(pl.p, n, 60000)
        print("decoder alone rep %d: %.1f frames/s" % (rep, (n - n0) / (time.perf_counter() - t0)), flush=True)
    pl.close()


def both():
    pl = Pipeline(w, h, settings=ST, custom=CU, loopback=True, keep_outputs=False)
    for rep in range(4):
        t0 = time.perf_counter(); n0 = pl.pushed
        for t in range(8 * nclip):
            assert pl.push_device_paced(dclip[pl.pushed % nclip], 6, 60000)
        pl.flush(); assert pl.wait(pl.pushed, 60000)
        print("both rep %d: %.1f frames/s" % (rep, (pl.pushed - n0) / (time.perf_counter() - t0)), flush=True)
    pl.close()


if not both_only:
    encoder_alone()
    decoder_alone()
both()
