#!/usr/bin/env python3
"""The decoder kernels with 1, 2, 4, 8 pictures per launch (csrc/batch.h): n decoder instances in this process are fed the same stream in
lockstep -- the submission layer is held while every decoder queues its next picture, then released, so that every launch carries exactly n
pictures -- and the launches are timed with HIP events on the decoders' stream.  Nothing else runs on the GPU: isolated kernel times, the
figures DESIGN.md section 5 quotes beside the single-picture ones.  Output: one JSON object per size.

  python tools/measure/batch_times.py [1080p|4k] [pictures]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                        # noqa: E402
from kvazzup_amd import synth                             # noqa: E402
from kvazzup_amd.codec import Decoder, Encoder            # noqa: E402

HBM = 8192.0                                              # GB/s, the device's own figure (bench.py reads it from hipDeviceProp)


def alg_bytes(kernel, P):
    return {"k_dec_inter": 3.0 * P, "k_dec_intra": 1.5 * P, "k_dec_deblock": 3.0 * P, "k_dec_sao": 3.0 * P}[kernel]


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "1080p"
    npic = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    w, h = {"1080p": (1920, 1080), "4k": (3840, 2160), "720p": (1280, 720)}[size]
    P = ((w + 63) // 64 * 64) * ((h + 63) // 64 * 64)
    enc = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16)))
    aus = [enc.encode(synth.frame(synth.MOVING, 0x5EED0002, w, h, t))[0] for t in range(npic)]
    enc.close()
    out = {"size": size, "pictures": npic, "by_batch": {}}
    for n in (1, 2, 4, 8):
        decs = [Decoder(threads=4, frame_threads=True, download=False) for _ in range(n)]
        lib = decs[0].lib
        lib.kvzx_batch_hold.argtypes = [C.c_int, C.c_int]
        lib.kvzx_batch_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        lib.kvzx_batch_kernel_name.restype = C.c_char_p
        for d in decs:
            d.set_profiling(1)
        lib.kvzx_batch_stats(0, None, None, None, None, None, None, 1)
        got = 0
        for t in range(npic + 6):
            if n > 1:
                lib.kvzx_batch_hold(0, 1)
            for d in decs:
                if t < npic:
                    got += len(d.decode_au(aus[t], t))
                else:                                        # end-of-sequence NAL units hand out the pictures the frame threads hold back
                    got += 0 if d.decode_nal(bytes([0, 0, 0, 1, 36 << 1, 1])) is None else 1
            if n > 1:
                lib.kvzx_batch_hold(0, 0)
            time.sleep(0.02)                                 # the batch has run before the next one is queued: isolated launches
        assert got == n * npic, (got, n * npic)
        row = {}
        if n == 1:
            for k, (ms, cnt) in decs[0].kernel_times().items():
                if cnt and k.startswith("k_"):
                    row[k] = {"avg_launch_us": round(ms / cnt * 1e3, 2), "pictures_per_launch": 1.0, "us_per_picture": round(ms / cnt * 1e3, 2),
                              "hbm_frac": round(alg_bytes(k, P) / (ms / cnt * 1e-3) / 1e9 / HBM, 5)}
        else:
            ms, ln, fr = (C.c_double * 4)(), (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
            sizes = (C.c_uint64 * 9)()
            nk = lib.kvzx_batch_stats(0, None, None, sizes, ms, ln, fr, 0)
            for i in range(nk):
                if ln[i]:
                    k = lib.kvzx_batch_kernel_name(i).decode()[:-2]
                    us, per = ms[i] / ln[i] * 1e3, fr[i] / ln[i]
                    row[k] = {"avg_launch_us": round(us, 2), "pictures_per_launch": round(per, 2), "us_per_picture": round(us / per, 2),
                              "hbm_frac": round(alg_bytes(k, P) * per / (us * 1e-6) / 1e9 / HBM, 5)}
            row["batches_by_size"] = {str(k): sizes[k] for k in range(1, 9) if sizes[k]}
        out["by_batch"][str(n)] = row
        for d in decs:
            d.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
