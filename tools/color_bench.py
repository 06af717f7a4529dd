#!/usr/bin/env python3
"""Row f1 measurement: I420 -> RGB32 of a picture resident in HBM (the display-side conversion, yuvtorgb32.cpp:29-64).
Prints one JSON line: frames/s, achieved HBM bandwidth (1.5 B read + 4 B written per sample) against the 8 TB/s peak,
and the reference's own AVX2 converter timed on the host beside it when oracle/_ref is present."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    import torch
    from kvazzup_amd import _native as N
    lib = N.load_library()
    w, h = args.width, args.height
    src = torch.randint(0, 256, (w * h * 3 // 2,), dtype=torch.uint8, device="cuda")
    dst = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    lib.kvzx_yuv420_to_rgb32_device.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    p = src.data_ptr()
    def once():
        assert lib.kvzx_yuv420_to_rgb32_device(p, p + w * h, p + w * h + w * h // 4, w, w // 2, dst.data_ptr(), w, h, 0, None) == 1
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        once()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.iters
    nbytes = w * h * 5.5
    out = {"metric": "i420_to_rgb32_fps", "value": round(1e3 / ms, 1), "unit": "frames/s", "width": w, "height": h,
           "us_per_picture": round(ms * 1e3, 2),
           "roofline": {"bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / 8000.0, 4),
                        "algorithmic_bytes_per_launch": int(nbytes)}}
    try:
        import refcolor
        if refcolor.available():
            import numpy as np
            host = src.cpu().numpy()
            refcolor.reference("avx2_mt", host, w, h)
            t0 = time.time(); n = 0
            while time.time() - t0 < 2.0:
                refcolor.reference("avx2_mt", host, w, h); n += 1
            out["cpu_baseline"] = {"value": round(n / (time.time() - t0), 1), "unit": "frames/s", "cores": 4, "kind": "reference",
                                   "sample": "yuv420_to_rgb_i_avx2_mt (4 OpenMP threads), same picture, host memory"}
    except Exception as e:
        out["cpu_baseline"] = {"value": None, "kind": "reference", "sample": "failed: %s" % e}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
