#!/usr/bin/env python3
"""Writes tests/golden/color_i420_to_rgb32.npz: inputs and the outputs of the REFERENCE's converters
(oracle/_ref/libyuvconversions_ref.so, built by __graft_entry__.build() from
/root/reference/src/media/processing/yuvconversions.cpp) for small pictures.  Run in the build container."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import refcolor

cases = {}
for name, (w, h, variant) in {"simd_64x32": (64, 32, "avx2_mt"), "simd_48x16": (48, 16, "sse41"), "c_34x18": (34, 18, "c"), "c_64x32": (64, 32, "c")}.items():
    src = refcolor.random_i420(0xC0101 + w + h, w, h)
    cases[name + "_in"] = src
    cases[name + "_out"] = refcolor.reference(variant, src, w, h)
    cases[name + "_dims"] = np.array([w, h], dtype=np.int32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "color_i420_to_rgb32.npz"), **cases)
print("wrote", len(cases) // 3, "cases")

# RGB32 -> I420 (yuvconversions.cpp:634-797): inputs and the reference's outputs
cases = {}
for name, (w, h, variant) in {"c_64x32": (64, 32, "c"), "c_36x18": (36, 18, "c"), "sse41_64x32": (64, 32, "sse41"), "sse41_40x10": (40, 10, "sse41")}.items():
    src = refcolor.random_rgb32(0xC0202 + w + h, w, h)
    cases[name + "_in"] = src
    cases[name + "_out"] = refcolor.reference_rgb2yuv(variant, src, w, h)
    cases[name + "_dims"] = np.array([w, h], dtype=np.int32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "color_rgb32_to_i420.npz"), **cases)
print("wrote", len(cases) // 3, "rgb32 -> i420 cases")
