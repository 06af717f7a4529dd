"""ctypes access to the reference's own yuvconversions.cpp, compiled from where it lies by __graft_entry__.build()
into oracle/_ref/libyuvconversions_ref.so (never copied into the repo), plus a numpy restatement of the two
arithmetics it contains (src/media/processing/yuvconversions.cpp:72-420 SIMD converters, :423-493 scalar fallback).
Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libyuvconversions_ref.so")

SYMS = {          # Itanium-mangled names of the free functions declared in yuvconversions.h:9-12
    "c": "_Z17yuv420_to_rgb_i_cPhS_tt",
    "sse41": "_Z21yuv420_to_rgb_i_sse41PhS_tt",
    "avx2": "_Z20yuv420_to_rgb_i_avx2PhS_tt",
    "avx2_mt": "_Z23yuv420_to_rgb_i_avx2_mtPhS_tth",
    # yuvconversions.h:15-17
    "rgb2yuv_c": "_Z17rgb_to_yuv420_i_cPhS_tt",
    "rgb2yuv_sse41": "_Z21rgb_to_yuv420_i_sse41PhS_ii",
}


def available():
    return os.path.exists(REF_SO)


def reference(variant, i420, w, h, fill=0x5A):
    """run the reference converter `variant` on a packed I420 picture; the output buffer starts filled with `fill`"""
    lib = C.CDLL(REF_SO)
    fn = getattr(lib, SYMS[variant])
    fn.restype = C.c_int
    i420 = np.ascontiguousarray(i420, dtype=np.uint8)
    out = np.full(w * h * 4, fill, dtype=np.uint8)
    args = [i420.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint16(w), C.c_uint16(h)]
    if variant == "avx2_mt":
        args.append(C.c_uint8(4))
    fn(*args)
    return out


def restatement(variant, i420, w, h, fill=0x5A):
    """numpy statement of the same arithmetic: variant "c" (scalar fallback) or "simd" (SSE4.1/AVX2 converters)"""
    i420 = np.asarray(i420, dtype=np.uint8)
    y = i420[:w * h].reshape(h, w).astype(np.int32)
    up = i420[w * h:w * h + w * h // 4].reshape(h // 2, w // 2).astype(np.int32) - 128
    vp = i420[w * h + w * h // 4:].reshape(h // 2, w // 2).astype(np.int32) - 128
    u = np.repeat(np.repeat(up, 2, 0), 2, 1)
    v = np.repeat(np.repeat(vp, 2, 0), 2, 1)
    out = np.full((h, w, 4), fill, dtype=np.uint8)
    if variant == "simd":
        r = y + v + (v >> 2) + (v >> 3) + (v >> 5)
        g = y - ((u >> 2) + (u >> 4) + (u >> 5) + (v >> 1) + (v >> 3) + (v >> 4) + (v >> 5))
        b = y + u + (u >> 1) + (u >> 2) + (u >> 6)
        out[..., 0] = np.clip(b, 0, 255); out[..., 1] = np.clip(g, 0, 255); out[..., 2] = np.clip(r, 0, 255); out[..., 3] = 0
    else:
        cr, cb = u, v
        out[..., 0] = np.clip(y + cr + (cr >> 2) + (cr >> 3) + (cr >> 5), 0, 255)
        out[..., 1] = np.clip(y - ((cb >> 2) + (cb >> 4) + (cb >> 5)) - ((cr >> 1) + (cr >> 3) + (cr >> 4) + (cr >> 5)), 0, 255)
        out[..., 2] = np.clip(y + cb + (cb >> 1) + (cb >> 2) + (cb >> 6), 0, 255)
    return out.reshape(-1)


def random_i420(seed, w, h):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    a[:64] = np.concatenate([np.zeros(16, np.uint8), np.full(16, 255, np.uint8), np.arange(32, dtype=np.uint8) * 8])   # extremes
    return a


def reference_rgb2yuv(variant, rgb32, w, h, fill=0x5A):
    """the reference's rgb_to_yuv420_i_c ("c") or rgb_to_yuv420_i_sse41 ("sse41") on a w x h RGB32 picture -> packed I420"""
    lib = C.CDLL(REF_SO)
    fn = getattr(lib, SYMS["rgb2yuv_" + variant])
    rgb32 = np.ascontiguousarray(rgb32, dtype=np.uint8)
    out = np.full(w * h * 3 // 2, fill, dtype=np.uint8)
    if variant == "c":
        fn.restype = None
        fn(rgb32.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint16(w), C.c_uint16(h))
    else:
        fn.restype = C.c_int
        fn(rgb32.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h))
    return out


def restatement_rgb2yuv(variant, rgb32, w, h):
    """numpy statement of the two arithmetics (yuvconversions.cpp:770-797 and :634-767)"""
    p = np.asarray(rgb32, dtype=np.uint8).reshape(h, w, 4).astype(np.int32)
    b0, b1, b2 = p[..., 0], p[..., 1], p[..., 2]
    if variant == "c":
        y = ((76 * b0 + 150 * b1 + 29 * b2 + 128) >> 8) & 255
        us = 127 * b0 - 84 * b1 - 43 * b2
        vs = -21 * b0 - 106 * b1 + 127 * b2
        blk = lambda a: a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]
        u = (((blk(us) + 512) >> 10) + 128) & 255
        v = (((blk(vs) + 512) >> 10) + 128) & 255
    else:
        y = np.clip((76 * b2 + 150 * b1 + 29 * b0) >> 8, 0, 255)[::-1]
        us = -43 * b2 - 84 * b1 + 127 * b0
        vs = 127 * b2 - 106 * b1 - 21 * b0
        col = lambda a: np.clip((a[0::2] + a[1::2] + 255 * 255) >> 9, 0, 255)
        tu, tv = col(us), col(vs)
        u = ((tu[:, 0::2] + tu[:, 1::2]) >> 1)[::-1]
        v = ((tv[:, 0::2] + tv[:, 1::2]) >> 1)[::-1]
    return np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).astype(np.uint8)


def random_rgb32(seed, w, h):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, w * h * 4, dtype=np.uint8)
    a[:32] = np.concatenate([np.zeros(16, np.uint8), np.full(16, 255, np.uint8)])       # extremes
    return a
