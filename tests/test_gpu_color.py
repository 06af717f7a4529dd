"""Row f1 on the GPU: the HIP I420 -> RGB32 kernels against the golden vectors, the numpy restatement at full size
and -- when oracle/_ref travelled with the snapshot -- the reference's own object code run beside them."""
import ctypes as C
import os

import numpy as np
import pytest

import refcolor

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_i420_to_rgb32.npz"))


def hip_convert(lib, src, w, h, variant, fill=0x5A):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.full(w * h * 4, fill, dtype=np.uint8)
    lib.kvzx_yuv420_to_rgb32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    assert lib.kvzx_yuv420_to_rgb32(src.ctypes.data, out.ctypes.data, w, h, variant) == 1
    return out


@pytest.mark.gpu
def test_golden_vectors(gpu):
    for case in sorted(k[:-3] for k in GOLD.files if k.endswith("_in")):
        w, h = (int(x) for x in GOLD[case + "_dims"])
        variant = 2 if case.startswith("simd") else 1
        assert np.array_equal(hip_convert(gpu, GOLD[case + "_in"], w, h, variant), GOLD[case + "_out"]), case


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (130, 70)])
def test_full_size_against_reference_and_restatement(gpu, w, h):
    src = refcolor.random_i420(w * 31 + h, w, h)
    got = hip_convert(gpu, src, w, h, 0)                     # the filter's own choice of arithmetic
    kind = "simd" if w % 16 == 0 else "c"
    assert np.array_equal(got, refcolor.restatement(kind, src, w, h))
    if refcolor.available():
        assert np.array_equal(got, refcolor.reference("avx2_mt" if kind == "simd" else "c", src, w, h))


@pytest.mark.gpu
def test_decoder_output_converted_in_hbm(gpu):
    """decode -> kvzx_decoder_output_rgb32_device: same bytes as converting the downloaded I420 picture"""
    import orc
    from kvazzup_amd.codec import Decoder
    w, h = 320, 192
    oe = orc.OracleEncoder(w, h, qp=30, period=4, me_range=8)
    gd = Decoder(download=True)
    hip = C.CDLL("libamdhip64.so")
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(w * h * 4)) == 0
    gpu.kvzx_decoder_output_rgb32_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    for t in range(3):
        pics = gd.decode_au(oe.encode(orc.synth_frame(0, 7, w, h, t)), t)
        assert len(pics) == 1
        assert gpu.kvzx_decoder_output_rgb32_device(gd.h, dptr, 0) == 1
        out = np.empty(w * h * 4, dtype=np.uint8)
        assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), dptr, C.c_size_t(out.size), 2) == 0
        assert np.array_equal(out, refcolor.restatement("simd", pics[0]["i420"], w, h)), t
    hip.hipFree(dptr)
    gd.close(); oe.close()


# ---- RGB32 -> I420
GOLD2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_rgb32_to_i420.npz"))


def hip_rgb2yuv(lib, src, w, h, variant):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.full(w * h * 3 // 2, 0x5A, dtype=np.uint8)
    lib.kvzx_rgb32_to_yuv420.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    assert lib.kvzx_rgb32_to_yuv420(src.ctypes.data, out.ctypes.data, w, h, variant) == 1
    return out


@pytest.mark.gpu
def test_rgb2yuv_golden_vectors(gpu):
    for case in sorted(k[:-3] for k in GOLD2.files if k.endswith("_in")):
        w, h = (int(x) for x in GOLD2[case + "_dims"])
        assert np.array_equal(hip_rgb2yuv(gpu, GOLD2[case + "_in"], w, h, 2 if case.startswith("sse41") else 1), GOLD2[case + "_out"]), case


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (132, 70)])
def test_rgb2yuv_full_size_against_reference_and_restatement(gpu, w, h):
    src = refcolor.random_rgb32(w * 17 + h, w, h)
    for variant, name in ((1, "c"), (2, "sse41")):
        got = hip_rgb2yuv(gpu, src, w, h, variant)
        assert np.array_equal(got, refcolor.restatement_rgb2yuv(name, src, w, h)), name
        if refcolor.available():
            assert np.array_equal(got, refcolor.reference_rgb2yuv(name, src, w, h)), name


@pytest.mark.gpu
def test_rgb_round_trip_through_both_converters(gpu):
    """I420 -> RGB32 -> I420 with the SIMD arithmetics comes back upside down and close to the original (a property that needs no checker)"""
    w, h = 640, 360
    src = refcolor.random_i420(3, w, h)
    src[:w * h] = np.clip(src[:w * h].astype(int) // 2 + 64, 16, 235)           # keep clear of the clamps
    src[w * h:] = 128
    rgb = hip_convert(gpu, src, w, h, 2)
    back = hip_rgb2yuv(gpu, rgb, w, h, 2)
    y0, y1 = src[:w * h].reshape(h, w).astype(int), back[:w * h].reshape(h, w).astype(int)[::-1]
    assert np.abs(y0 - y1).max() <= 2
