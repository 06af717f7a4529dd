"""Row f1 (I420 -> RGB32): the reference's converters pinned three ways -- golden vectors made from the reference's
own object code (tests/golden/make_color_golden.py), a numpy restatement of both arithmetics, and (when
oracle/_ref/ has been built) the reference object code itself, including that its three SIMD variants agree."""
import os

import numpy as np
import pytest

import refcolor

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_i420_to_rgb32.npz"))
CASES = sorted(k[:-3] for k in GOLD.files if k.endswith("_in"))


@pytest.mark.parametrize("case", CASES)
def test_numpy_restatement_matches_the_golden_vectors(case):
    w, h = (int(x) for x in GOLD[case + "_dims"])
    variant = "simd" if case.startswith("simd") else "c"
    assert np.array_equal(refcolor.restatement(variant, GOLD[case + "_in"], w, h), GOLD[case + "_out"])


@pytest.mark.skipif(not refcolor.available(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_object_code_matches_the_golden_vectors_and_its_simd_variants_agree():
    for case in CASES:
        w, h = (int(x) for x in GOLD[case + "_dims"])
        src = GOLD[case + "_in"]
        if case.startswith("simd"):
            outs = [refcolor.reference(v, src, w, h) for v in ("sse41", "avx2", "avx2_mt")]
            assert all(np.array_equal(o, GOLD[case + "_out"]) for o in outs), case
        else:
            assert np.array_equal(refcolor.reference("c", src, w, h), GOLD[case + "_out"]), case
    # the two arithmetics really are different (the scalar fallback swaps the chroma planes and keeps byte 3)
    src = refcolor.random_i420(5, 64, 32)
    assert not np.array_equal(refcolor.reference("c", src, 64, 32), refcolor.reference("avx2", src, 64, 32))


# ---- RGB32 -> I420 (rgb_to_yuv420_i_c / rgb_to_yuv420_i_sse41, yuvconversions.cpp:634-797)
GOLD2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_rgb32_to_i420.npz"))
CASES2 = sorted(k[:-3] for k in GOLD2.files if k.endswith("_in"))


@pytest.mark.parametrize("case", CASES2)
def test_rgb2yuv_numpy_restatement_matches_the_golden_vectors(case):
    w, h = (int(x) for x in GOLD2[case + "_dims"])
    assert np.array_equal(refcolor.restatement_rgb2yuv("sse41" if case.startswith("sse41") else "c", GOLD2[case + "_in"], w, h), GOLD2[case + "_out"])


@pytest.mark.skipif(not refcolor.available(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_rgb2yuv_reference_object_code_matches_the_golden_vectors():
    for case in CASES2:
        w, h = (int(x) for x in GOLD2[case + "_dims"])
        assert np.array_equal(refcolor.reference_rgb2yuv("sse41" if case.startswith("sse41") else "c", GOLD2[case + "_in"], w, h), GOLD2[case + "_out"]), case
    # the two converters differ in more than rounding: the SSE one turns the picture upside down
    src = refcolor.random_rgb32(9, 64, 32)
    a, b = refcolor.reference_rgb2yuv("c", src, 64, 32), refcolor.reference_rgb2yuv("sse41", src, 64, 32)
    assert not np.array_equal(a, b)
