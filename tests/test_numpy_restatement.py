"""A second, independent restatement (numpy) of the arithmetic stages, written from H.265 directly, to
catch transcription errors in the C checker: transform / quantisation (8.6), intra prediction (8.4.4.2),
interpolation (8.5.3.3.3)."""
import ctypes as C

import numpy as np
import pytest

import orc
from test_oracle_kat import table

P = lambda a: a.ctypes.data_as(C.c_void_p)


def dct_matrix(n):
    """integer DCT-II basis from the cosine definition: round(64 * sqrt(2) * cos(...)) with the standard's hand-tuned values"""
    m = table(0, np.int8, (32, 32)).astype(np.int64)
    return m[::32 // n, :n]


def np_forward(res, n):
    c = dct_matrix(n)
    l2 = int(np.log2(n))
    s1, s2 = l2 - 1, l2 + 6
    tmp = (res.astype(np.int64) @ c.T + (1 << (s1 - 1))) >> s1
    return np.clip((c @ tmp + (1 << (s2 - 1))) >> s2, -32768, 32767)


def np_inverse(coef, n):
    c = dct_matrix(n)
    g = np.clip((c.T @ coef.astype(np.int64) + 64) >> 7, -32768, 32767)
    return np.clip((g @ c + 2048) >> 12, -32768, 32767)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_transform_pair(n):
    rng = np.random.default_rng(n)
    L = orc.lib()
    for trial in range(40):
        amp = [255, 255, 30, 3][trial % 4]
        res = rng.integers(-amp, amp + 1, size=(n, n)).astype(np.int16)
        out = np.zeros((n, n), np.int16)
        L.orc_fwd_transform(P(res), P(out), n, 0)
        assert np.array_equal(out, np_forward(res, n))
        coef = (rng.integers(-2000, 2001, size=(n, n)) * (rng.random((n, n)) < 0.2)).astype(np.int16)
        L.orc_inv_transform(P(coef), P(out), n, 0)
        assert np.array_equal(out, np_inverse(coef, n))
    # the cosine definition itself: entries deviate from 64*sqrt(2)*cos by less than 1.5 (hand-tuned integers)
    k, x = np.mgrid[0:32, 0:32]
    ideal = 64 * np.sqrt(2) * np.cos(np.pi * (2 * x + 1) * k / 64)
    ideal[0] = 64
    assert np.abs(table(0, np.int8, (32, 32)) - ideal).max() < 1.5


@pytest.mark.parametrize("qp", [0, 10, 22, 32, 37, 51])
def test_quant_dequant(qp):
    rng = np.random.default_rng(qp)
    L = orc.lib()
    f = [26214, 23302, 20560, 18396, 16384, 14564][qp % 6]
    g = [40, 45, 51, 57, 64, 72][qp % 6]
    for n in (4, 8, 16, 32):
        l2 = int(np.log2(n))
        coef = rng.integers(-32768, 32768, size=(n, n)).astype(np.int16)
        for intra in (0, 1):
            lev = np.zeros((n, n), np.int16)
            L.orc_quant(P(coef), P(lev), n, qp, intra)
            shift = 14 + qp // 6 + (15 - 8 - l2)
            q = (np.abs(coef.astype(np.int64)) * f + ((171 if intra else 85) << (shift - 9))) >> shift
            want = np.sign(coef) * np.minimum(q, 32767)
            assert np.array_equal(lev, want)
            deq = np.zeros((n, n), np.int16)
            L.orc_dequant(P(lev), P(deq), n, qp)
            bd = 8 + l2 - 5
            want = np.clip((lev.astype(np.int64) * 16 * (g << (qp // 6)) + (1 << (bd - 1))) >> bd, -32768, 32767)
            assert np.array_equal(deq, want)


def np_intra(left, top, n, mode, luma=True):
    """8.4.4.2.3-8.4.4.2.6 with strong_intra_smoothing_enabled_flag = 1; left[0] = top[0] = corner"""
    left, top = left.astype(int), top.astype(int)
    ang = table(9, np.int8, (35,)).astype(int)
    inv = table(10, np.int16, (35,)).astype(int)
    if luma and mode != 1 and n != 4:
        thr = {8: 7, 16: 1, 32: 0}[n]
        if min(abs(mode - 26), abs(mode - 10)) > thr:
            c = left[0]
            if n == 32 and abs(c + top[64] - 2 * top[32]) < 8 and abs(c + left[64] - 2 * left[32]) < 8:
                i = np.arange(1, 64)
                lf = left.copy(); tf = top.copy()
                lf[1:64] = ((64 - i) * c + i * left[64] + 32) >> 6
                tf[1:64] = ((64 - i) * c + i * top[64] + 32) >> 6
            else:
                lf = left.copy(); tf = top.copy()
                lf[0] = tf[0] = (left[1] + 2 * c + top[1] + 2) >> 2
                lf[1:2 * n] = (left[2:2 * n + 1] + 2 * left[1:2 * n] + left[0:2 * n - 1] + 2) >> 2
                tf[1:2 * n] = (top[2:2 * n + 1] + 2 * top[1:2 * n] + top[0:2 * n - 1] + 2) >> 2
            left, top = lf, tf
    pred = np.zeros((n, n), int)
    y, x = np.mgrid[0:n, 0:n]
    l2 = int(np.log2(n))
    if mode == 0:
        pred = ((n - 1 - x) * left[1 + y] + (x + 1) * top[1 + n] + (n - 1 - y) * top[1 + x] + (y + 1) * left[1 + n] + n) >> (l2 + 1)
    elif mode == 1:
        dc = (left[1:n + 1].sum() + top[1:n + 1].sum() + n) >> (l2 + 1)
        pred[:] = dc
        if luma and n < 32:
            pred[0, 1:] = (top[2:n + 1] + 3 * dc + 2) >> 2
            pred[1:, 0] = (left[2:n + 1] + 3 * dc + 2) >> 2
            pred[0, 0] = (left[1] + 2 * dc + top[1] + 2) >> 2
    else:
        a = ang[mode]
        main, side = (top, left) if mode >= 18 else (left, top)
        ref = {i: main[i] for i in range(0, 2 * n + 1)}
        last = (n * a) >> 5
        if a < 0 and last < -1:
            for i in range(last, 0):
                ref[i] = side[(i * inv[mode] + 128) >> 8]
        for yy in range(n):
            for xx in range(n):
                u, v = (xx, yy) if mode >= 18 else (yy, xx)        # u along the main reference, v away from it
                idx, fact = ((v + 1) * a) >> 5, ((v + 1) * a) & 31
                val = ref[u + idx + 1] if fact == 0 else ((32 - fact) * ref[u + idx + 1] + fact * ref[u + idx + 2] + 16) >> 5
                pred[yy, xx] = val
        if luma and n < 32 and mode == 26:
            pred[:, 0] = np.clip(top[1] + ((left[1:n + 1] - left[0]) >> 1), 0, 255)
        if luma and n < 32 and mode == 10:
            pred[0, :] = np.clip(left[1] + ((top[1:n + 1] - left[0]) >> 1), 0, 255)
    return pred.astype(np.uint8)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_intra_prediction_all_modes(n):
    rng = np.random.default_rng(100 + n)
    L = orc.lib()
    for trial in range(6):
        if trial % 3 == 0:      # smooth references: exercises the strong filter at 32x32
            base = rng.integers(40, 200)
            left = (base + np.arange(2 * n + 1) // 8).astype(np.uint8)
            top = (base + np.arange(2 * n + 1) // 6).astype(np.uint8)
        else:
            left = rng.integers(0, 256, 2 * n + 1).astype(np.uint8)
            top = rng.integers(0, 256, 2 * n + 1).astype(np.uint8)
        top[0] = left[0]
        for cidx in (0, 1):
            for mode in range(35):
                out = np.zeros((n, n), np.uint8)
                L.orc_api_intra_predict(P(left), P(top), n, cidx, mode, 1, P(out))
                want = np_intra(left, top, n, mode, luma=(cidx == 0))
                assert np.array_equal(out, want), (n, cidx, mode, trial)


def test_interpolation_filters():
    rng = np.random.default_rng(5)
    L = orc.lib()
    w = h = 48
    ref = rng.integers(0, 256, (h, w)).astype(np.uint8)
    lf = table(11, np.int8, (4, 8)).astype(int)
    cf = table(12, np.int8, (8, 4)).astype(int)
    pad = np.pad(ref.astype(int), 8, mode="edge")

    def at(y, x):
        return pad[y + 8, x + 8]
    for (mvx, mvy) in [(0, 0), (4, -8), (1, 0), (0, 3), (2, 2), (-5, 7), (9, -3)]:
        out = np.zeros((8, 8), np.int16)
        L.orc_mc_luma(P(ref), w, w, h, 16, 16, 8, 8, mvx, mvy, P(out), 8)
        xf, yf, xi, yi = mvx & 3, mvy & 3, 16 + (mvx >> 2), 16 + (mvy >> 2)
        want = np.zeros((8, 8), int)
        for y in range(8):
            for x in range(8):
                if xf == 0 and yf == 0:
                    v = at(yi + y, xi + x) << 6
                elif yf == 0:
                    v = sum(lf[xf][i] * at(yi + y, xi + x + i - 3) for i in range(8))
                elif xf == 0:
                    v = sum(lf[yf][i] * at(yi + y + i - 3, xi + x) for i in range(8))
                else:
                    v = sum(lf[yf][j] * sum(lf[xf][i] * at(yi + y + j - 3, xi + x + i - 3) for i in range(8)) for j in range(8)) >> 6
                want[y, x] = v
        assert np.array_equal(out, want), (mvx, mvy)
    for (mvx, mvy) in [(0, 0), (4, 0), (0, 4), (4, 4), (3, -5), (-9, 6)]:
        out = np.zeros((8, 8), np.int16)
        L.orc_mc_chroma(P(ref), w, w, h, 16, 16, 8, 8, mvx, mvy, P(out), 8)
        xf, yf, xi, yi = mvx & 7, mvy & 7, 16 + (mvx >> 3), 16 + (mvy >> 3)
        want = np.zeros((8, 8), int)
        for y in range(8):
            for x in range(8):
                if xf == 0 and yf == 0:
                    v = at(yi + y, xi + x) << 6
                elif yf == 0:
                    v = sum(cf[xf][i] * at(yi + y, xi + x + i - 1) for i in range(4))
                elif xf == 0:
                    v = sum(cf[yf][i] * at(yi + y + i - 1, xi + x) for i in range(4))
                else:
                    v = sum(cf[yf][j] * sum(cf[xf][i] * at(yi + y + j - 1, xi + x + i - 1) for i in range(4)) for j in range(4)) >> 6
                want[y, x] = v
        assert np.array_equal(out, want), (mvx, mvy)
