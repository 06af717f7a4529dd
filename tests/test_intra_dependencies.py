"""Which intra prediction modes read a block's above-right / below-left reference samples (directly or through the [1 2 1] reference filter of
8.4.4.2.3): the masks in kvazzup_amd/csrc/hevc_core.h (intra_uses_above_right / intra_uses_below_left) drive two things -- the intra chains
(k_intra_recon, k_dec_intra) wait for a neighbouring CTU only as far as a block's MODE reads it, and the encoder's "intra-chain" restriction keeps
the blocks whose above-right / below-left samples lie in another CTU to the modes that do not read them (statement: oracle/hevc_enc.c
intra_analyse_size).  Here the masks are re-derived by perturbation -- replace those samples by random ones, see whether the prediction moves --
on three predictors: the checker's, the product's host build and the Python decoder's."""
import ctypes as C

import numpy as np
import pytest

import hc
import orc


def _orc_predict(left, top, n, cidx, mode):
    L = orc.lib()
    L.orc_api_intra_predict.argtypes = [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]
    out = np.zeros(n * n, np.uint8)
    L.orc_api_intra_predict(left.ctypes.data, top.ctypes.data, n, cidx, mode, 1, out.ctypes.data)
    return out


def _hc_predict(left, top, n, cidx, mode):
    out = np.zeros(n * n, np.uint8)
    hc.lib().hc_intra_predict(left.ctypes.data, top.ctypes.data, n, cidx, mode, out.ctypes.data)
    return out


def _masks(predict, n, cidx, trials=16):
    rng = np.random.default_rng(n * 8 + cidx)
    tr = bl = 0
    for mode in range(35):
        for _ in range(trials):
            left = rng.integers(0, 256, 2 * n + 1, dtype=np.uint8)
            top = rng.integers(0, 256, 2 * n + 1, dtype=np.uint8)
            top[0] = left[0]
            base = predict(left, top, n, cidx, mode)
            t2 = top.copy(); t2[1 + n:] = rng.integers(0, 256, n, dtype=np.uint8)          # p[x][-1], x >= n
            l2 = left.copy(); l2[1 + n:] = rng.integers(0, 256, n, dtype=np.uint8)         # p[-1][y], y >= n
            if not np.array_equal(base, predict(left, t2, n, cidx, mode)):
                tr |= 1 << mode
            if not np.array_equal(base, predict(l2, top, n, cidx, mode)):
                bl |= 1 << mode
    return tr, bl


@pytest.mark.parametrize("cidx,n", [(0, 4), (0, 8), (0, 16), (0, 32), (1, 4), (1, 8), (1, 16)])
@pytest.mark.parametrize("which", ["checker", "product"])
def test_masks_match_the_predictors(cidx, n, which):
    L = hc.lib()
    L.hc_intra_uses.restype = C.c_uint64
    L.hc_intra_uses.argtypes = [C.c_int, C.c_int, C.c_int]
    log2n = n.bit_length() - 1
    want = (L.hc_intra_uses(log2n, cidx, 0), L.hc_intra_uses(log2n, cidx, 1))
    got = _masks(_orc_predict if which == "checker" else _hc_predict, n, cidx)
    if n == 32:                      # the strong filter (random samples rarely meet its condition) ties every filtered mode to both far corners: the masks say so wholesale
        assert got[0] & ~want[0] == 0 and got[1] & ~want[1] == 0 and want[0] == want[1] == 0x7ffffffff & ~((1 << 1) | (1 << 10) | (1 << 26))
        return
    assert got == want, ("above-right 0x%x / 0x%x, below-left 0x%x / 0x%x" % (got[0], want[0], got[1], want[1]))


def test_intra_chain_restriction_in_the_checker_encoder():
    """"intra-chain" on (default): a CTU's above-right corner block and its left-edge blocks only take modes that do not read the neighbouring CTU's above-right /
    below-left samples; off: all modes; both streams decode to the encoder's reconstruction, the restriction costs the intra picture a few per cent"""
    w, h = 320, 192
    L = hc.lib()
    L.hc_intra_uses.restype = C.c_uint64
    L.hc_intra_uses.argtypes = [C.c_int, C.c_int, C.c_int]
    sizes = {}
    for on in (1, 0):
        oe = orc.OracleEncoder(w, h, qp=30, period=1, me_range=8)
        oe.set_option("intra-chain", on)
        od = orc.OracleDecoder()
        au = oe.encode(orc.synth_frame(0, 5, w, h, 0))
        out = od.decode_au(au, 0)
        assert len(out) == 1 and np.array_equal(out[0]["i420"], oe.recon())
        d = oe.debug()
        sizes[on] = len(au)
        bad = 0
        for by in range(h // 8):
            for bx in range(w // 8):
                l2, mode = int(d["cu_log2"][by, bx]), int(d["cu_intra_mode"][by, bx])
                n = 1 << l2
                x0, y0 = (bx * 8) & ~(n - 1), (by * 8) & ~(n - 1)
                corner = (y0 & 63) == 0 and ((x0 + n) & 63) == 0 and y0 > 0 and x0 + n < (w + 63) // 64 * 64
                edge = (x0 & 63) == 0 and x0 > 0 and ((y0 + n) & 63) != 0
                if corner and (L.hc_intra_uses(l2, 0, 0) >> mode) & 1:
                    bad += 1
                if edge and (L.hc_intra_uses(l2, 0, 1) >> mode) & 1:
                    bad += 1
        assert (bad == 0) if on else (bad > 0), (on, bad)
        oe.close(); od.close()
    assert sizes[0] <= sizes[1] < sizes[0] * 1.06, sizes
