"""CPU tests of the PRODUCT's serial building blocks (kvazzup_amd/csrc/hevc_core.h, hevc_headers.h,
entropy token path) compiled for the host by tests/hostcheck: same access units as the CPU checker from
the same decisions, same merge/AMVP signalling, same intra prediction, same deblocking, same tables."""
import ctypes as C

import numpy as np
import pytest

import hc
import orc
from test_oracle_kat import table

P = lambda a: a.ctypes.data_as(C.c_void_p)


def test_tables_typed_into_the_product_equal_the_generated_ones():
    for which, dtype, shape, owhich in ((0, np.int8, (32, 32), 0), (1, np.uint8, (64, 4), 1), (2, np.uint8, (64,), 2), (3, np.uint8, (3, 154), 3), (4, np.uint16, (52,), 4)):
        a = np.zeros(shape, dtype)
        assert hc.lib().hc_table(which, P(a)) == a.nbytes
        assert np.array_equal(a, table(owhich, dtype, shape))


@pytest.mark.parametrize("cfg", [
    (128, 64, 32, 1, 8, 0, 2, 1), (320, 240, 32, 64, 8, 0, 4, 1), (320, 240, 22, 64, 8, 2, 3, 1), (192, 128, 10, 64, 8, 2, 3, 1),
    (256, 192, 40, 2, 16, 2, 4, 0), (416, 240, 27, 4, 16, 0, 5, 1), (130, 70, 0, 64, 1, 2, 2, 1),
])
def test_entropy_coding_and_signalling_match_the_checker(cfg):
    w, h, qp, period, rng, kind, frames, wpp = cfg
    e = orc.OracleEncoder(w, h, qp=qp, period=period, me_range=rng, wpp=wpp)
    for t in range(frames):
        au = e.encode(orc.synth_frame(kind, 0x5EED0000, w, h, t))
        dbg = e.debug()
        fr, hold = hc.make_frame(dbg, w, h, qp, wpp=wpp, write_ps=1 if dbg["is_intra"] else 0)
        if not dbg["is_intra"]:
            want = {k: hold[k].copy() for k in ("cu_flags", "cu_merge_idx", "cu_mvp_idx")}
            for k in want:
                hold[k][:] = 255
            hc.lib().hc_inter_signal(C.byref(fr))
            merge = (want["cu_flags"] & 2) != 0
            assert np.array_equal(hold["cu_flags"], want["cu_flags"])
            assert np.array_equal(hold["cu_merge_idx"][merge], want["cu_merge_idx"][merge])
            assert np.array_equal(hold["cu_mvp_idx"][~merge], want["cu_mvp_idx"][~merge])
        direct, bins = hc.encode_au(fr)                   # CABAC driven directly
        tokens, ntok = hc.encode_au_tokens(fr)            # k_tokenize emulation + token replay (entropy_host.h)
        assert direct == au and tokens == au, (t, len(au), len(direct), len(tokens))
        assert bins == dbg["bins"] and 0 < ntok <= bins
    e.close()


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_intra_prediction(n):
    rng = np.random.default_rng(n)
    for trial in range(4):
        left = rng.integers(0, 256, 2 * n + 1).astype(np.uint8) if trial else np.full(2 * n + 1, 90, np.uint8)
        top = rng.integers(0, 256, 2 * n + 1).astype(np.uint8) if trial else (90 + np.arange(2 * n + 1) // 9).astype(np.uint8)
        top[0] = left[0]
        for cidx in (0, 1):
            for mode in range(35):
                a, b = np.zeros((n, n), np.uint8), np.zeros((n, n), np.uint8)
                hc.lib().hc_intra_predict(P(left), P(top), n, cidx, mode, P(a))
                orc.lib().orc_api_intra_predict(P(left), P(top), n, cidx, mode, 1, P(b))
                assert np.array_equal(a, b), (n, cidx, mode)


@pytest.mark.parametrize("intra", [True, False])
def test_deblocking(intra):
    w, h, qp = 320, 192, 34
    e = orc.OracleEncoder(w, h, qp=qp, period=1 if intra else 64, me_range=8)
    e.encode(orc.synth_frame(0, 9, w, h, 0))
    if not intra:
        e.encode(orc.synth_frame(0, 9, w, h, 1))
    d = e.debug()
    fr, hold = hc.make_frame(d, w, h, qp)
    planes = [d["predeblock%d" % c].copy() for c in range(3)]
    hc.lib().hc_deblock(C.byref(fr), P(planes[0]), P(planes[1]), P(planes[2]))
    for c in range(3):
        assert np.array_equal(planes[c], d["rec%d" % c]), c
    assert not np.array_equal(d["predeblock0"], d["rec0"])        # the filter did something
    e.close()


def test_scalar_quantiser():
    rng = np.random.default_rng(1)
    L = orc.lib()
    for qp in (0, 17, 32, 51):
        for n in (4, 8, 16, 32):
            coef = rng.integers(-32768, 32768, (n, n)).astype(np.int16)
            lev = np.zeros((n, n), np.int16)
            deq = np.zeros((n, n), np.int16)
            for intra in (0, 1):
                L.orc_quant(P(coef), P(lev), n, qp, intra)
                L.orc_dequant(P(lev), P(deq), n, qp)
                l2 = int(np.log2(n))
                for (y, x) in rng.integers(0, n, (24, 2)):
                    assert hc.lib().hc_quant(int(coef[y, x]), qp, l2, intra) == lev[y, x]
                    assert hc.lib().hc_dequant(int(lev[y, x]), qp, l2) == deq[y, x]
