"""Known-answer tests that pin the CPU checker (oracle/) to the normative constants of ITU-T H.265
 -- the reference itself holds no vectors for this path,
so these, the closed loop and the numpy second restatement are what the checker stands on."""
import ctypes as C
import json
import os

import numpy as np

import orc

HERE = os.path.dirname(os.path.abspath(__file__))


def table(which, dtype, shape):
    out = np.zeros(shape, dtype=dtype)
    orc.lib().orc_api_tables(which, out.ctypes.data_as(C.c_void_p))
    return out


def test_transform_matrices():
    m = table(0, np.int8, (32, 32)).astype(int)
    assert (m[0] == 64).all()
    # 4-point basis = rows 0, 8, 16, 24 restricted to 4 columns (H.265 8.6.4.2)
    assert m[[0, 8, 16, 24], :4].tolist() == [[64, 64, 64, 64], [83, 36, -36, -83], [64, -64, -64, 64], [36, -83, 83, -36]]
    assert m[4, :8].tolist() == [89, 75, 50, 18, -18, -50, -75, -89]
    assert m[1, :16].tolist() == [90, 90, 88, 85, 82, 78, 73, 67, 61, 54, 46, 38, 31, 22, 13, 4]
    assert m[3, :8].tolist() == [90, 82, 67, 46, 22, -4, -31, -54]
    # rows are (anti)symmetric and nearly orthogonal with norm ~ 64 * sqrt(32)
    for k in range(32):
        assert (m[k] == (1 if k % 2 == 0 else -1) * m[k, ::-1]).all()
    g = m @ m.T
    assert np.abs(g - np.diag(np.diag(g))).max() <= 400 and np.abs(np.diag(g) - 64 * 64 * 32).max() <= 200   # off-diagonal < 0.3 % of the norm
    assert table(5, np.int8, (4, 4)).tolist() == [[29, 55, 74, 84], [74, 74, 0, -74], [84, -29, -74, 55], [55, -84, 74, -29]]


def test_prediction_and_filter_tables():
    ang = table(9, np.int8, (35,)).tolist()
    assert ang[2:] == [32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32]
    inv = table(10, np.int16, (35,)).tolist()
    assert inv[11:26] == [-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096]
    for m in range(11, 26):                      # invAngle = round(8192 / angle)
        if ang[m]:
            assert abs(inv[m] - round(8192 / ang[m])) <= 1
    lf = table(11, np.int8, (4, 8))
    cf = table(12, np.int8, (8, 4))
    assert (lf.sum(axis=1) == 64).all() and (cf.sum(axis=1) == 64).all()
    assert lf[2].tolist() == [-1, 4, -11, 40, 40, -11, 4, -1] and lf[1].tolist() == lf[3][::-1].tolist()
    assert cf[4].tolist() == [-4, 36, 36, -4] and cf[1].tolist() == cf[7][::-1].tolist()


def test_deblocking_and_qp_tables():
    beta, tc, cqp = table(6, np.uint8, (52,)), table(7, np.uint8, (54,)), table(8, np.uint8, (58,))
    assert (beta[:16] == 0).all() and beta[16] == 6 and beta[28] == 18 and beta[29] == 20 and beta[51] == 64
    assert (np.diff(beta.astype(int)) >= 0).all() and (np.diff(tc.astype(int)) >= 0).all()
    assert (tc[:18] == 0).all() and tc[18] == 1 and tc[27] == 2 and tc[53] == 24 and tc[47] == 13
    assert cqp[:30].tolist() == list(range(30)) and cqp[30:44].tolist() == [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37]
    assert cqp[44:].tolist() == [q - 6 for q in range(44, 58)]


def test_cabac_tables_follow_the_probability_model():
    lps = table(1, np.uint8, (64, 4)).astype(float)
    nxt = table(2, np.uint8, (64,))
    # H.264/H.265 derive rangeTabLps[s][q] ~ p_s * (288 + 64 q) with p_s = 0.5 * alpha^s, alpha = (0.01875/0.5)^(1/63)
    alpha = (0.01875 / 0.5) ** (1.0 / 63)
    for s in range(63):
        for q in range(4):
            model = min(128.0, 0.5 * alpha ** s * (288 + 64 * q)) if q == 0 else 0.5 * alpha ** s * (288 + 64 * q)   # first column is capped at 128
            assert abs(lps[s, q] - model) <= max(2.0, 0.03 * model), (s, q, lps[s, q], model)
    assert (lps[63] == 2).all() and (np.diff(lps[:63], axis=0) <= 0).all() and (np.diff(lps, axis=1) >= 0).all()
    assert nxt[0] == 0 and nxt[63] == 63 and (np.diff(nxt[:63].astype(int)) >= 0).all()
    for s in range(1, 63):                      # an LPS moves the estimate towards p = 0.5 by roughly the adaptation rate
        assert nxt[s] < s
    assert table(13, np.uint8, (64,)).tolist() == [min(s + 1, 62) for s in range(62)] + [62, 63]


def test_context_initialisation_formula():
    out = np.zeros(154, np.uint8)
    init = table(3, np.uint8, (3, 154)).astype(int)
    for t in range(3):
        for qp in (0, 22, 32, 51):
            orc.lib().orc_api_cabac_init(t, qp, out.ctypes.data_as(C.c_void_p))
            v = init[t]
            pre = np.clip((((v >> 4) * 5 - 45) * qp >> 4) + ((v & 15) << 3) - 16, 1, 126)
            mps = (pre > 63).astype(int)
            state = np.where(mps == 1, pre - 64, 63 - pre)
            assert (out == ((state << 1) | mps)).all()
    # a few values every HEVC implementation agrees on
    assert init[0][2:5].tolist() == [139, 141, 157] and init[1][6:9].tolist() == [197, 185, 201] and init[1][16] == 79


def test_arithmetic_coder_round_trip_and_stop_bit():
    rng = np.random.default_rng(7)
    for n in (1, 17, 1000, 20000):
        kinds = rng.choice([0, 0, 0, 1, 2], size=n).astype(np.uint8)
        ci = rng.integers(0, 154, size=n).astype(np.uint8)
        bins = (rng.random(n) < 0.2).astype(np.uint8)
        bins[kinds == 2] = 0
        out = np.zeros(n + 64, np.uint8)
        dec = np.zeros(n, np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        orc.lib().orc_api_cabac_roundtrip.restype = C.c_long
        ln = orc.lib().orc_api_cabac_roundtrip(p(kinds), p(ci), p(bins), C.c_long(n), 30, p(out), C.c_long(len(out)), p(dec))
        assert ln > 0, "terminating bin / consumed length mismatch"
        assert (dec == bins).all()
        assert out[ln - 1] != 0          # the last byte carries the stop bit: a substream never ends in 0x00


def test_golden_fixture():
    """tests/golden/oracle_streams.json: digests of access units and reconstructions produced by the
    checker when the fixture was made (make_golden.py); any later change to the checker must be deliberate."""
    import hashlib
    with open(os.path.join(HERE, "golden", "oracle_streams.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        c = case["config"]
        e = orc.OracleEncoder(c["w"], c["h"], qp=c["qp"], period=c["period"], me_range=c["me_range"], wpp=c["wpp"], deblock=c["deblock"],
                                tile_rows=c.get("tile_rows", 1), sao=c.get("sao", 0), subme=c.get("subme", 0), tile_cols=c.get("tile_cols", 1), slices=c.get("slices", 0))
        if c.get("scaling_list"):
            e.set_option("scaling-list", 1)
        if c.get("lossless"):
            e.set_option("lossless", 1)
        for t, want in enumerate(case["frames"]):
            au = e.encode(orc.synth_frame(c["kind"], c["seed"], c["w"], c["h"], t))
            assert hashlib.md5(au).hexdigest() == want["au_md5"], (c, t)
            assert hashlib.md5(e.recon().tobytes()).hexdigest() == want["recon_md5"], (c, t)
            assert len(au) == want["au_bytes"]
        e.close()
