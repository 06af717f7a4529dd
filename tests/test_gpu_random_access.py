"""GPU parity for random access points that are no IDR pictures (round 6): CRA pictures with RASL / RADL leading pictures read in the middle of a stream and
as its start (the RASL pictures are dropped, 8.1.3), the same picture called BLA by a splicer (BLA_W_LP, BLA_W_RADL, BLA_N_LP), an end of sequence NAL unit in
front of it, and pictures with pic_output_flag = 0 (decoded, referenced, never handed out).  The synthesiser writes the streams (open_gop, hidden_pics); the
cuts are made here.  The HIP decoder must hand out the checker's pictures -- the same ones, in the same order, bit for bit; tests/test_random_access.py holds the
checker itself to the properties of such streams.  Further down, as there: temporal sub-layers, VUI parts, reference picture set forms, ignorable syntax."""
import numpy as np
import pytest

import orc
from test_gpu_foreign import PLAIN
from test_random_access import EOS, rasl_of, rename, vcl_type


def both(cut, pts, threads=1, frame_threads=False):
    """the access units `cut` (time stamps `pts`) through the checker and the product: the pictures must agree, in order"""
    from kvazzup_amd.codec import Decoder
    od = orc.OracleDecoder()
    gd = Decoder(threads=threads, frame_threads=frame_threads) if frame_threads else Decoder()
    want, got = [], []
    try:
        for p, au in zip(pts, cut):
            want += od.decode_au(au, p)
            got += gd.decode_au(au, p)
        want += od.flush()
        got += gd.drain()
        assert [f["pts"] for f in got] == [f["pts"] for f in want]
        for a, b in zip(got, want):
            assert (a["width"], a["height"]) == (b["width"], b["height"])
            if not np.array_equal(a["i420"], b["i420"]):
                d = np.flatnonzero(a["i420"] != b["i420"])
                pytest.fail("picture with time stamp %d: %d samples differ, first at %d" % (a["pts"], len(d), d[0]))
    finally:
        gd.close()
        od.close()
    return [f["pts"] for f in want]


def stream(w, h, n, seed, **kw):
    """access units, their slice NAL unit types, the indices of the CRA pictures -- of the first seed from `seed` on (in steps of 1000) whose stream has a CRA
    picture with leading pictures in its first two thirds"""
    for s in range(seed, seed + 20000, 1000):
        cfg = dict(gop=(2, 4, 8)[s % 3], open_gop=1, intra_period=64, b_slices=50, num_refs=1 + s % 4)
        cfg.update(kw)
        g = orc.OracleGen(w, h, seed=s, **cfg)
        aus = [g.picture() for _ in range(n)]
        g.close()
        types = [vcl_type(a) for a in aus]
        cras = [i for i, t in enumerate(types) if t == 21]
        if cras and cras[0] < 2 * n // 3:
            return aus, types, cras
    raise AssertionError("no CRA picture")


FEATURES = [
    dict(PLAIN),
    dict(PLAIN, tmvp=1, num_refs=3, all_part_modes=1, amp=1),
    dict(PLAIN, tmvp=1, wpp=1, slices=1, sao=1, intra_in_p=20),
    dict(PLAIN, wpp=0, tile_rows=2, tile_cols=2, slices=2, tmvp=1),
    dict(tmvp=1),                                       # everything else drawn from the seed
    dict(tmvp=1, weighted=40, list_mod=40),
    dict(slices=3, tmvp=1),                             # pictures of free slices: closed by what follows them -- which may be a dropped picture's NAL unit
]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("feature", range(len(FEATURES)))
def test_cra_pictures_in_the_stream_and_as_its_start(gpu, feature, seed, threads):
    aus, types, cras = stream(416, 240, 30, seed, **FEATURES[feature])
    assert cras
    shown = both(aus, range(len(aus)), threads, threads > 1)
    assert len(shown) == len(aus)
    for k in cras[:2]:
        shown = both(aus[k:], range(k, len(aus)), threads, threads > 1)
        assert shown and sorted(shown) == [i for i in range(k, len(aus)) if i not in rasl_of(types, k)]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("bla", [16, 17, 18])
@pytest.mark.parametrize("seed", [1, 2, 3, 6, 7])
def test_a_cra_picture_called_bla(gpu, seed, bla, threads):
    aus, types, cras = stream(352, 288, 30, seed, tmvp=1, slices=(0, 1, 3)[seed % 3])
    for k in cras[:2]:
        drop = set(rasl_of(types, k))
        if bla == 18:
            drop |= {i for i in range(k + 1, len(types)) if types[i] in (6, 7) and all(not 16 <= t <= 23 for t in types[k + 1:i])}
        keep = [i for i in range(len(aus)) if bla == 16 or i not in drop]
        shown = both([rename(aus[i], 21, bla) if i == k else aus[i] for i in keep], keep, threads, threads > 1)
        assert sorted(shown) == [i for i in range(len(aus)) if i not in drop]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("seed", [1, 4, 9, 10])
def test_an_end_of_sequence_nal_unit_before_a_cra_picture(gpu, seed, threads):
    aus, types, cras = stream(416, 240, 30, seed, tmvp=1)
    for k in cras[:2]:
        shown = both(aus[:k] + [EOS + aus[k]] + aus[k + 1:], range(len(aus)), threads, threads > 1)
        assert sorted(shown) == [i for i in range(len(aus)) if i not in rasl_of(types, k)]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("seed", range(1, 7))
def test_pictures_that_are_not_output(gpu, seed, threads):
    """pic_output_flag = 0: in streams that reorder (with CRA pictures) and in low-delay ones"""
    aus, types, cras = stream(416, 240, 28, seed, hidden_pics=30, tmvp=1)
    shown = both(aus, range(len(aus)), threads, threads > 1)
    assert 8 < len(shown) < len(aus)
    g = orc.OracleGen(352, 288, seed=seed, hidden_pics=30, intra_period=12, gop=0, b_slices=0)
    low = [g.picture() for _ in range(24)]
    g.close()
    shown = both(low, range(len(low)), threads, threads > 1)
    assert shown == sorted(shown) and 8 < len(shown) < len(low)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(20, 32))
def test_random_streams_with_random_access_points(gpu, seed):
    sizes = [(416, 240), (352, 288), (200, 136), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    aus, types, cras = stream(w, h, 36, seed, hidden_pics=(0, 15)[seed & 1], ctb_log2=(6, 5, 4)[seed % 3], tmvp=1)
    threads = 1 + 2 * (seed % 3)
    both(aus, range(len(aus)), threads, threads > 1)
    for k in cras[:2]:
        both(aus[k:], range(k, len(aus)), threads, threads > 1)
        both([rename(a, 21, 16) if i == k else a for i, a in enumerate(aus)], range(len(aus)), threads, threads > 1)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("seed", range(1, 9))
def test_no_output_of_prior_pics_flag(gpu, seed, threads):
    """C.5.2.2: IDR pictures (and a CRA picture called BLA) whose flag says that what still waits for its turn is not to be shown -- exactly the pictures the standard's
    process holds at that instant disappear, whatever the frame threads' timing"""
    from test_random_access import discard_prior
    g = orc.OracleGen(416, 240, seed=seed, gop=(2, 4, 8)[seed % 3], open_gop=seed & 1, intra_period=7 + seed % 4, b_slices=50, num_refs=1 + seed % 4, tmvp=1,
                      hidden_pics=(0, 10)[seed % 2], slices=(0, 1, 3)[seed % 3])
    aus = [g.picture() for _ in range(32)]
    g.close()
    types = [vcl_type(a) for a in aus]
    cras = [i for i, t in enumerate(types) if t == 21]
    plain = both(aus, range(len(aus)), threads, threads > 1)
    cut = [discard_prior(rename(a, 21, 16)) if cras and i == cras[-1] else discard_prior(a) for i, a in enumerate(aus)]
    shown = both(cut, range(len(aus)), threads, threads > 1)
    assert len(shown) < len(plain)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("seed", range(1, 9))
def test_temporal_sub_layers(gpu, seed, threads):
    """TemporalId in the NAL unit headers, parameter sets with sub-layers, sub-layer non-reference pictures (what Kvazaar's gop=8 sends) -- the whole stream, and the
    stream a middlebox has thinned to its lower sub-layers"""
    from test_random_access import layered, tid_of
    aus = layered(seed, n=28, w=416, h=240, slices=(0, 1, 3)[seed % 3], hidden_pics=(0, 10)[seed & 1])
    tids = [tid_of(a) for a in aus]
    both(aus, range(len(aus)), threads, threads > 1)
    for keep in range(max(tids)):
        k = [i for i in range(len(aus)) if tids[i] <= keep]
        shown = both([aus[i] for i in k], k, threads, threads > 1)
        assert shown and set(shown) <= set(k)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 13))
def test_optional_vui_parts_and_where_the_picture_rate_is_said(gpu, seed):
    """x265 fills the VUI (aspect ratio, video signal type, bitstream restriction ...); the picture rate may be in the VUI, in the VPS, in both or nowhere: the rate
    libOpenHevcGetPictureInfo reports (OpenHEVCFilter takes it for the frame's rate, openhevcfilter.cpp:183-200) is the checker's"""
    from kvazzup_amd.codec import Decoder
    g = orc.OracleGen(208, 144, seed=seed, vui_extras=1, intra_period=4, temporal_layers=seed & 1, gop=(0, 4)[seed & 1])
    aus = [g.picture() for _ in range(6)]
    g.close()
    od, gd = orc.OracleDecoder(), Decoder()
    want, got = [], []
    try:
        for t, au in enumerate(aus):
            want += od.decode_au(au, t)
            got += gd.decode_au(au, t)
        want += od.flush()
        got += gd.drain()
    finally:
        gd.close()
        od.close()
    assert len(got) == len(want) == 6
    for a, b in zip(got, want):
        assert np.array_equal(a["i420"], b["i420"])
        assert tuple(a["fps"]) == tuple(b["fps"]) or (b["fps"] == (0, 0) and a["fps"][0] == 0), (a["fps"], b["fps"])


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("seed", range(1, 13))
def test_every_way_to_write_a_reference_picture_set(gpu, seed, threads):
    """candidate sets in the SPS (explicit or predicted), slices that name one, predict from one or write their own (7.3.7): HM-style streams"""
    kw = dict(rps_forms=1, num_refs=1 + seed % 4, intra_period=16, tmvp=1)
    if seed & 1:
        kw.update(gop=(2, 4, 8)[seed % 3], b_slices=50, open_gop=(seed >> 1) & 1, temporal_layers=(seed >> 2) & 1)
    else:
        kw.update(long_term=(seed >> 1) & 1)
    g = orc.OracleGen(416, 240, seed=seed, **kw)
    aus = [g.picture() for _ in range(24)]
    g.close()
    assert len(both(aus, range(len(aus)), threads, threads > 1)) == len(aus)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("seed", range(1, 13))
def test_what_a_decoder_steps_over(gpu, seed, threads):
    """slice_reserved_flag bits, slice segment header extension bytes, SPS / PPS extension data, access unit delimiters, unknown SEI messages, filler data -- also between
    the pictures of free slices, which the next access unit's first NAL unit closes (an access unit delimiter or a prefix SEI message does, filler data does not)"""
    kw = dict(hdr_extras=1, intra_period=6, slices=(0, 1, 3, 2)[seed % 4], tmvp=1)
    if seed % 4 == 3:
        kw.update(tile_rows=2, tile_cols=2, wpp=0)
    g = orc.OracleGen(416, 240, seed=seed, **kw)
    aus = [g.picture() for _ in range(14)]
    g.close()
    assert len(both(aus, range(len(aus)), threads, threads > 1)) == len(aus)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("seed", range(1, 7))
def test_the_highest_sub_layer_to_decode_can_be_set(gpu, seed, threads):
    """libOpenHevcSetTemporalLayer_id(h, k): the WHOLE stream goes in, the slice NAL units of the sub-layers above k are dropped inside -- the pictures are those of the
    stream thinned by hand.  (OpenHEVC's default is 7; uvgComm's filter passes 0, openhevcfilter.cpp:54 -- a peer's gop=8 stream plays at its base layer's rate there.)"""
    from kvazzup_amd.codec import Decoder
    from test_random_access import layered, tid_of
    aus = layered(seed, n=28, w=416, h=240)
    tids = [tid_of(a) for a in aus]
    for keep in range(max(tids) + 1):
        od = orc.OracleDecoder()
        gd = Decoder(threads=threads, frame_threads=threads > 1, temporal_layer=keep)
        want, got = [], []
        try:
            for t, au in enumerate(aus):
                if tids[t] <= keep:
                    want += od.decode_au(au, t)
                got += gd.decode_au(au, t)
            want += od.flush()
            got += gd.drain()
        finally:
            gd.close()
            od.close()
        assert [f["pts"] for f in got] == [f["pts"] for f in want] and len(got) == sum(t <= keep for t in tids)
        for a, b in zip(got, want):
            assert np.array_equal(a["i420"], b["i420"]), (keep, a["pts"])
