"""Random access into a stream, CPU side: CRA pictures with RASL / RADL leading pictures, BLA pictures, end of sequence NAL units,
pic_output_flag (8.1.3, 8.3.1, 8.3.2, C.5.2.2) -- what a decoder meets when it joins a stream at a random access point that is no
IDR picture, or reads one a splicer has cut (the reference hands every NAL unit it receives to libOpenHevcDecode,
/root/reference/src/media/processing/openhevcfilter.cpp:134-172; a Kvazaar peer sends IDR pictures only, other senders do not).

The checker's decoder against PROPERTIES of the streams -- a decode that starts at a CRA picture shows exactly the full decode's
pictures from there on minus that picture's RASL pictures, bit for bit; a CRA picture renamed BLA, or behind an end of sequence
NAL unit, does the same to the pictures that follow while everything before it still comes out -- and against the second,
independently written decoder (tests/pyhevc.py).  The product's decoder meets the same streams in tests/test_gpu_random_access.py.

Further down: stream-level syntax the decoders had parsers for and no stream of -- temporal sub-layers (and the stream thinned to its lower layers), the VUI's optional
parts, every way to write a reference picture set (inter RPS prediction), and what a decoder steps over (reserved header bits, header extension bytes, parameter set
extension data, access unit delimiters, unknown SEI messages, filler data)."""
import numpy as np
import pytest

import orc
import pyhevc
from test_python_decoder import tabs

EOS = bytes([0, 0, 0, 1, 36 << 1, 1])


def nal_type(nal):
    i = 0
    while nal[i] == 0:
        i += 1
    return (nal[i + 1] >> 1) & 63


def vcl_type(au):
    return next(t for t in (nal_type(n) for n in orc.split_nals(au)) if t < 32)


def rename(au, old, new):
    """the access unit with its slice NAL units of type `old` called `new` (the type sits in bits 1..6 of the header's first byte)"""
    out = bytearray()
    for n in orc.split_nals(au):
        n = bytearray(n)
        i = 0
        while n[i] == 0:
            i += 1
        if (n[i + 1] >> 1) & 63 == old:
            n[i + 1] = (n[i + 1] & 0x81) | (new << 1)
        out += n
    return bytes(out)


def stream(seed, n=36, **kw):
    cfg = dict(gop=(2, 4, 8)[seed % 3], open_gop=1, intra_period=48, b_slices=50, num_refs=1 + seed % 4)
    cfg.update(kw)
    g = orc.OracleGen(64, 64, seed=seed, **cfg)
    aus = [g.picture() for _ in range(n)]
    g.close()
    return aus


def oracle_pictures(aus, first_pts=0):
    """(pts, picture) in output order; the pts is the access unit's index"""
    d = orc.OracleDecoder()
    out = []
    for k, au in enumerate(aus):
        out += d.decode_au(au, pts=first_pts + k)
    out += d.flush()
    d.close()
    return [(f["pts"], f["i420"]) for f in out]


def rasl_of(types, k):
    """indices of the RASL pictures that belong to the CRA picture at index k"""
    out = []
    for i in range(k + 1, len(types)):
        if 16 <= types[i] <= 23:
            break
        if types[i] in (8, 9):
            out.append(i)
    return out


def same(a, b):
    assert [p for p, _ in a] == [p for p, _ in b]
    for (p, x), (_, y) in zip(a, b):
        assert np.array_equal(x, y), p


@pytest.mark.parametrize("seed", (1, 2, 3, 4, 6, 7, 9, 10, 11, 12, 13))
def test_decoding_from_a_cra_picture_drops_its_rasl_pictures_and_nothing_else(seed):
    aus = stream(seed)
    types = [vcl_type(a) for a in aus]
    cras = [i for i, t in enumerate(types) if t == 21]
    assert cras
    full = oracle_pictures(aus)
    assert len(full) == len(aus)
    for k in cras:
        drop = set(rasl_of(types, k))
        want = [(p, x) for p, x in full if p >= k and p not in drop]
        same(oracle_pictures(aus[k:], first_pts=k), want)


@pytest.mark.parametrize("bla", (16, 17, 18))
@pytest.mark.parametrize("seed", (1, 2, 3, 6))
def test_a_cra_picture_renamed_bla(seed, bla):
    """what a splicer does (BLA_W_LP as it is; BLA_W_RADL with the RASL pictures removed; BLA_N_LP with every leading picture removed): the pictures before it come
    out untouched, the ones behind it as a decode that starts there gives them"""
    aus = stream(seed)
    types = [vcl_type(a) for a in aus]
    full = oracle_pictures(aus)
    for k in [i for i, t in enumerate(types) if t == 21][:3]:
        drop = set(rasl_of(types, k))
        if bla == 18:
            drop |= {i for i in range(k + 1, len(types)) if types[i] in (6, 7) and all(not 16 <= t <= 23 for t in types[k + 1:i])}
        cut = [rename(a, 21, bla) if i == k else a for i, a in enumerate(aus) if bla == 16 or i not in drop]
        pts = [i for i in range(len(aus)) if bla == 16 or i not in drop]
        d = orc.OracleDecoder()
        got = []
        for p, au in zip(pts, cut):
            got += d.decode_au(au, pts=p)
        got += d.flush()
        d.close()
        want = [(p, x) for p, x in full if p not in drop]
        # (the sequence the BLA picture starts follows ALL of the one before it in output order: what was decoded before the cut precedes what came behind it)
        want = [e for e in want if e[0] < k] + [e for e in want if e[0] >= k]
        same([(f["pts"], f["i420"]) for f in got], want)


@pytest.mark.parametrize("seed", (1, 4, 10))
def test_an_end_of_sequence_nal_unit_makes_the_next_cra_picture_a_starting_point(seed):
    aus = stream(seed)
    types = [vcl_type(a) for a in aus]
    full = oracle_pictures(aus)
    for k in [i for i, t in enumerate(types) if t == 21][:3]:
        drop = set(rasl_of(types, k))
        got = oracle_pictures(aus[:k] + [EOS + aus[k]] + aus[k + 1:])
        want = [(p, x) for p, x in full if p not in drop]
        want = [e for e in want if e[0] < k] + [e for e in want if e[0] >= k]
        same(got, want)


@pytest.mark.parametrize("seed", (2, 3, 7))
def test_pictures_with_pic_output_flag_zero_are_referenced_but_not_shown(seed):
    shown = oracle_pictures(stream(seed, n=30, hidden_pics=35))
    assert 10 < len(shown) < 30
    # without reordering: the shown pictures are in decoding order, the others leave a gap in the time stamps
    g = orc.OracleGen(64, 64, seed=seed, hidden_pics=35, intra_period=10)
    aus = [g.picture() for _ in range(30)]
    g.close()
    low_delay = oracle_pictures(aus)
    pts = [p for p, _ in low_delay]
    assert pts == sorted(pts) and 10 < len(pts) < 30


def python_pictures(aus):
    d = pyhevc.Decoder(tabs())
    for au in aus:
        for nal in pyhevc.split_nals(au):
            d.decode_nal(nal)
    return d.flush()


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"hidden_pics": 30}), (3, {"wpp": 1, "slices": 1}), (6, {"tmvp": 1, "hidden_pics": 20})])
def test_the_python_decoder_agrees(seed, kw):
    aus = stream(seed, n=22, **kw)
    types = [vcl_type(a) for a in aus]
    k = next(i for i, t in enumerate(types) if t == 21)
    cases = [aus, aus[k:], [rename(a, 21, 16) if i == k else a for i, a in enumerate(aus)], aus[:k] + [EOS + aus[k]] + aus[k + 1:]]
    for c in cases:
        want = [x for _, x in oracle_pictures(c)]
        got = python_pictures(c)
        assert len(want) == len(got) > 0
        for a, b in zip(want, got):
            assert np.array_equal(a, b["i420"])


def discard_prior(au):
    """the access unit with no_output_of_prior_pics_flag = 1 in its IDR / BLA slice NAL units (the second bit of the slice segment header)"""
    out = bytearray()
    for n in orc.split_nals(au):
        n = bytearray(n)
        i = 0
        while n[i] == 0:
            i += 1
        if 16 <= (n[i + 1] >> 1) & 63 <= 20:
            n[i + 3] |= 0x40
        out += n
    return bytes(out)


@pytest.mark.parametrize("seed", (1, 2, 3, 4, 5, 6))
def test_no_output_of_prior_pics_flag_discards_what_still_waits(seed):
    """C.5.2.2: an IDR or BLA picture with the flag empties the buffer without output -- the pictures of the sequence before that had not had their turn yet (at most
    sps_max_num_reorder_pics of them, the last ones in output order) are never seen; a CRA picture ignores the flag"""
    aus = stream(seed, n=30, intra_period=7 + seed % 3, open_gop=seed & 1)
    types = [vcl_type(a) for a in aus]
    idrs = [i for i, t in enumerate(types) if t == 19 and i > 0]
    assert idrs
    full = oracle_pictures(aus)
    got = oracle_pictures([discard_prior(a) for a in aus])
    kept = [p for p, _ in got]
    assert len(kept) < len(full) and [p for p, _ in full if p in kept] == kept            # (the same order, some pictures missing)
    lg = {2: 1, 4: 2, 8: 3}[(2, 4, 8)[seed % 3]]
    for a, b in zip([0] + idrs, idrs + [len(aus)]):
        lost = [p for p, _ in full if a <= p < b and p not in kept]
        assert len(lost) <= (lg if b < len(aus) else 0)                                   # (nothing is lost at the end of the stream)
        tail = [p for p, _ in full if a <= p < b][len([p for p, _ in full if a <= p < b]) - len(lost):]
        assert lost == tail                                                               # (the sequence's last pictures in output order)
    same(got, [e for e in full if e[0] in kept])
    want = [x for _, x in got]
    py = python_pictures([discard_prior(a) for a in aus])
    assert len(py) == len(want)
    for x, y in zip(want, py):
        assert np.array_equal(x, y["i420"])


@pytest.mark.parametrize("seed", (1, 2, 3, 6))
def test_a_bla_picture_with_no_output_of_prior_pics_flag(seed):
    aus = stream(seed, n=30)
    types = [vcl_type(a) for a in aus]
    for k in [i for i, t in enumerate(types) if t == 21][1:3]:
        cut = [discard_prior(rename(a, 21, 16)) if i == k else a for i, a in enumerate(aus)]
        plain = oracle_pictures([rename(a, 21, 16) if i == k else a for i, a in enumerate(aus)])
        got = oracle_pictures(cut)
        kept = [p for p, _ in got]
        assert [p for p, _ in plain if p in kept] == kept and all(p < k for p, _ in plain if p not in kept)
        same(got, [e for e in plain if e[0] in kept])
        py = python_pictures(cut)
        assert len(py) == len(got)
        for (_, x), y in zip(got, py):
            assert np.array_equal(x, y["i420"])


def tid_of(au):
    """TemporalId of the access unit's slice NAL units"""
    for n in orc.split_nals(au):
        i = 0
        while n[i] == 0:
            i += 1
        if (n[i + 1] >> 1) & 63 < 32:
            return (n[i + 2] & 7) - 1
    raise ValueError("no slice")


def layered(seed, n=26, w=64, h=64, **kw):
    cfg = dict(gop=(2, 4, 8)[seed % 3], temporal_layers=1, open_gop=seed & 1, intra_period=24, b_slices=50, num_refs=1 + seed % 4, tmvp=1)
    cfg.update(kw)
    g = orc.OracleGen(w, h, seed=seed, **cfg)
    aus = [g.picture() for _ in range(n)]
    g.close()
    return aus


@pytest.mark.parametrize("seed", range(1, 9))
def test_temporal_sub_layers(seed):
    """TemporalId in the NAL unit headers, parameter sets with sub-layers (profile_tier_level's sub-layer part, ordering info for all sub-layers or the highest),
    sub-layer non-reference pictures: both decoders agree; and -- the point of sub-layers -- the stream without its upper layers, from any layer up, decodes to the
    same pictures minus the ones taken out (8.3.1: the picture order count follows the TemporalId 0 pictures alone)"""
    aus = layered(seed)
    tids = [tid_of(a) for a in aus]
    top = max(tids)
    assert top == {2: 1, 4: 2, 8: 3}[(2, 4, 8)[seed % 3]] and tids[0] == 0
    full = oracle_pictures(aus)
    assert len(full) == len(aus)
    py = python_pictures(aus)
    for (_, x), y in zip(full, py):
        assert np.array_equal(x, y["i420"])
    for keep in range(top):                      # sub-layers 0 .. keep stay
        d = orc.OracleDecoder()
        got = []
        for k, au in enumerate(aus):
            if tids[k] <= keep:
                got += d.decode_au(au, pts=k)
        got += d.flush()
        d.close()
        same([(f["pts"], f["i420"]) for f in got], [e for e in full if tids[e[0]] <= keep])
    thin = python_pictures([a for k, a in enumerate(aus) if tids[k] < top])
    want = [x for p, x in full if tids[p] < top]
    assert len(thin) == len(want)
    for x, y in zip(want, thin):
        assert np.array_equal(x, y["i420"])


def test_the_products_parser_reads_sub_layer_syntax():
    """(CPU: the host half alone through the parse-only hook -- every picture parsed, with and without the upper sub-layers)"""
    import parser_probe as PP
    for seed in (1, 2, 3):
        aus = layered(seed, w=128, h=64)
        tids = [tid_of(a) for a in aus]
        for keep in range(max(tids) + 1):
            nals = [n for k, a in enumerate(aus) if tids[k] <= keep for n in orc.split_nals(a)]
            for threads in (0, 3):
                assert PP.probe(nals, threads)["pictures"] == sum(t <= keep for t in tids)


@pytest.mark.parametrize("seed", range(1, 9))
def test_optional_vui_parts_and_where_the_picture_rate_is_said(seed):
    """the VUI's optional parts in front of the timing information (extended aspect ratio, overscan, video signal type, chroma location, default display window), POC
    proportional timing and the bitstream restriction behind it; the rate in the VUI, the VPS, both or nowhere: the decoders skip what they do not use and agree"""
    g = orc.OracleGen(64, 64, seed=seed, vui_extras=1, intra_period=4, temporal_layers=seed & 1, gop=(0, 4)[seed & 1])
    aus = [g.picture() for _ in range(6)]
    g.close()
    d = orc.OracleDecoder()
    fr = []
    for a in aus:
        fr += d.decode_au(a)
    fr += d.flush()
    d.close()
    assert len(fr) == 6 and len({f["fps"] for f in fr}) == 1
    py = python_pictures(aus)
    assert len(py) == 6
    for a, b in zip(fr, py):
        assert np.array_equal(a["i420"], b["i420"])
    import parser_probe as PP
    assert PP.probe([n for a in aus for n in orc.split_nals(a)], 0)["pictures"] == 6


@pytest.mark.parametrize("seed", range(1, 13))
def test_every_way_to_write_a_reference_picture_set(seed):
    """7.3.7: candidate sets in the SPS (explicit, or predicted from the set before), slices that name a candidate, slices that predict their set from a candidate
    (delta_idx_minus1, delta_rps, used_by_curr_pic_flag / use_delta_flag), slices that write it out -- in low-delay streams and in reordered ones, where the sets
    have entries on both sides.  Three parsers written from the standard's text (the checker, the Python decoder, the product through the parse-only hook) read what
    the synthesiser's writer -- the derivation run backwards -- produced, and the pictures agree"""
    import parser_probe as PP
    kw = dict(rps_forms=1, num_refs=1 + seed % 4, intra_period=16)
    if seed & 1:
        kw.update(gop=(2, 4, 8)[seed % 3], b_slices=50, open_gop=(seed >> 1) & 1, temporal_layers=(seed >> 2) & 1)
    g = orc.OracleGen(64, 64, seed=seed, **kw)
    aus = [g.picture() for _ in range(20)]
    g.close()
    before = dict(pyhevc.RPS_FORMS)
    want = oracle_pictures(aus)
    got = python_pictures(aus)
    assert len(want) == len(got) == len(aus)
    for (_, x), y in zip(want, got):
        assert np.array_equal(x, y["i420"])
    seen = {k: pyhevc.RPS_FORMS[k] - before[k] for k in before}
    assert seen["inter_sps"] + seen["explicit_sps"] >= 2 and seen["inter_slice"] + seen["named"] > 0, seen
    assert PP.probe([n for a in aus for n in orc.split_nals(a)], 0)["pictures"] == len(aus)


@pytest.mark.parametrize("seed", range(1, 13))
def test_what_a_decoder_steps_over(seed):
    """slice_reserved_flag bits, slice segment header extension bytes (in dependent segments too), SPS / PPS extension data, access unit delimiters, SEI messages no
    decoder knows in front of and behind the picture, filler data: nothing of it changes a sample"""
    import parser_probe as PP
    kw = dict(hdr_extras=1, intra_period=6, slices=(0, 1, 3, 0)[seed % 4], tile_rows=1, tile_cols=1)
    if seed % 4 == 3:
        kw.update(tile_rows=2, tile_cols=2, wpp=0)             # (the Python decoder reads a picture with tiles as one slice)
    g = orc.OracleGen(128, 128, seed=seed, **kw)
    aus = [g.picture() for _ in range(8)]
    g.close()
    types = {nal_type(n) for a in aus for n in orc.split_nals(a)}
    assert {35, 38, 39, 40} <= types
    want = oracle_pictures(aus)
    got = python_pictures(aus)
    assert len(want) == len(got) == len(aus)
    for (_, x), y in zip(want, got):
        assert np.array_equal(x, y["i420"])
    # the same pictures as without the extras? (the extras draw from the same random sequence, so the streams differ; what can be said: stripping the NAL units
    # a decoder ignores changes nothing)
    bare = [b"".join(n for n in orc.split_nals(a) if nal_type(n) not in (35, 38, 39, 40)) for a in aus]
    same(oracle_pictures(bare), want)
    for threads in (0, 3):
        assert PP.probe([n for a in aus for n in orc.split_nals(a)], threads)["pictures"] == len(aus)
