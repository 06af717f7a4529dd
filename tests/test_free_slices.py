"""CPU: pictures of FREE slices (round 6) -- slice segments that begin at any coding tree block, independent slices and dependent segments mixed, with and without
WPP (oracle/hevc_gen.c, slices = 3).  The synthesiser and the checker's decoder agree on every stream (a wrong context state or availability anywhere ends in a
parse error or a wrong picture count); the PRODUCT's parser (parse-only hook: csrc/decoder.hip's assembly of a picture from its segments, the take-back of a picture
whose first segment only looked whole, its slice-data parser) accepts the same streams and produces the same records with one row thread and with four.  The
pictures themselves are compared on the GPU box (tests/test_gpu_slices.py) and by the second decoder (tests/test_python_decoder.py::test_free_slices)."""
import pytest

import orc
import parser_probe as PP


def stream(seed, wpp, ctb_log2, n=5, w=192, h=128, **kw):
    g = orc.OracleGen(w, h, seed=seed, slices=3, wpp=wpp, ctb_log2=ctb_log2, **kw)
    aus = [g.picture() for _ in range(n)]
    g.close()
    return aus


@pytest.mark.parametrize("ctb_log2", [6, 4])
@pytest.mark.parametrize("wpp", [0, 1])
def test_checker_decodes_what_the_synthesiser_writes(wpp, ctb_log2):
    segments = 0
    for seed in range(1, 25):
        aus = stream(seed, wpp, ctb_log2)
        od = orc.OracleDecoder()
        n = sum(len(od.decode_au(au, t)) for t, au in enumerate(aus)) + len(od.flush())
        od.close()
        assert n == len(aus), seed
        segments += sum(1 for au in aus for nal in orc.split_nals(au) if ((nal[4] >> 1) & 63) < 32)
    assert segments > 24 * 5 * 2                                   # (more than two segments per picture on average)


@pytest.mark.parametrize("ctb_log2", [6, 5, 4])
@pytest.mark.parametrize("wpp", [0, 1])
def test_product_parser_accepts_them_with_one_and_four_row_threads(wpp, ctb_log2):
    for seed in range(1, 13):
        nals = [n for au in stream(seed, wpp, ctb_log2) for n in orc.split_nals(au)]
        a, b = PP.probe(nals, 1), PP.probe(nals, 4)
        assert a == b and a["pictures"] == 5, (seed, a, b)


def test_a_stream_that_turns_to_free_slices_and_back():
    """one decoder: whole pictures, then free slices (the first such picture is taken back when its first segment -- submitted as the whole picture -- ends early),
    Kvazaar's dependent segment per CTU row, whole pictures again: every picture is parsed, none twice"""
    nals = []
    for k, slices in enumerate((0, 3, 1, 3, 0)):
        g = orc.OracleGen(192, 128, seed=60 + k, slices=slices, wpp=k & 1, intra_period=4)
        for _ in range(4):
            nals += list(orc.split_nals(g.picture()))
        g.close()
    assert PP.probe(nals, 1)["pictures"] == 20 and PP.probe(nals, 3)["pictures"] == 20


def test_a_lost_segment_is_an_error_not_a_wrong_picture():
    """the second of a picture's segments removed: where the first ends is then where the THIRD begins -- the parser finds end_of_slice_segment_flag elsewhere"""
    import ctypes as C
    lib = PP._lib()
    for seed in range(1, 30):
        aus = stream(seed, 0, 6, n=1)
        nals = list(orc.split_nals(aus[0]))
        vcl = [i for i, n in enumerate(nals) if ((n[4] >> 1) & 63) < 32]
        if len(vcl) < 3:
            continue
        del nals[vcl[1]]
        h = lib.libOpenHevcInit(1, 2)
        assert lib.kvzx_decoder_set_parse_only(h, 1) == 1 and lib.libOpenHevcStartDecoder(h) == 0
        rcs = [lib.libOpenHevcDecode(h, n, len(n), 0) for n in nals + [bytes([0, 0, 0, 1, 36 << 1, 1])]]
        out = (C.c_uint64 * 5)()
        lib.kvzx_decoder_parse_probe_stats(h, out, None)
        lib.libOpenHevcClose(h)
        assert int(out[0]) == 0 and lib is not None, (seed, rcs)      # no picture was booked
        return
    pytest.skip("no picture with three segments among the seeds")


@pytest.mark.parametrize("min_cb,ctb,w,h", [(4, 6, 208, 144), (4, 4, 208, 144), (5, 6, 192, 128), (5, 5, 192, 128)])
def test_minimum_coding_blocks_of_16_and_32_samples(min_cb, ctb, w, h):
    """MinCbLog2SizeY 4 / 5 (round 6): the synthesiser, the checker's decoder and the product's parser (one row thread and four) on the same streams"""
    for seed in range(1, 11):
        g = orc.OracleGen(w, h, seed=seed, min_cb_log2=min_cb, ctb_log2=ctb, slices=3 if seed % 3 == 0 else 0, all_part_modes=1)
        assert g.config["min_cb_log2"] == min_cb
        aus = [g.picture() for _ in range(4)]
        g.close()
        od = orc.OracleDecoder()
        assert sum(len(od.decode_au(au, t)) for t, au in enumerate(aus)) + len(od.flush()) == 4, seed
        od.close()
        nals = [n for au in aus for n in orc.split_nals(au)]
        a, b = PP.probe(nals, 1), PP.probe(nals, 4)
        assert a == b and a["pictures"] == 4, (seed, a, b)


@pytest.mark.parametrize("layout", [dict(slices=3, wpp=0), dict(slices=3, wpp=1), dict(slices=2, tile_rows=2, tile_cols=2, wpp=0), dict(slices=0, tile_rows=3, tile_cols=2, wpp=1), dict(slices=2, tile_rows=2, tile_cols=1)])
def test_boundaries_closed_to_the_loop_filters(layout):
    """loop_filter_across_tiles_enabled_flag / slice_loop_filter_across_slices_enabled_flag drawn (lf_across = 1) or all off (2): the synthesiser, the checker's decoder
    and the product's parser (which takes the edge marks off the records along closed boundaries and lays out the neighbour map) on the same streams; the flags change
    what the checker decodes"""
    import numpy as np
    changed = 0
    for seed in range(1, 9):
        pics = {}
        for lf in (0, 1, 2):
            g = orc.OracleGen(256, 192, seed=seed, lf_across=lf, sao=1, **layout)
            aus = [g.picture() for _ in range(3)]
            g.close()
            od = orc.OracleDecoder()
            pics[lf] = [f["i420"] for t, au in enumerate(aus) for f in od.decode_au(au, t)] + [f["i420"] for f in od.flush()]
            od.close()
            assert len(pics[lf]) == 3
            nals = [n for au in aus for n in orc.split_nals(au)]
            a, b = PP.probe(nals, 1), PP.probe(nals, 3)
            assert a == b and a["pictures"] == 3, (seed, lf, a, b)
        changed += any(not np.array_equal(x, y) for x, y in zip(pics[0], pics[2]))
    assert changed >= 6


@pytest.mark.parametrize("kw", [dict(wpp=0), dict(wpp=1), dict(wpp=1, slices=3), dict(wpp=0, tile_rows=2, tile_cols=2, slices=0), dict(intra_period=1, ctb_log2=4)])
def test_pcm_coding_units(kw):
    """PCM units (round 6): the synthesiser, the checker's decoder and the product's parser (which restarts its arithmetic decoder behind the samples and hands them to
    the kernels as the levels of a transquant-bypass block over no prediction) on the same streams, one row thread and four"""
    for seed in range(1, 13):
        g = orc.OracleGen(192, 128, seed=seed, pcm=30, intra_in_p=40, **kw)
        aus = [g.picture() for _ in range(4)]
        g.close()
        od = orc.OracleDecoder()
        assert sum(len(od.decode_au(au, t)) for t, au in enumerate(aus)) + len(od.flush()) == 4, seed
        od.close()
        nals = [n for au in aus for n in orc.split_nals(au)]
        a, b = PP.probe(nals, 1), PP.probe(nals, 4)
        assert a == b and a["pictures"] == 4, (seed, a, b)


@pytest.mark.parametrize("kw", [dict(num_refs=1), dict(num_refs=3, tmvp=1), dict(num_refs=2, tmvp=1, list_mod=50), dict(num_refs=4, tmvp=1, wpp=0)])
def test_long_term_reference_pictures(kw):
    """long-term reference pictures (round 6): the synthesiser, the checker's decoder and the product's parser on the same streams (22 pictures: the POC's LSBs wrap
    when the stream drew four of them), one row thread and four"""
    for seed in range(1, 11):
        g = orc.OracleGen(192, 128, seed=seed, long_term=1, intra_period=32, **kw)
        aus = [g.picture() for _ in range(22)]
        g.close()
        od = orc.OracleDecoder()
        assert sum(len(od.decode_au(au, t)) for t, au in enumerate(aus)) + len(od.flush()) == 22, seed
        od.close()
        nals = [n for au in aus for n in orc.split_nals(au)]
        a, b = PP.probe(nals, 1), PP.probe(nals, 4)
        assert a == b and a["pictures"] == 22, (seed, a, b)


def test_a_segment_at_a_rows_start_behind_a_whole_picture():
    """(soak seed 1578) A stream of free slices without WPP: after a picture that came in ONE segment the synchronous decoder takes the stream for whole pictures again;
    the next picture's second segment happens to begin exactly where a CTB row does and reaches to the picture's end -- counted as one row (Kvazaar's form), the
    picture looked incomplete and was dropped.  It is read as a picture of free slices when its access unit ends.  The product's host half through the parse-only
    hook: every picture parsed."""
    import parser_probe as PP
    g = orc.OracleGen(64, 64, seed=1578, intra_period=16, tmvp=1, ctb_log2=4, min_cb_log2=3, cip=0, pcm=0, lf_across=2, intra_in_p=30, slices=3, gop=4, b_slices=0, open_gop=0,
                      temporal_layers=1, rps_forms=1)
    aus = [g.picture() for _ in range(24)]
    g.close()
    nals = [n for a in aus for n in orc.split_nals(a)]
    for threads in (0, 3):
        assert PP.probe(nals, threads)["pictures"] == 24
