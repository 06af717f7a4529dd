"""GPU parity for streams this project's own encoder never writes: what libOpenHevcDecode gets from a foreign peer
(/root/reference/src/media/processing/openhevcfilter.cpp:134-172 feeds arbitrary NAL units; the peer normally runs Kvazaar with
gop=lp-g4d3t1, kvazaarfilter.cpp:233).  The streams come from the conformance-style synthesiser oracle/hevc_gen.c -- random but
valid Main-profile syntax -- and the HIP decoder must reproduce the general CPU decoder (oracle/hevc_dec.c) bit for bit."""
import numpy as np
import pytest

import orc


def run_stream(w, h, pictures, threads=1, frame_threads=False, **cfg):
    from kvazzup_amd.codec import Decoder
    g = orc.OracleGen(w, h, **cfg)
    od = orc.OracleDecoder()
    gd = Decoder(threads=threads, frame_threads=frame_threads) if frame_threads else Decoder()
    refs, got, pocs = [], [], []
    try:
        reorder = g.config.get("gop", 0) > 1              # (pictures come out in POC order, later than they go in)
        for t in range(pictures):
            au = g.picture()
            r = od.decode_au(au, t)
            assert reorder or len(r) == 1, (t, g.config)
            refs += [f["i420"] for f in r]
            pocs += [f["poc"] for f in r]
            got += gd.decode_au(au, t)
        refs += [f["i420"] for f in od.flush()]
        if frame_threads or reorder or g.config.get("slices") == 3:      # (free slices: a picture is closed by what follows it in the stream)
            got += gd.drain()
        assert len(got) == pictures and len(refs) == pictures, (len(got), len(refs), g.config)
        for t in range(pictures):
            assert got[t]["width"] == w and got[t]["height"] == h
            if not np.array_equal(got[t]["i420"], refs[t]):
                d = np.flatnonzero(got[t]["i420"] != refs[t])
                plane = "Y" if d[0] < w * h else "C"
                pytest.fail("picture %d: %d samples differ, first at %d (%s, x=%d y=%d); config %r"
                            % (t, len(d), d[0], plane, d[0] % w, d[0] // w, g.config))
    finally:
        gd.close()
        od.close()
        g.close()


PLAIN = dict(num_refs=1, tmvp=0, amp=0, sao=0, strong_intra=1, sign_hiding=0, transform_skip=0, cabac_init=0, wpp=1, tile_rows=1, uniform_tiles=1,
             th_depth_inter=0, th_depth_intra=0, qp_delta=0, chroma_qp_offsets=0, deblock_mode=0, par_mrg_level=2, intra_in_p=0, all_part_modes=0,
             chroma_modes=0, nxn_intra=0, max_cu_log2=5, min_cu_log2=3, big_mvd=0)


@pytest.mark.gpu
@pytest.mark.parametrize("feature", [
    dict(),                                             # all CU sizes 8..32, fractional vectors, merge / AMVP, nothing else
    dict(max_cu_log2=6),                                # 64x64 CUs: four 32x32 transform blocks
    dict(intra_in_p=30),                                # intra CUs in P pictures
    dict(intra_in_p=30, nxn_intra=1),                   # 4x4 luma intra blocks (DST), chroma at the parent
    dict(intra_in_p=30, chroma_modes=1),                # intra_chroma_pred_mode 0..3 (and mode 34)
    dict(strong_intra=0, intra_in_p=40),
    dict(all_part_modes=1),                             # 2NxN, Nx2N
    dict(all_part_modes=1, amp=1),                      # asymmetric partitions
    dict(th_depth_inter=2, th_depth_intra=2, intra_in_p=20),       # transform trees
    dict(th_depth_inter=1, all_part_modes=1),
    dict(num_refs=3),                                   # several reference pictures, list longer than the set
    dict(tmvp=1),                                       # temporal candidates
    dict(num_refs=4, tmvp=1, par_mrg_level=4),
    dict(sign_hiding=1),
    dict(transform_skip=1, th_depth_inter=2, th_depth_intra=2, nxn_intra=1, intra_in_p=20),
    dict(qp_delta=1), dict(qp_delta=3), dict(qp_delta=4, chroma_qp_offsets=1),
    dict(deblock_mode=1), dict(deblock_mode=2), dict(deblock_mode=3),
    dict(sao=1),
    dict(cabac_init=1),
    dict(wpp=0), dict(wpp=0, tile_rows=3), dict(wpp=1, tile_rows=2, uniform_tiles=0),
    dict(big_mvd=1),
    # a peer whose uvgComm has the "scaling list" / "lossless" boxes ticked (kvazaarfilter.cpp:235-244), and what else those syntax elements allow
    dict(scaling_lists=1),                              # scaling_list_enabled_flag, default lists (Kvazaar `scaling-list default`)
    dict(scaling_lists=1, max_cu_log2=6, intra_in_p=30, nxn_intra=1, th_depth_inter=2, th_depth_intra=2),      # ... every block size, intra and inter matrices
    dict(scaling_lists=2, intra_in_p=25, th_depth_inter=2, th_depth_intra=2, transform_skip=1, nxn_intra=1),   # lists in the SPS; 4x4 transform-skip blocks are scaled
    dict(scaling_lists=3, max_cu_log2=6, intra_in_p=25),                                                      # default in the SPS, the PPS's lists
    dict(scaling_lists=4, qp_delta=2, chroma_qp_offsets=1, th_depth_inter=1),
    dict(tq_bypass=35, intra_in_p=30, th_depth_inter=2, th_depth_intra=2, nxn_intra=1),                        # cu_transquant_bypass_flag on some coding units
    dict(tq_bypass=50, sao=1, sign_hiding=1, transform_skip=1, intra_in_p=30, max_cu_log2=6),                  # ... beside SAO, sign hiding, transform skip: the loop filters keep out
    dict(tq_bypass=100, sao=1, intra_in_p=20),                                                                # a lossless stream (Kvazaar `lossless`)
    dict(tq_bypass=30, scaling_lists=4, deblock_mode=2, qp_delta=3),
])
def test_feature_matches_oracle(gpu, feature):
    """one tool at a time on top of a plain stream; 416x240 (partial CTUs on both axes), 6 pictures"""
    cfg = dict(PLAIN); cfg.update(feature)
    run_stream(416, 240, 6, seed=7, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 25))
def test_random_streams_match_oracle(gpu, seed):
    """every switch drawn from the seed"""
    sizes = [(416, 240), (352, 288), (200, 136), (64, 64), (24, 16), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 8, seed=seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 112))
def test_random_streams_with_scaling_lists_and_transquant_bypass(gpu, seed):
    """every other switch drawn from the seed; scaling lists in all five forms, no / some / all coding units bypassing transform and quantisation"""
    sizes = [(416, 240), (352, 288), (200, 136), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 6, seed=seed, scaling_lists=seed % 5, tq_bypass=(0, 25, 100)[seed % 3], threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(scaling_lists=1), dict(tq_bypass=100), dict(scaling_lists=3, tq_bypass=15, sao=1)])
def test_kvazaar_shaped_stream_1080p_scaling_list_and_lossless(gpu, kw):
    """1080p in the shape a Kvazaar peer sends, with uvgComm's "scaling list" box ticked (`scaling-list default`), with `lossless`, and with both tools mixed"""
    cfg = dict(num_refs=3, tmvp=1, strong_intra=0, sign_hiding=1, wpp=1, tile_rows=1, intra_in_p=15, all_part_modes=0, amp=0, sao=0, qp_delta=0, deblock_mode=0,
               th_depth_inter=0, th_depth_intra=0, max_cu_log2=6, min_cu_log2=3, nxn_intra=1, chroma_modes=1, transform_skip=0, cabac_init=0, chroma_qp_offsets=0,
               par_mrg_level=2, big_mvd=0, uniform_tiles=1)
    cfg.update(kw)
    run_stream(1920, 1080, 4, seed=12, **cfg)


@pytest.mark.gpu
def test_kvazaar_shaped_stream_1080p(gpu):
    """the shape a Kvazaar peer with uvgComm's settings sends: 1080p (coded height 1080, not 1088), lp-g4d3t1-like reference structure
    (3 pictures, RPS in the slice header), TMVP, intra CUs in P pictures, no strong intra smoothing, sign hiding, WPP"""
    run_stream(1920, 1080, 6, seed=11, num_refs=3, tmvp=1, strong_intra=0, sign_hiding=1, wpp=1, tile_rows=1, intra_in_p=15,
               all_part_modes=0, amp=0, sao=0, qp_delta=0, deblock_mode=0, th_depth_inter=0, th_depth_intra=0, max_cu_log2=6, min_cu_log2=3,
               nxn_intra=1, chroma_modes=1, transform_skip=0, cabac_init=0, chroma_qp_offsets=0, par_mrg_level=2, big_mvd=0, uniform_tiles=1)


@pytest.mark.gpu
def test_everything_on_1080p(gpu):
    run_stream(1920, 1080, 4, seed=5, num_refs=4, tmvp=1, amp=1, sao=1, strong_intra=0, sign_hiding=1, transform_skip=1, cabac_init=1, wpp=1,
               tile_rows=3, uniform_tiles=0, th_depth_inter=2, th_depth_intra=2, qp_delta=3, chroma_qp_offsets=1, deblock_mode=3, par_mrg_level=3,
               intra_in_p=25, all_part_modes=1, chroma_modes=1, nxn_intra=1, max_cu_log2=6, min_cu_log2=3, big_mvd=1)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_frame_threaded_decoder_foreign_streams(gpu, seed):
    """OpenHEVC 'Frame' parallelisation: pictures parsed concurrently; temporal prediction makes a parser follow the collocated picture's"""
    run_stream(416, 240, 12, threads=4, frame_threads=True, seed=seed, tmvp=1, num_refs=3)


@pytest.mark.gpu
@pytest.mark.parametrize("slices,wpp,tile_rows,frame", [(1, 1, 1, False), (1, 0, 1, False), (1, 1, 3, False), (1, 0, 2, False), (2, 1, 3, False), (2, 0, 4, False),
                                                        (1, 1, 1, True), (2, 1, 2, True)])
def test_pictures_in_several_slice_segments(gpu, slices, wpp, tile_rows, frame):
    """uvgComm video/Slices (kvazaarfilter.cpp:205-215): a Kvazaar peer cuts every picture into slice segments, one NAL unit each -- a
    dependent slice segment per CTU row (slices=wpp) or an independent slice per tile (slices=tiles).  libOpenHevcDecode gets them one by
    one and hands out the picture with the last; same pictures as the checker's decoder."""
    run_stream(200, 264, 10, threads=3 if frame else 1, frame_threads=frame, seed=31, density=35, num_refs=2, tmvp=1, sao=1, cabac_init=1, wpp=wpp, tile_rows=tile_rows,
               uniform_tiles=1, qp_delta=2, intra_in_p=20, sign_hiding=1, slices=slices)


@pytest.mark.gpu
def test_slice_segments_1080p_kvazaar_shape(gpu):
    """1080p, WPP, a dependent slice segment per CTU row (17 NAL units per picture), Kvazaar-like references"""
    run_stream(1920, 1080, 4, seed=5, density=25, num_refs=2, tmvp=1, wpp=1, tile_rows=1, slices=1, intra_in_p=10, max_cu_log2=5)


@pytest.mark.gpu
@pytest.mark.parametrize("frame", [False, True])
def test_lost_slice_segments(gpu, frame):
    """packet loss with a picture in several NAL units: a missing segment costs its picture (and the pictures that refer to it) and
    nothing else -- no crash, no stale segment joined to the wrong picture; from the next IDR picture on the output is exact again"""
    from kvazzup_amd.codec import Decoder
    w, h, period, pictures = 200, 264, 6, 24
    g = orc.OracleGen(w, h, seed=9, density=30, intra_period=period, num_refs=1, tmvp=0, wpp=1, tile_rows=1, slices=1)
    od = orc.OracleDecoder()
    gd = Decoder(threads=3, frame_threads=True) if frame else Decoder()
    rng = np.random.default_rng(3)
    refs, got, clean_from = {}, {}, {}
    damaged_period = set()
    for t in range(pictures):
        au = g.picture()
        r = od.decode_au(au, t)
        assert len(r) == 1
        refs[t] = r[0]["i420"]
        nals = [b"\x00\x00\x00\x01" + x for x in au.split(b"\x00\x00\x00\x01")[1:]]
        vcl = [i for i, n in enumerate(nals) if (n[4] >> 1) < 32]
        if t % period == 2 or t % period == 4:                     # lose one segment of this picture: the first, a middle one or the last
            k = vcl[int(rng.integers(0, len(vcl)))] if t % period == 2 else vcl[0]
            nals = nals[:k] + nals[k + 1:]
            damaged_period.add(t // period)
        for n in nals:
            try:
                o = gd.decode_nal(n, t)
            except RuntimeError:                                    # a negative return: what OpenHEVCFilter::process logs (openhevcfilter.cpp:154-157)
                o = None
            if o is not None:
                got[o["pts"]] = o["i420"]
    if frame:
        for _ in range(5):
            try:
                for o in gd.drain():
                    got[o["pts"]] = o["i420"]
                break
            except RuntimeError:
                pass
    for t in range(pictures):
        if t // period not in damaged_period or t % period < 2:    # periods without loss, and the pictures before the first loss of a period
            assert t in got and np.array_equal(got[t], refs[t]), t
    assert len(got) < pictures                                      # the damaged pictures did not come out as if nothing had happened
    gd.close(); od.close(); g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cols,rows,wpp,slices,frame,uniform", [(2, 1, 0, 0, False, 1), (2, 2, 1, 0, False, 1), (3, 2, 0, 2, False, 1), (4, 3, 1, 2, False, 0), (2, 2, 1, 0, True, 1),
                                                               (5, 1, 1, 0, False, 1), (2, 2, 0, 2, True, 1)])
def test_tile_columns(gpu, cols, rows, wpp, slices, frame, uniform):
    """uvgComm video/Tiles with its tile dimension defaults ("2x2" .. "16x16", defaultsettings.cpp:283-324): a Kvazaar peer sends tile grids
    with columns.  CTBs in tile-scan order, contexts and WPP hand-over per tile, nothing available across a tile boundary, one substream
    per tile (or per CTB row of a tile with WPP), optionally a slice per tile."""
    run_stream(328, 264, 8, threads=3 if frame else 1, frame_threads=frame, seed=5, density=30, num_refs=2, tmvp=1, sao=1, qp_delta=2, intra_in_p=20, cabac_init=1,
               tile_cols=cols, tile_rows=rows, wpp=wpp, slices=slices, uniform_tiles=uniform)


@pytest.mark.gpu
def test_tile_grid_1080p(gpu):
    """1080p in 4 x 4 tiles with WPP inside the tiles (68 substreams per picture), a slice per tile"""
    run_stream(1920, 1080, 4, seed=6, density=25, num_refs=2, tmvp=1, wpp=1, tile_cols=4, tile_rows=4, slices=2, intra_in_p=10, max_cu_log2=5, sao=1)


B_FEATURES = [
    dict(b_slices=70),                                                      # low-delay B (Kvazaar bipred=1 with its default GOP): both lists hold the same pictures
    dict(b_slices=70, num_refs=3, tmvp=1),                                  # ... temporal candidates out of bi-predicted collocated blocks
    dict(b_slices=100, num_refs=2, all_part_modes=1, amp=1),                # every partitioning: 8x4 / 4x8 blocks are never bi-predicted
    dict(b_slices=60, gop=4, num_refs=3, tmvp=1),                           # groups of four (decoded 4 2 1 3): references on both sides, output reordering
    dict(b_slices=50, gop=8, num_refs=4, tmvp=1, par_mrg_level=4),          # Kvazaar gop=8
    dict(b_slices=80, gop=2, num_refs=2, tmvp=1, cabac_init=1, intra_in_p=20),
    dict(b_slices=0, gop=4, num_refs=3, tmvp=1),                            # P pictures in reordered groups
    dict(b_slices=60, gop=8, num_refs=4, tmvp=1, sao=1, qp_delta=2, deblock_mode=2, th_depth_inter=2, wpp=0, tile_rows=2, big_mvd=1),
]


@pytest.mark.gpu
@pytest.mark.parametrize("feature", B_FEATURES, ids=lambda kw: "-".join("%s%s" % (k[:3], v) for k, v in kw.items()))
@pytest.mark.parametrize("frame", [False, True])
def test_b_slices_and_output_reordering(gpu, feature, frame):
    """what a uvgComm peer sends once `bipred=1` / `gop=8` sit in its custom-parameter list (kvazaarfilter.cpp:351-371) and OpenHEVC simply decodes
    (openhevcfilter.cpp:145): B slices -- two reference lists, bi-prediction, the merge and AMVP derivations over both lists, temporal candidates out
    of bi-predicted blocks, boundary strength with two vectors a side -- and pictures handed out in POC order, not decoding order"""
    cfg = dict(PLAIN)
    cfg.update(feature)
    run_stream(416, 240, 19, threads=4 if frame else 1, frame_threads=frame, seed=41, intra_period=13, density=25, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(300, 316))
def test_random_b_streams_match_oracle(gpu, seed):
    """every other switch drawn from the seed"""
    sizes = [(416, 240), (352, 288), (200, 136), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 12, seed=seed, b_slices=(40, 70, 100)[seed % 3], gop=(0, 2, 4, 8)[(seed // 3) % 4], intra_period=9,
               threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(400, 424))
def test_random_weighted_streams_match_oracle(gpu, seed):
    """explicit weighted prediction (pred_weight_table(): what x265 writes by default, `weightp`) in P and B slices, every other switch drawn from the seed:
    weights and offsets per list, reference index and component on the 14-bit predictions (8.5.3.3.4.3), uni- and bi-predicted blocks, with and without
    frame threads"""
    sizes = [(416, 240), (352, 288), (200, 136), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 12, seed=seed, weighted=(40, 70, 100)[seed % 3], b_slices=(0, 60, 100)[(seed // 3) % 3], gop=(0, 0, 4, 8)[(seed // 2) % 4], intra_period=9,
               threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(500, 516))
def test_random_streams_with_modified_reference_lists_match_oracle(gpu, seed):
    """ref_pic_lists_modification() (7.3.6.2 / 8.3.4) in P and B slices, with and without weights (their table is indexed through the modified lists) and
    temporal candidates (so is the collocated picture), every other switch drawn from the seed"""
    sizes = [(416, 240), (352, 288), (200, 136), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 12, seed=seed, list_mod=(50, 80, 100)[seed % 3], num_refs=4, b_slices=(0, 60, 100)[(seed // 3) % 3], gop=(0, 0, 4, 8)[(seed // 2) % 4], intra_period=9,
               weighted=(0, 50)[(seed // 4) % 2], threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
def test_weighted_pictures_1080p(gpu):
    """1080p P and B pictures with explicit weights, four reference pictures, TMVP, WPP"""
    run_stream(1920, 1080, 8, seed=23, weighted=70, b_slices=50, gop=0, num_refs=4, tmvp=1, strong_intra=0, sign_hiding=1, wpp=1, tile_rows=1, intra_in_p=10,
               all_part_modes=1, amp=1, sao=1, qp_delta=0, deblock_mode=0, th_depth_inter=1, th_depth_intra=1, max_cu_log2=6, min_cu_log2=3, nxn_intra=1,
               chroma_modes=1, transform_skip=0, cabac_init=0, chroma_qp_offsets=0, par_mrg_level=2, big_mvd=0, uniform_tiles=1, density=20)


@pytest.mark.gpu
def test_b_pictures_1080p_gop8(gpu):
    """1080p, Kvazaar gop=8 shape: hierarchical B pictures, four reference pictures, TMVP, WPP"""
    run_stream(1920, 1080, 10, seed=21, b_slices=80, gop=8, num_refs=4, tmvp=1, strong_intra=0, sign_hiding=1, wpp=1, tile_rows=1, intra_in_p=10,
               all_part_modes=1, amp=0, sao=1, qp_delta=0, deblock_mode=0, th_depth_inter=1, th_depth_intra=1, max_cu_log2=6, min_cu_log2=3, nxn_intra=1,
               chroma_modes=1, transform_skip=0, cabac_init=0, chroma_qp_offsets=0, par_mrg_level=2, big_mvd=0, uniform_tiles=1, density=20)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 6])
def test_b_stream_through_the_filter_graph(gpu, threads):
    """the receiving side of a call as uvgComm builds it -- WireAdapter -> OpenHEVCFilter' (one libOpenHevcDecode per NAL unit, at most one picture out per
    call, its output stage on a thread of its own) -- fed a peer's gop-8 B stream: the pictures arrive complete, in output order, and the stream's end
    (end-of-sequence NAL units) lets the held-back ones out"""
    from kvazzup_amd.pipeline import Pipeline
    w, h, n = 416, 240, 21
    g = orc.OracleGen(w, h, seed=77, intra_period=17, density=25, b_slices=70, gop=8, num_refs=4, tmvp=1, sao=1, wpp=1)
    od = orc.OracleDecoder()
    pl = Pipeline(w, h, settings={"video/OPENHEVC_threads": threads, "video/OH_parallelization": "Frame" if threads > 1 else "Slice", "uvgx/asyncOutput": 1})
    want = []
    for t in range(n):
        au = g.picture()
        want += [f["i420"] for f in od.decode_au(au, t)]
        assert pl.push_encoded(au, t)
    want += [f["i420"] for f in od.flush()]
    pl.push_encoded(None)
    assert len(want) == n and pl.wait(n, 120000), pl.stats()
    for i in range(n):
        got = pl.pop_decoded()
        assert got is not None and (got["width"], got["height"]) == (w, h), i
        assert np.array_equal(got["i420"], want[i]), (i, [k for k in range(n) if np.array_equal(got["i420"], want[k])], int(np.flatnonzero(got["i420"] != want[i])[0]), int(np.count_nonzero(got["i420"] != want[i])))
    pl.close(); od.close(); g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [0, 60])
@pytest.mark.parametrize("frame", [False, True])
def test_decoder_survives_corrupted_b_streams(gpu, frame, weighted):
    """bit flips, truncations and garbage in B pictures of reordered groups: every call returns (a picture, nothing, or an error code), nothing
    hangs or crashes -- vectors, reference indices and list sizes out of a damaged slice never reach a kernel unchecked --, and from the next
    clean IDR picture on the output is the checker's again, in output order"""
    from kvazzup_amd.codec import Decoder, split_nals
    w, h, period = 200, 136, 9
    g = orc.OracleGen(w, h, seed=91, intra_period=period, density=30, b_slices=70, gop=4, num_refs=3, tmvp=1, sao=1, wpp=1, all_part_modes=1, weighted=weighted)      # (weighted: pred_weight_table() in the damaged slice headers too)
    aus = [g.picture() for _ in range(3 * period)]
    g.close()
    od = orc.OracleDecoder()
    want = []
    for t in range(2 * period, 3 * period):
        want += [f["i420"] for f in od.decode_au(aus[t], t)]
    want += [f["i420"] for f in od.flush()]
    od.close()
    assert len(want) == period
    rng = np.random.default_rng(4242)
    gd = Decoder(threads=4, frame_threads=True) if frame else Decoder()
    errors = 0
    for trial in range(int(__import__("os").environ.get("KVZ_FUZZ_TRIALS", "80"))):
        t = int(rng.integers(0, 2 * period))
        au = bytearray(aus[t])
        kind = trial % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                au[min(40 + int(rng.integers(0, max(len(au) - 40, 1))), len(au) - 1)] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            au = au[:max(8, int(rng.integers(8, len(au))))]
        elif kind == 2:
            i = int(rng.integers(6, len(au)))
            au[i:i + 16] = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
        else:
            au = au + bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        for nal in split_nals(bytes(au)):
            try:
                gd.decode_nal(nal, t)
            except RuntimeError:
                errors += 1
    assert errors > 0
    got = []
    for t in range(2 * period, 3 * period):
        for nal in split_nals(aus[t]):
            try:
                f = gd.decode_nal(nal, t)
            except RuntimeError:                      # (what is left of the damaged part may still fail while it drains)
                f = None
            if f is not None:
                got.append(f["i420"])
    got += [f["i420"] for f in gd.drain()]
    assert len(got) >= period
    for k in range(period):
        assert np.array_equal(got[len(got) - period + k], want[k]), k
    gd.close()
