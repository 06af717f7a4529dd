"""The second, independently written decoder (tests/pyhevc.py: plain Python + numpy, from the standard's text) against the
checker's decoder (oracle/hevc_dec.c) on streams of the checker's encoder (the golden configurations) and of the stream
synthesiser (every partitioning, transform trees, several references, TMVP, cu_qp_delta, sign hiding, transform skip, SAO,
tiles, WPP).  Pictures must match bit for bit.  CPU only."""
import numpy as np
import pytest

import orc
import pyhevc
from test_oracle_kat import table


def tabs():
    return {"range_lps": table(1, np.uint8, (64, 4)), "trans_lps": table(2, np.uint8, (64,)), "trans_mps": table(13, np.uint8, (64,)),
            "dct": table(0, np.int8, (32, 32)).astype(int), "dst": table(5, np.int8, (4, 4)).astype(int),
            "luma_filter": table(11, np.int8, (4, 8)).astype(int), "chroma_filter": table(12, np.int8, (8, 4)).astype(int),
            "beta": table(6, np.uint8, (52,)).astype(int), "tc": table(7, np.uint8, (54,)).astype(int),
            "intra_angle": table(9, np.int8, (35,)).astype(int), "inv_angle": table(10, np.int16, (35,)).astype(int)}


def test_context_init_values_agree_with_the_checker_and_the_product():
    """the per-syntax-element initValue tables typed in pyhevc.py against the checker's table (its own context order)"""
    init = table(3, np.uint8, (3, 154)).astype(int)
    # order of the checker's / product's context indices (hevc_core.h CTX_*)
    order = [("sao_merge", 1), ("sao_type", 1), ("split_cu", 3), ("tq_bypass", 1), ("skip", 3), ("pred_mode", 1), ("part_mode", 4), ("prev_intra", 1),
             ("chroma_mode", 1), ("rqt_root", 1), ("merge_flag", 1), ("merge_idx", 1), ("inter_pred_idc", 5), ("ref_idx", 2), ("mvp", 1), ("split_tf", 3),
             ("cbf_luma", 2), ("cbf_chroma", 4), ("mvd_gt0", 1), ("mvd_gt1", 1), ("qp_delta", 2), ("ts_flag", 2), ("last_x", 18), ("last_y", 18),
             ("csbf", 4), ("sig", 42), ("gt1", 24), ("gt2", 6)]
    at = 0
    for name, n in order:
        if name is not None:
            for t in range(3):
                vals = pyhevc.INIT[name][t]
                if vals is None:
                    continue
                got = init[t, at:at + len(vals)].tolist()
                assert got == vals, (name, t, got, vals)
        at += n
    assert at == 154


def decode_both(aus):
    od = orc.OracleDecoder()
    want = []
    for au in aus:
        want += od.decode_au(au)
    want += od.flush()                       # (pictures held back for reordering)
    od.close()
    pd = pyhevc.Decoder(tabs())
    for au in aus:
        pd.decode(au)
    return want, pd.flush()


def compare(aus):
    want, got = decode_both(aus)
    assert len(want) == len(got) and len(got) > 0
    for k, (a, b) in enumerate(zip(want, got)):
        assert (a["width"], a["height"], a["poc"]) == (b["width"], b["height"], b["poc"]), k
        d = np.nonzero(a["i420"] != b["i420"])[0]
        assert d.size == 0, "picture %d: %d samples differ, first at %d" % (k, d.size, d[0])


CASES = [
    dict(w=128, h=64, qp=32, period=1, me_range=8, kind=0, seed=0x5EED0001, wpp=1, deblock=1, frames=2),
    dict(w=192, h=128, qp=30, period=64, me_range=8, kind=0, seed=0x5EED0002, wpp=1, deblock=1, frames=3),
    dict(w=192, h=128, qp=22, period=64, me_range=8, kind=2, seed=0x5EED0003, wpp=0, deblock=1, frames=2),
    dict(w=192, h=128, qp=40, period=64, me_range=8, kind=1, seed=0x5EED0005, wpp=1, deblock=0, frames=2),
    dict(w=192, h=128, qp=30, period=3, me_range=8, kind=0, seed=0x5EED0006, wpp=1, deblock=1, frames=4, tile_rows=2, sao=1),
    dict(w=128, h=128, qp=35, period=64, me_range=8, kind=2, seed=0x5EED0007, wpp=0, deblock=1, frames=2, sao=1, qp_in_cu=1),
    dict(w=256, h=192, qp=30, period=3, me_range=8, kind=0, seed=0x5EED0008, wpp=1, deblock=1, frames=4, tile_rows=2, tile_cols=2, sao=1, subme=2),   # a tile grid, fractional vectors
    dict(w=256, h=128, qp=32, period=64, me_range=8, kind=0, seed=0x5EED0009, wpp=0, deblock=1, frames=3, tile_rows=1, tile_cols=3, vaq=6),           # columns only, a QP chain per tile
]


@pytest.mark.parametrize("c", CASES, ids=lambda c: "%dx%d-qp%d-%x" % (c["w"], c["h"], c["qp"], c["seed"] & 0xff))
def test_checker_encoder_streams(c):
    e = orc.OracleEncoder(c["w"], c["h"], qp=c["qp"], period=c["period"], me_range=c["me_range"], wpp=c["wpp"], deblock=c["deblock"],
                           tile_rows=c.get("tile_rows", 1), sao=c.get("sao", 0), qp_in_cu=c.get("qp_in_cu", 0), mv_jitter=c.get("mv_jitter", 0),
                           tile_cols=c.get("tile_cols", 1), subme=c.get("subme", 0), vaq=c.get("vaq", 0))
    if c.get("qp_in_cu"):
        e.set_roi(2, 2, [-4, 3, 6, -7])
    aus = [e.encode(orc.synth_frame(c["kind"], c["seed"], c["w"], c["h"], t)) for t in range(c["frames"])]
    e.close()
    compare(aus)


@pytest.mark.parametrize("c", [dict(w=192, h=128, qp=32, period=64, subme=0, iip=0, tile_rows=1, frames=4),
                               dict(w=128, h=128, qp=27, period=1, subme=0, iip=0, tile_rows=2, frames=2),
                               dict(w=256, h=128, qp=30, period=64, subme=4, iip=2, tile_rows=1, frames=4)], ids=lambda c: "%dx%d-p%d" % (c["w"], c["h"], c["period"]))
def test_checker_encoder_lossless(c):
    """`lossless` (oracle/hevc_enc.h, uvgComm's check box kvazaarfilter.cpp:244): cu_transquant_bypass everywhere -- the checker's decoder and the decoder
    written from the standard's text both return the SOURCE pictures"""
    e = orc.OracleEncoder(c["w"], c["h"], qp=c["qp"], period=c["period"], me_range=8, subme=c["subme"], tile_rows=c["tile_rows"], sao=1, deblock=1)
    e.set_option("intra-in-p", c["iip"]); e.set_option("lossless", 1)
    src = [orc.synth_frame(0 if t < 2 else 2, 0x5EED0010, c["w"], c["h"], t) for t in range(c["frames"])]     # (a change of content: intra units in P pictures)
    aus = [e.encode(f) for f in src]
    e.close()
    want, got = decode_both(aus)
    assert len(want) == len(got) == len(src)
    for t, f in enumerate(src):
        assert np.array_equal(want[t]["i420"], f), t
        assert np.array_equal(got[t]["i420"], f), t


GEN = [
    dict(width=64, height=64, seed=11, pictures=2),
    dict(width=136, height=72, seed=12, pictures=3),
    dict(width=128, height=128, seed=13, pictures=3, wpp=1),
    dict(width=128, height=192, seed=14, pictures=3, tile_rows=2),
    dict(width=96, height=80, seed=15, pictures=4, num_refs=3),
    dict(width=128, height=64, seed=16, pictures=4, tmvp=1, num_refs=2),
]


@pytest.mark.parametrize("g", GEN, ids=lambda g: "gen%d" % g["seed"])
def test_generator_streams(g):
    g = dict(g)
    n = g.pop("pictures")
    gen = orc.OracleGen(**g)
    aus = [gen.picture() for _ in range(n)]
    gen.close()
    compare(aus)


@pytest.mark.parametrize("seed", range(200, 224))
def test_generator_random_sweep(seed):
    """everything drawn from the seed (orc_gen_config fields left at -1), odd sizes included"""
    gen = orc.OracleGen(width=64 + 8 * (seed % 13), height=64 + 8 * (seed % 7), seed=seed, density=40)
    aus = [gen.picture() for _ in range(5)]
    gen.close()
    compare(aus)


def test_generator_everything_on():
    gen = orc.OracleGen(width=200, height=136, seed=77, density=45, num_refs=4, tmvp=1, amp=1, sao=1, sign_hiding=1, transform_skip=1, cabac_init=1,
                        wpp=1, tile_rows=2, th_depth_inter=2, th_depth_intra=2, qp_delta=2, chroma_qp_offsets=1, deblock_mode=2, par_mrg_level=3,
                        intra_in_p=25, all_part_modes=1, chroma_modes=1, nxn_intra=1, big_mvd=1)
    aus = [gen.picture() for _ in range(6)]
    gen.close()
    compare(aus)


@pytest.mark.parametrize("kw", [
    dict(scaling_lists=1),                                                   # scaling_list_enabled_flag with the default lists: Kvazaar's `scaling-list default` (uvgComm's checkbox)
    dict(scaling_lists=2, th_depth_inter=2, th_depth_intra=2, nxn_intra=1),  # lists in the SPS (explicit, copied, default), every block size
    dict(scaling_lists=3, transform_skip=1),                                 # default in the SPS, the PPS's lists override them; 4x4 transform-skip blocks are scaled too
    dict(scaling_lists=4, max_cu_log2=6, density=20),                        # both; 32x32 blocks
    dict(tq_bypass=35),                                                      # cu_transquant_bypass_flag: the residual is the level array
    dict(tq_bypass=50, sao=1, sign_hiding=1, transform_skip=1, deblock_mode=2, intra_in_p=30),   # ... and the loop filters leave those samples alone
    dict(tq_bypass=100, sao=1),                                              # a lossless stream (Kvazaar's `lossless`, uvgComm's checkbox)
    dict(scaling_lists=4, tq_bypass=25, qp_delta=2, chroma_qp_offsets=1),
], ids=lambda kw: "-".join("%s%s" % (k[:4], v) for k, v in kw.items()))
def test_generator_scaling_lists_and_transquant_bypass(kw):
    """the two tools behind uvgComm's "scaling list" and "lossless" checkboxes (kvazaarfilter.cpp:235-244), read from the standard's text a second
    time: scaling_list_data with its three ways of coding a list, the scaling factors of 7.4.5, cu_transquant_bypass in parse, reconstruction,
    deblocking and SAO"""
    cfg = dict(width=200, height=136, seed=91, density=35, num_refs=2, tmvp=1)
    cfg.update(kw)
    gen = orc.OracleGen(**cfg)
    aus = [gen.picture() for _ in range(5)]
    gen.close()
    compare(aus)


@pytest.mark.parametrize("cols,rows,wpp,uniform", [(2, 1, 0, 1), (2, 2, 1, 1), (3, 2, 0, 1), (4, 3, 1, 0), (5, 1, 1, 1)])
def test_generator_tile_columns(cols, rows, wpp, uniform):
    """tile grids with columns (what a Kvazaar peer with uvgComm's tile dimension defaults sends): tile-scan order, contexts and the WPP
    hand-over per tile, nothing available across a tile boundary -- read here from the standard's text, independently of the checker"""
    gen = orc.OracleGen(width=328, height=264, seed=5, density=30, num_refs=2, tmvp=1, sao=1, qp_delta=2, intra_in_p=20, cabac_init=1,
                        tile_cols=cols, tile_rows=rows, wpp=wpp, uniform_tiles=uniform)
    aus = [gen.picture() for _ in range(4)]
    gen.close()
    compare(aus)


B_CASES = [
    dict(b_slices=70),                                                       # low-delay B: both lists hold the same pictures (Kvazaar bipred=1 with its default low-delay GOP)
    dict(b_slices=60, gop=4, tmvp=1),                                        # groups of four, decoded 4 2 1 3: references on both sides, collocated pictures out of either list
    dict(b_slices=50, gop=8, num_refs=4, tmvp=1, amp=1, all_part_modes=1),   # Kvazaar gop=8: three pictures held back for output; 8x4 / 4x8 blocks (never bi-predicted)
    dict(b_slices=80, gop=2, num_refs=3, tmvp=1, par_mrg_level=4, cabac_init=1),
    dict(b_slices=0, gop=4, num_refs=3, tmvp=1),                             # reordered P pictures: a P slice whose collocated block is... still uni-predicted, but forward references exist
    dict(b_slices=100, gop=8, num_refs=4, tmvp=1, sao=1, qp_delta=2, wpp=1, tile_rows=2, intra_in_p=15),
]


@pytest.mark.parametrize("kw", B_CASES, ids=lambda kw: "-".join("%s%s" % (k[:3], v) for k, v in kw.items()))
@pytest.mark.parametrize("seed", (301, 302, 303))
def test_generator_b_slices_and_reordering(kw, seed):
    """B slices as a uvgComm peer produces them through Kvazaar's custom parameters (bipred=1, gop=8 -- kvazaarfilter.cpp:351-371), read from the
    standard's text a second time: inter_pred_idc, the two reference lists, mvd_l1_zero_flag, merge candidates with the combined bi-predictive
    and two-list zero candidates, AMVP across lists, the temporal candidate out of a bi-predicted collocated block (NoBackwardPredFlag,
    collocated_from_l0_flag), the rounded average of two 14-bit predictions, boundary strength with two vectors per side, and pictures handed
    out in POC order rather than decoding order"""
    cfg = dict(width=136 + 8 * (seed % 9), height=72 + 8 * (seed % 5), seed=seed, density=25, intra_period=9)
    cfg.update(kw)
    gen = orc.OracleGen(**cfg)
    aus = [gen.picture() for _ in range(11)]
    gen.close()
    want, got = decode_both(aus)
    compare(aus)
    assert [f["poc"] for f in want] == sorted(f["poc"] for f in want[:9]) + sorted(f["poc"] for f in want[9:])      # output order: by POC inside each coded video sequence
    if kw["b_slices"]:
        assert any(f["slice_type"] == 0 for f in want)


W_CASES = [
    dict(weighted=60, num_refs=3, tmvp=1),                                   # P slices: what x265 writes by default (weightp) when it detects a fade
    dict(weighted=50, b_slices=70, num_refs=4, tmvp=1, gop=4),               # B slices: weights on both lists, the bi-predictive form with both offsets
    dict(weighted=100, b_slices=50, num_refs=2, amp=1, all_part_modes=1, sao=1),
]


@pytest.mark.parametrize("kw", W_CASES, ids=lambda kw: "-".join("%s%s" % (k[:3], v) for k, v in kw.items()))
@pytest.mark.parametrize("seed", (401, 402, 403))
def test_generator_weighted_prediction(kw, seed):
    """explicit weighted sample prediction (pred_weight_table() 7.3.6.3, the derived weights 7.4.7.3, the sample formulas 8.5.3.3.4.3), read from the
    standard's text a second time: denominators 0..7, weights and offsets over their whole ranges, uni- and bi-predicted blocks"""
    cfg = dict(width=136 + 8 * (seed % 9), height=72 + 8 * (seed % 5), seed=seed, density=25, intra_period=9)
    cfg.update(kw)
    gen = orc.OracleGen(**cfg)
    aus = [gen.picture() for _ in range(10)]
    gen.close()
    compare(aus)
    pd = pyhevc.Decoder(tabs())
    seen = 0
    for au in aus:
        pd.decode(au)
        wp = pd.last_sh.get("wp")
        if wp:
            ld, cd, lists = wp
            seen += sum(1 for ent in lists for e in ent if e[0] != (1 << ld, 0) or e[1] != (1 << cd, 0) or e[2] != (1 << cd, 0))
    assert seen > 0                                                          # (the streams do carry weights that differ from the defaults)


@pytest.mark.parametrize("kw", [dict(list_mod=70, num_refs=4, tmvp=1), dict(list_mod=60, b_slices=70, num_refs=4, tmvp=1, gop=4, weighted=40),
                                dict(list_mod=100, b_slices=50, num_refs=3, gop=8)], ids=lambda kw: "-".join("%s%s" % (k[:3], v) for k, v in kw.items()))
@pytest.mark.parametrize("seed", (501, 502, 503))
def test_generator_reference_list_modification(kw, seed):
    """ref_pic_lists_modification() (7.3.6.2, 8.3.4): the slice orders the entries of the initial lists as it likes (repeats included), in P and B slices, with
    temporal candidates and weights indexed by the modified lists"""
    cfg = dict(width=136 + 8 * (seed % 9), height=72 + 8 * (seed % 5), seed=seed, density=25, intra_period=9)
    cfg.update(kw)
    gen = orc.OracleGen(**cfg)
    aus = [gen.picture() for _ in range(12)]
    gen.close()
    compare(aus)
    pd = pyhevc.Decoder(tabs())
    seen = 0
    for au in aus:
        pd.decode(au)
        seen += sum(1 for e in (pd.last_sh.get("list_entry") or []) if e)
    assert seen > 0                                                          # (the streams do modify lists)


@pytest.mark.parametrize("ctb_log2", [5, 4])
@pytest.mark.parametrize("seed,kw", [(3, dict(wpp=1, sao=1, intra_in_p=20, nxn_intra=1, tmvp=1, num_refs=2)),
                                     (7, dict(wpp=0, sao=1, tile_rows=2, tile_cols=2, qp_delta=2, intra_in_p=30, tmvp=1, num_refs=3, all_part_modes=1, amp=1)),
                                     (11, dict(b_slices=60, gop=4, tmvp=1, sao=0, deblock_mode=2, wpp=1)),
                                     (17, dict(scaling_lists=2, tq_bypass=20, transform_skip=1, sign_hiding=1)),
                                     (19, dict(qp_delta=4, chroma_qp_offsets=1, wpp=1, tile_rows=3))])
def test_coding_tree_blocks_of_32_and_16_samples(ctb_log2, seed, kw):
    """round 6: the synthesiser writes streams with CtbLog2SizeY 5 and 4 (what encoders other than Kvazaar choose: z-scan availability by CTB raster order,
    WPP rows, SAO parameters, quantisation groups, the intra mode's above candidate and the collocated block's row all per CTB of that size) -- the two
    independently written decoders must agree on every picture before the HIP decoder is held to either"""
    g = orc.OracleGen(328, 200, seed=seed, ctb_log2=ctb_log2, slices=0, **kw)
    assert g.config["ctb_log2"] == ctb_log2 and g.config["max_cu_log2"] <= ctb_log2
    aus = [g.picture() for _ in range(5)]
    g.close()
    compare(aus)


@pytest.mark.parametrize("wpp", [0, 1])
@pytest.mark.parametrize("seed,kw", [(3, dict(sao=1, intra_in_p=30, nxn_intra=1)),
                                     (7, dict(qp_delta=2, intra_in_p=30, tmvp=1, num_refs=3, all_part_modes=1, amp=1)),
                                     (11, dict(intra_period=1, strong_intra=0, chroma_modes=1)),
                                     (13, dict(ctb_log2=4, sao=1, intra_in_p=20, cabac_init=1)),
                                     (19, dict(ctb_log2=5, qp_delta=3, chroma_qp_offsets=1, deblock_mode=2))])
def test_free_slices(wpp, seed, kw):
    """round 6: slice segments that begin at any coding tree block, independent slices (own slice_qp_delta) and dependent segments mixed (oracle/hevc_gen.c,
    slices = 3) -- availability by slice (prediction, context selection, merge candidates, SAO merging), the context variables at the start of a segment and of
    a CTB row under WPP (9.3.1), the QP predictor per slice: the two independently written decoders must agree before the HIP decoder is held to either"""
    g = orc.OracleGen(200, 136, seed=seed, slices=3, wpp=wpp, **kw)
    aus = [g.picture() for _ in range(4)]
    g.close()
    assert max(sum(1 for n in orc.split_nals(au) if ((n[4] >> 1) & 63) < 32) for au in aus) > 1      # (more than one slice segment in some picture)
    compare(aus)


@pytest.mark.parametrize("seed,min_cb,ctb,w,h,kw", [(5, 5, 6, 192, 128, dict(all_part_modes=1, amp=1, intra_in_p=20, sao=1)),
                                                    (7, 4, 4, 208, 144, dict(all_part_modes=1, nxn_intra=1, intra_in_p=30, qp_delta=1)),
                                                    (9, 4, 5, 208, 144, dict(all_part_modes=1, tmvp=1, num_refs=2, slices=3)),
                                                    (11, 5, 5, 192, 128, dict(all_part_modes=1, nxn_intra=1, intra_in_p=40)),
                                                    (13, 4, 6, 208, 144, dict(all_part_modes=1, nxn_intra=1, intra_in_p=30, th_depth_inter=2, th_depth_intra=2))])
def test_minimum_coding_blocks_of_16_and_32_samples(seed, min_cb, ctb, w, h, kw):
    """round 6: MinCbLog2SizeY 4 / 5 -- no split flag at that size, the partitioning's binarisation at the minimum size (inter NxN above 8x8), intra NxN with larger
    prediction blocks: the two independently written decoders must agree before the HIP decoder is held to either"""
    g = orc.OracleGen(w, h, seed=seed, min_cb_log2=min_cb, ctb_log2=ctb, **kw)
    assert g.config["min_cb_log2"] == min_cb
    aus = [g.picture() for _ in range(4)]
    g.close()
    compare(aus)


@pytest.mark.parametrize("lf", [1, 2])
@pytest.mark.parametrize("seed,kw", [(3, dict(slices=3, wpp=0)), (5, dict(slices=3, wpp=1)), (7, dict(slices=0, tile_rows=2, tile_cols=2, wpp=0)), (9, dict(slices=0, tile_rows=3, tile_cols=1, wpp=1))])
def test_boundaries_closed_to_the_loop_filters(lf, seed, kw):
    """round 6: loop_filter_across_tiles_enabled_flag = 0 (what Kvazaar writes with tiles) and slice_loop_filter_across_slices_enabled_flag = 0 per slice -- deblocking
    skips the edges on a closed boundary (8.7.2.3), SAO the samples whose neighbour lies across one (8.7.3.2); of two slices the later one's flag decides (7.4.7.1).
    pyhevc asks per sample pair (lf_ok), the checker per coding tree block and neighbour: two restatements that must agree"""
    g = orc.OracleGen(200, 136, seed=seed, lf_across=lf, sao=1, intra_in_p=20, **kw)
    aus = [g.picture() for _ in range(3)]
    g.close()
    compare(aus)


@pytest.mark.parametrize("seed,kw", [(3, dict(wpp=0)), (4, dict(wpp=1, sao=1)), (5, dict(wpp=1, slices=3)), (6, dict(wpp=0, tile_rows=2, tile_cols=2, slices=0, sao=1)), (7, dict(intra_period=1, max_cu_log2=6)),
                                     (8, dict(ctb_log2=4, qp_delta=1))])
def test_pcm_coding_units(seed, kw):
    """round 6: pcm_flag (a terminating bin), pcm_alignment_zero_bits, the samples at PcmBitDepthY / C, the arithmetic decoder started again behind them with the
    contexts as they are (9.3.2.5); the unit is DC for its neighbours' candidate lists, its QpY the predicted one, pcm_loop_filter_disabled_flag keeps deblocking and
    SAO off its samples -- the two independently written decoders must agree before the HIP decoder is held to either"""
    g = orc.OracleGen(200, 136, seed=seed, pcm=30, intra_in_p=40, **kw)
    aus = [g.picture() for _ in range(3)]
    g.close()
    compare(aus)


@pytest.mark.parametrize("seed,kw", [(3, dict(num_refs=1)), (4, dict(num_refs=3, tmvp=1)), (5, dict(num_refs=2, tmvp=1, list_mod=50)), (6, dict(num_refs=4, tmvp=1, wpp=0, all_part_modes=1))])
def test_long_term_reference_pictures(seed, kw):
    """round 6: long_term_ref_pics_present_flag -- candidates of the SPS (lt_idx_sps) and explicit entries (poc_lsb_lt), delta_poc_msb_present_flag / the accumulated
    cycles (7-52), marking (8.3.2: the long-term entries first, among all reference pictures), the lists closing with RefPicSetLtCurr (8.3.4), and the vector rules:
    a spatial neighbour's vector into another picture counts when both pictures are long-term (as it is) or both short-term (scaled), the temporal candidate likewise
    (8.5.3.2.7, 8.5.3.2.9) -- the two independently written decoders must agree before the HIP decoder is held to either"""
    g = orc.OracleGen(136, 72, seed=seed, long_term=1, intra_period=32, density=20, **kw)
    aus = [g.picture() for _ in range(20)]
    g.close()
    compare(aus)


@pytest.mark.parametrize("seed,kw", [(3, dict(wpp=0)), (4, dict(wpp=1, nxn_intra=1, chroma_modes=1)), (5, dict(wpp=1, slices=3)), (6, dict(wpp=0, all_part_modes=1, amp=1, max_cu_log2=6, strong_intra=1)), (7, dict(ctb_log2=4, pcm=20))])
def test_constrained_intra_prediction(seed, kw):
    """round 6: constrained_intra_pred_flag -- a neighbouring sample of a block that is not intra-coded is marked "not available" for intra prediction (8.4.4.2.2), the
    substitution process runs over whatever pattern that leaves; nothing else changes (the candidate modes, the inter blocks).  The two independently written decoders
    must agree before the HIP decoder is held to either; the flag changes every P picture of these streams"""
    g = orc.OracleGen(200, 136, seed=seed, cip=1, intra_in_p=45, **kw)
    aus = [g.picture() for _ in range(4)]
    g.close()
    compare(aus)
