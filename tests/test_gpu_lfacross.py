"""GPU parity for pictures with boundaries CLOSED to the in-loop filters (round 6): loop_filter_across_tiles_enabled_flag = 0 -- what Kvazaar writes for every stream
with tiles (its tiles are filtered one by one) -- and slices with slice_loop_filter_across_slices_enabled_flag = 0, per slice (of two slices the later one's flag
decides: 7.4.7.1).  Deblocking leaves the edges on a closed boundary alone, SAO the samples whose neighbour lies across one.  The synthesiser draws the flags
(lf_across); the HIP decoder must reproduce the checker's decoder bit for bit."""
import pytest

from test_gpu_foreign import PLAIN, run_stream


@pytest.mark.gpu
@pytest.mark.parametrize("lf", [1, 2])
@pytest.mark.parametrize("layout", [
    dict(slices=0, tile_rows=2, tile_cols=2, wpp=0),       # tiles, one slice
    dict(slices=0, tile_rows=3, tile_cols=1, wpp=1),       # tile rows with WPP
    dict(slices=2, tile_rows=2, tile_cols=3, wpp=0),       # a slice per tile (Kvazaar's slices=tiles): both flags
    dict(slices=2, tile_rows=2, tile_cols=1, wpp=1),
    dict(slices=3, wpp=0),                                 # free slices, each its own flag
    dict(slices=3, wpp=1),
])
@pytest.mark.parametrize("feature", [
    dict(sao=1, intra_in_p=20),
    dict(sao=1, intra_period=1),
    dict(sao=0, deblock_mode=2, all_part_modes=1),
    dict(sao=1, qp_delta=2, tq_bypass=30, intra_in_p=20),
])
def test_closed_boundaries_match_oracle(gpu, lf, layout, feature):
    cfg = dict(PLAIN); cfg.update(layout); cfg.update(feature)
    run_stream(416, 240, 5, seed=13, lf_across=lf, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("ctb_log2", [6, 5, 4])
@pytest.mark.parametrize("seed", range(1, 13))
def test_random_streams_with_closed_boundaries(gpu, ctb_log2, seed):
    """every other switch drawn from the seed; tiles, a slice per tile or free slices by turns; with and without frame threads"""
    sizes = [(416, 240), (352, 288), (200, 136), (128, 128), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    layout = [dict(slices=0, tile_rows=2, tile_cols=2), dict(slices=2, tile_rows=2, tile_cols=2), dict(slices=3)][seed % 3]
    run_stream(w, h, 6, seed=seed, ctb_log2=ctb_log2, lf_across=1 + (seed & 1), sao=1, threads=3 if seed & 2 else 1, frame_threads=bool(seed & 2), **layout)


@pytest.mark.gpu
def test_1080p_kvazaar_style_tiles(gpu):
    """what a Kvazaar peer with uvgComm's tiles setting sends: loop_filter_across_tiles_enabled_flag = 0, pps_loop_filter_across_slices_enabled_flag = 0"""
    run_stream(1920, 1080, 4, seed=5, density=20, num_refs=2, tmvp=1, wpp=1, tile_rows=2, tile_cols=2, intra_in_p=10, sao=1, slices=0, lf_across=2)
