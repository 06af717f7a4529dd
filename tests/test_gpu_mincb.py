"""GPU parity for streams whose minimum coding block is 16 or 32 samples (MinCbLog2SizeY 4 / 5; round 6) -- no split_cu_flag at that size, the partitioning's
binarisation of the minimum size, an inter coding unit cut into four square prediction blocks (PART_NxN, allowed above 8x8 only), intra NxN with 8x8 / 16x16
prediction blocks.  The synthesiser writes them (min_cb_log2); the HIP decoder must reproduce the checker's decoder bit for bit."""
import pytest

from test_gpu_foreign import PLAIN, run_stream


@pytest.mark.gpu
@pytest.mark.parametrize("min_cb,ctb,size", [(4, 6, (416, 240)), (4, 5, (416, 240)), (4, 4, (208, 144)), (5, 6, (416, 224)), (5, 5, (192, 128))])
@pytest.mark.parametrize("feature", [
    dict(all_part_modes=1),                                  # inter NxN
    dict(all_part_modes=1, amp=1, tmvp=1, num_refs=3),
    dict(intra_in_p=40, nxn_intra=1, chroma_modes=1),        # intra NxN: 8x8 / 16x16 prediction blocks, one chroma mode
    dict(intra_period=1, nxn_intra=1),
    dict(qp_delta=2, sao=1, intra_in_p=20, all_part_modes=1),
    dict(th_depth_inter=2, th_depth_intra=2, all_part_modes=1, intra_in_p=20, nxn_intra=1, transform_skip=1),
])
def test_feature_with_larger_minimum_coding_blocks(gpu, min_cb, ctb, size, feature):
    cfg = dict(PLAIN); cfg.update(feature); cfg["max_cu_log2"] = ctb
    run_stream(size[0], size[1], 5, seed=9, ctb_log2=ctb, min_cb_log2=min_cb, slices=0, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 17))
def test_random_streams_with_larger_minimum_coding_blocks(gpu, seed):
    """every other switch drawn from the seed (free slices in a third of them); with and without frame threads"""
    min_cb, ctb, w, h = [(4, 6, 416, 240), (4, 5, 352, 288), (4, 4, 208, 144), (5, 6, 192, 128), (5, 5, 640, 352), (4, 6, 64, 64)][seed % 6]
    run_stream(w, h, 8, seed=seed, ctb_log2=ctb, min_cb_log2=min_cb, slices=3 if seed % 3 == 0 else 0, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
def test_1080p_with_16_sample_minimum_coding_blocks(gpu):
    run_stream(1920, 1088, 4, seed=5, density=20, num_refs=2, tmvp=1, wpp=1, intra_in_p=10, sao=1, all_part_modes=1, min_cb_log2=4, slices=0)
