"""GPU parity for streams with constrained_intra_pred_flag = 1 (round 6): neighbouring samples of blocks that are not intra-coded are no reference samples for intra
prediction -- availability becomes any pattern along a block's two borders, and the substitution process of 8.4.4.2.2 runs over it (a wave ballot in the chain's
per-wave blocks, a unit mask in the 32x32 form).  The synthesiser sets the flag (cip); the HIP decoder must reproduce the checker's decoder bit for bit."""
import pytest

from test_gpu_foreign import PLAIN, run_stream


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5])
@pytest.mark.parametrize("feature", [
    dict(intra_in_p=45),                                          # intra blocks among inter ones: every border pattern
    dict(intra_in_p=45, nxn_intra=1, chroma_modes=1),             # 4x4 luma blocks, chroma modes of their own
    dict(intra_in_p=45, max_cu_log2=6, strong_intra=1),           # 32x32 blocks (the workgroup-shaped form), strong smoothing over substituted samples
    dict(intra_in_p=45, all_part_modes=1, amp=1),                 # 8x4 / 4x8 / asymmetric inter blocks beside intra ones: four-sample granularity
    dict(intra_in_p=45, wpp=0, tile_rows=2, tile_cols=2),
    dict(intra_in_p=45, slices=3, lf_across=1, sao=1),
    dict(intra_in_p=45, ctb_log2=4, th_depth_intra=2),
    dict(intra_in_p=45, pcm=20, tq_bypass=20),
])
def test_constrained_intra_prediction_matches_oracle(gpu, seed, feature):
    cfg = dict(PLAIN); cfg.update(feature)
    if "slices" not in cfg:
        cfg["slices"] = 0
    run_stream(416, 240, 5, seed=seed, cip=1, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 13))
def test_random_streams_with_constrained_intra_prediction(gpu, seed):
    sizes = [(416, 240), (352, 288), (200, 136), (64, 64), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 6, seed=seed, cip=1, intra_in_p=40, ctb_log2=(6, 5, 4)[seed % 3], slices=3 if seed % 4 == 0 else 0, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
def test_1080p_with_constrained_intra_prediction(gpu):
    run_stream(1920, 1080, 3, seed=5, density=20, wpp=1, intra_in_p=35, sao=1, cip=1, slices=0, max_cu_log2=6)
