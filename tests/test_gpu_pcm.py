"""GPU parity for streams with PCM coding units (pcm_flag: raw samples at their own bit depths in the middle of the arithmetic codeword, which ends in front of them
and starts again behind them; pcm_loop_filter_disabled_flag) -- round 6.  The synthesiser writes them (pcm = probability); the HIP decoder must reproduce the
checker's decoder bit for bit."""
import pytest

from test_gpu_foreign import PLAIN, run_stream


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5])                      # (bit depths, PCM sizes and the loop filter flag are drawn per stream)
@pytest.mark.parametrize("feature", [
    dict(intra_period=1),                                     # all intra: PCM units as neighbours of every prediction mode
    dict(intra_in_p=50),                                      # P pictures
    dict(intra_in_p=40, sao=1, deblock_mode=2),               # the loop filters around (and, flag permitting, inside) PCM units
    dict(intra_in_p=40, qp_delta=2, max_cu_log2=6),           # the QP predictor runs through PCM units; 32x32 PCM units
    dict(intra_in_p=40, wpp=0, tile_rows=2, tile_cols=2),
    dict(intra_in_p=40, slices=3, tq_bypass=20),
    dict(intra_in_p=40, ctb_log2=4, nxn_intra=1),
])
def test_pcm_units_match_oracle(gpu, seed, feature):
    cfg = dict(PLAIN); cfg.update(feature)
    if "slices" not in cfg:
        cfg["slices"] = 0
    run_stream(416, 240, 5, seed=seed, pcm=35, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 13))
def test_random_streams_with_pcm_units(gpu, seed):
    sizes = [(416, 240), (352, 288), (200, 136), (64, 64), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 6, seed=seed, pcm=25, intra_in_p=30, ctb_log2=(6, 5, 4)[seed % 3], slices=3 if seed % 4 == 0 else 0, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
def test_1080p_with_pcm_units(gpu):
    run_stream(1920, 1080, 3, seed=5, density=20, wpp=1, intra_in_p=30, sao=1, pcm=20, slices=0)
