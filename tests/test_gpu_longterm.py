"""GPU parity for streams with long-term reference pictures (round 6): candidates in the SPS and explicit entries in the slice header, with and without
delta_poc_msb_present_flag, used by the current picture or only kept; the reference lists close with them; vectors into them are never scaled (8.5.3.2.7, 8.5.3.2.9).
The synthesiser writes them (long_term); the HIP decoder -- whose kernels only ever see picture buffers -- must reproduce the checker's decoder bit for bit."""
import pytest

from test_gpu_foreign import PLAIN, run_stream


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5, 6])
@pytest.mark.parametrize("feature", [
    dict(num_refs=1),
    dict(num_refs=3, tmvp=1),
    dict(num_refs=2, tmvp=1, list_mod=50, all_part_modes=1),
    dict(num_refs=4, tmvp=1, wpp=0, intra_in_p=15, sao=1),
    dict(num_refs=3, tmvp=1, weighted=50),
])
def test_long_term_reference_pictures_match_oracle(gpu, seed, feature):
    cfg = dict(PLAIN); cfg.update(feature)
    run_stream(416, 240, 24, seed=seed, long_term=1, intra_period=32, slices=0, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 9))
def test_random_streams_with_long_term_reference_pictures(gpu, seed):
    sizes = [(416, 240), (352, 288), (200, 136), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 28, seed=seed, long_term=1, intra_period=20, gop=0, b_slices=0, ctb_log2=(6, 5, 4)[seed % 3], slices=3 if seed % 4 == 0 else 0, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))
