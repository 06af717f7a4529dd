"""GPU parity for streams with coding tree blocks of 32 and 16 samples (round 6) -- what encoders other than Kvazaar write (openhevcfilter.cpp:145 decodes
whatever arrives; the SDP offers plain "H265").  The synthesiser (oracle/hevc_gen.c, ctb_log2) writes them; the HIP decoder must reproduce the checker's
decoder -- which tests/test_python_decoder.py holds to the second, independently written decoder on the same kind of stream -- bit for bit."""
import pytest

from test_gpu_foreign import PLAIN, run_stream


@pytest.mark.gpu
@pytest.mark.parametrize("ctb_log2", [5, 4])
@pytest.mark.parametrize("feature", [
    dict(),                                               # plain: P pictures, one reference, WPP
    dict(intra_in_p=30, nxn_intra=1, chroma_modes=1),     # intra blocks in P pictures: the chain's work unit is the CTB
    dict(intra_period=1, nxn_intra=1, strong_intra=0),    # all intra
    dict(sao=1),                                          # SAO parameters per CTB
    dict(wpp=0, tile_rows=2, tile_cols=2, intra_in_p=20), # tiles whose boundaries fall inside a 64x64 area
    dict(qp_delta=2, chroma_qp_offsets=1),                # quantisation groups relative to the CTB
    dict(num_refs=3, tmvp=1, all_part_modes=1, amp=1),    # the collocated block's CTB row
    dict(th_depth_inter=2, th_depth_intra=2, intra_in_p=20, transform_skip=1, sign_hiding=1),
    dict(b_slices=70, gop=4, tmvp=1, num_refs=2),
])
def test_feature_with_small_ctbs_matches_oracle(gpu, ctb_log2, feature):
    cfg = dict(PLAIN); cfg.update(feature); cfg["max_cu_log2"] = min(cfg["max_cu_log2"], ctb_log2)
    run_stream(416, 240, 6, seed=7, ctb_log2=ctb_log2, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("ctb_log2", [5, 4])
@pytest.mark.parametrize("seed", range(1, 13))
def test_random_streams_with_small_ctbs_match_oracle(gpu, ctb_log2, seed):
    """every other switch drawn from the seed; with and without frame threads"""
    sizes = [(416, 240), (352, 288), (200, 136), (64, 64), (24, 16), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 8, seed=seed, ctb_log2=ctb_log2, slices=0, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
@pytest.mark.parametrize("ctb_log2", [5, 4])
def test_1080p_with_small_ctbs_matches_oracle(gpu, ctb_log2):
    run_stream(1920, 1080, 4, seed=5, density=20, num_refs=2, tmvp=1, wpp=1, tile_rows=1, intra_in_p=10, sao=1, ctb_log2=ctb_log2, slices=0)


@pytest.mark.gpu
def test_ctb_size_changes_between_sequences_of_one_size(gpu):
    """the same picture size with 64, 16, 32 and again 64-sample CTBs, one decoder: every change re-lays the input block out (per-CTB tables, edge words, ticket order)"""
    import numpy as np
    import orc
    from kvazzup_amd.codec import Decoder
    gd = Decoder()
    try:
        t = 0
        for ctb in (6, 4, 5, 6):
            g = orc.OracleGen(352, 288, seed=20 + ctb, ctb_log2=ctb, slices=0, intra_in_p=20, sao=1, wpp=1)
            od = orc.OracleDecoder()
            for _ in range(4):
                au = g.picture()
                ref = od.decode_au(au, t); got = gd.decode_au(au, t)
                assert len(ref) == 1 and len(got) == 1 and np.array_equal(got[0]["i420"], ref[0]["i420"]), (ctb, t)
                t += 1
            g.close(); od.close()
    finally:
        gd.close()
