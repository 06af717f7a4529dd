"""GPU parity for pictures of FREE slices (round 6): slice segments that begin at any coding tree block, independent slices (own slice_qp_delta) and dependent
segments mixed, with and without WPP -- what an encoder that cuts slices by bytes or block counts sends (openhevcfilter.cpp:145 decodes whatever arrives).
The synthesiser writes them (oracle/hevc_gen.c, slices = 3); the HIP decoder must reproduce the checker's decoder bit for bit.  Such a picture is complete when
its access unit ends: the decoder hands it out with the first NAL unit of the next one (or the end of the sequence)."""
import numpy as np
import pytest

import orc
from test_gpu_foreign import PLAIN, run_stream


@pytest.mark.gpu
@pytest.mark.parametrize("wpp", [0, 1])
@pytest.mark.parametrize("feature", [
    dict(),                                                  # P pictures: merge / AMVP candidates stop at the slice border
    dict(intra_in_p=40, nxn_intra=1, chroma_modes=1),        # intra prediction: neighbours in another slice are not available (the kernels' availability test)
    dict(intra_period=1, nxn_intra=1, strong_intra=0),       # all intra
    dict(sao=1, intra_in_p=20),                              # SAO merge candidates stay inside the slice; the filter itself crosses
    dict(qp_delta=2, chroma_qp_offsets=1, intra_in_p=20),    # the QP predictor starts again with every slice, from ITS SliceQpY
    dict(num_refs=3, tmvp=1, all_part_modes=1, amp=1),
    dict(deblock_mode=2, sign_hiding=1, th_depth_inter=2, th_depth_intra=2, intra_in_p=20),
    dict(cabac_init=1, max_cu_log2=6, intra_in_p=20),
])
def test_feature_with_free_slices_matches_oracle(gpu, wpp, feature):
    cfg = dict(PLAIN); cfg.update(feature); cfg["wpp"] = wpp
    run_stream(416, 240, 6, seed=11, slices=3, **cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("ctb_log2", [6, 5, 4])
@pytest.mark.parametrize("seed", range(1, 17))
def test_random_streams_with_free_slices_match_oracle(gpu, ctb_log2, seed):
    """every other switch drawn from the seed (WPP, dependent segments allowed or not, tools); with and without frame threads"""
    sizes = [(416, 240), (352, 288), (200, 136), (64, 64), (136, 16), (648, 360)]
    w, h = sizes[seed % len(sizes)]
    run_stream(w, h, 8, seed=seed, ctb_log2=ctb_log2, slices=3, threads=3 if seed & 1 else 1, frame_threads=bool(seed & 1))


@pytest.mark.gpu
@pytest.mark.parametrize("wpp", [0, 1])
def test_1080p_with_free_slices_matches_oracle(gpu, wpp):
    run_stream(1920, 1080, 4, seed=5, density=20, num_refs=2, tmvp=1, wpp=wpp, intra_in_p=10, sao=1, slices=3)


@pytest.mark.gpu
def test_a_stream_turns_to_free_slices_and_back(gpu):
    """one decoder: one-slice pictures, then pictures of free slices (the first of them is taken back when its only-looking-whole first segment ends early), then
    Kvazaar's dependent segment per CTU row, again one slice -- every picture against the checker, in order"""
    from kvazzup_amd.codec import Decoder
    gd = Decoder()
    try:
        t = 0
        for k, slices in enumerate((0, 3, 1, 3, 0)):
            g = orc.OracleGen(352, 288, seed=40 + k, slices=slices, wpp=k & 1, intra_in_p=20, sao=1)
            od = orc.OracleDecoder()
            refs, got = [], []
            for _ in range(4):
                au = g.picture()
                refs += [f["i420"] for f in od.decode_au(au, t)]
                got += gd.decode_au(au, t)
                t += 1
            got += gd.drain()
            assert len(refs) == 4 and len(got) == 4, (k, slices, len(got))
            for i in range(4):
                assert np.array_equal(got[i]["i420"], refs[i]), (k, slices, i)
            g.close(); od.close()
    finally:
        gd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
def test_a_segment_at_a_rows_start_behind_a_whole_picture(gpu, threads):
    """(soak seed 1578; tests/test_free_slices.py has the story) the pictures are the checker's"""
    from test_gpu_random_access import both
    g = orc.OracleGen(64, 64, seed=1578, intra_period=16, tmvp=1, ctb_log2=4, min_cb_log2=3, cip=0, pcm=0, lf_across=2, intra_in_p=30, slices=3, gop=4, b_slices=0, open_gop=0,
                      temporal_layers=1, rps_forms=1)
    aus = [g.picture() for _ in range(24)]
    g.close()
    assert len(both(aus, range(24), threads, threads > 1)) == 24
