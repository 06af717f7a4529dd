"""GPU: the C++ mirrors of KvazaarFilter / OpenHEVCFilter (csrc/filters.hip), threads and queues
included, give the same access units as the CPU checker and decode them back to its reconstruction."""
import numpy as np
import pytest

import orc

SEED = 0x5EED0000


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,qp,period,tile_rows", [(320, 240, 32, 64, 1), (416, 240, 27, 4, 1), (320, 256, 30, 4, 2)])
def test_filter_chain_matches_oracle(gpu, w, h, qp, period, tile_rows):
    from kvazzup_amd.pipeline import Pipeline
    frames = 8
    oe = orc.OracleEncoder(w, h, qp=qp, period=period, me_range=16, tile_rows=tile_rows)
    settings = {"video/QP": qp, "video/Intra": period}
    if tile_rows > 1:                                   # uvgComm settings video/Tiles + video/tileDimensions (kvazaarfilter.cpp:196-202)
        settings.update({"video/Tiles": 1, "video/tileDimensions": "1x%d" % tile_rows})
    pl = Pipeline(w, h, settings=settings, custom=(("me-range", 16),))
    try:
        clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
        for f in clip:
            pl.push(f)
        assert pl.wait(frames, 60000)
        for t in range(frames):
            au_o = oe.encode(clip[t])
            au_g, pts = pl.pop_encoded()
            assert pts == t and au_g == au_o, "AU %d differs" % t
            d = pl.pop_decoded()
            assert d["width"] == w and d["height"] == h and d["pts"] == t
            assert np.array_equal(d["i420"], oe.recon()), "decoded picture %d differs from the reconstruction" % t
        st = pl.stats()
        assert st["encoded_pictures"] == frames and st["decoded_pictures"] == frames and st["dropped"] == 0
        # every AU reaches the decoder as single NAL units: 3 parameter sets + 1 slice for IDR pictures, 1 slice otherwise
        n_idr = sum(1 for t in range(frames) if t % period == 0)
        assert st["received_nals"] == frames + 3 * n_idr
    finally:
        pl.close()
        oe.close()


@pytest.mark.gpu
def test_filter_chain_with_the_lossless_box_ticked(gpu):
    """uvgComm's `video/lossless` (videosettings.cpp:188 -> kvazaarfilter.cpp:244): what leaves the OpenHEVCFilter mirror IS what went into the KvazaarFilter mirror"""
    from kvazzup_amd.pipeline import Pipeline
    w, h, frames = 416, 240, 6
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16)
    oe.set_option("lossless", 1)
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/lossless": 1}, custom=(("me-range", 16),))
    try:
        clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
        for f in clip:
            pl.push(f)
        assert pl.wait(frames, 60000)
        for t in range(frames):
            au_g, pts = pl.pop_encoded()
            assert pts == t and au_g == oe.encode(clip[t]), "AU %d differs" % t
            d = pl.pop_decoded()
            assert d["pts"] == t and np.array_equal(d["i420"], clip[t]), "picture %d is not the source" % t
    finally:
        pl.close()
        oe.close()


@pytest.mark.gpu
def test_encoder_rejects_mismatching_input_and_unknown_option(gpu):
    from kvazzup_amd.pipeline import Pipeline
    pl = Pipeline(256, 128, custom=(("no-such-option", 1),))     # logged as invalid custom parameter, like kvazaarfilter.cpp:363-367
    try:
        pl.push(orc.synth_frame(0, SEED, 256, 128, 0))
        assert pl.wait(1, 30000)
        pl.lib.uvgx_pipeline_push_host(pl.p, np.zeros(128 * 64 * 3 // 2, np.uint8).ctypes.data, 128, 64, 30, 1, 99)   # wrong size: dropped (kvazaarfilter.cpp:381-399)
        pl.push(orc.synth_frame(0, SEED, 256, 128, 1))
        assert pl.wait(2, 30000)
        assert pl.stats()["encoded_pictures"] == 2
    finally:
        pl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,owf,threads,recon", [(320, 240, 0, 1, 1), (640, 368, 2, 4, 1), (1920, 1080, 6, 8, 0), (416, 240, 3, 3, 1)])
def test_host_boundary_pipelined_matches_oracle(gpu, w, h, owf, threads, recon):
    """The reference's own boundary with everything that overlaps it switched on: host pictures borrowed from the caller, copied into
    page-locked kvz_pictures, uploaded on the encoder's copy stream (ring of device buffers), video/OWF pictures in flight; the decoder's
    pictures come down on its download stream into the ring of host buffers behind libOpenHevcGetOutput.  Every access unit and every
    decoded picture equals the CPU checker's (416x240: a host pitch wider than the picture)."""
    from kvazzup_amd.pipeline import Pipeline
    frames = 20 if w < 1000 else 10
    oe = orc.OracleEncoder(w, h, qp=32, period=8, me_range=8)
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 8, "video/OWF": owf, "video/OPENHEVC_threads": threads,
                                  "video/OH_parallelization": "Frame" if threads > 1 else "Slice"},
                  custom=(("me-range", 8), ("recon-output", recon)))
    try:
        clip = [orc.synth_frame(0, SEED + 3, w, h, t) for t in range(frames)]
        for f in clip:
            assert pl.push_host_paced(f, 6, 60000, borrow=True)
        pl.flush()
        assert pl.wait(frames, 120000)
        for t in range(frames):
            au_o = oe.encode(clip[t])
            au_g, pts = pl.pop_encoded()
            assert pts == t and au_g == au_o, "AU %d differs" % t
            d = pl.pop_decoded()
            assert d["width"] == w and d["height"] == h and d["pts"] == t
            assert np.array_equal(d["i420"], oe.recon()), "decoded picture %d differs from the reconstruction" % t
    finally:
        pl.close()
        oe.close()
