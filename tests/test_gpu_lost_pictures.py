"""GPU side of tests/test_lost_pictures.py: access units lost on the way -- the HIP decoder puts a copy of the nearest reference picture (a grey one when it has none) where a reference picture is missing (concealment v2,
decoder.h) and goes on; every picture it hands out, wrong as the ones behind a loss are, is the checker's bit for bit; nothing but the lost pictures is missing.
Before round 6's last day every picture up to the next IDR picture was answered with an error code -- two seconds of frozen video per lost packet at uvgComm's
intra period of 64."""
import numpy as np
import pytest

import orc
from test_lost_pictures import lossy


def run(cut, threads, must_conceal=True):
    from kvazzup_amd.codec import Decoder
    od = orc.OracleDecoder()
    gd = Decoder(threads=threads, frame_threads=True) if threads > 1 else Decoder()
    want, got = [], []
    try:
        for t, au in cut:
            want += od.decode_au(au, t)
            got += gd.decode_au(au, t)
        want += od.flush()
        got += gd.drain()
        assert od.concealed() > 0 or not must_conceal
    finally:
        gd.close()
        od.close()
    assert [f["pts"] for f in got] == [f["pts"] for f in want], ([f["pts"] for f in got], [f["pts"] for f in want])
    assert len(got) == len(cut) or not must_conceal
    for a, b in zip(got, want):
        if not np.array_equal(a["i420"], b["i420"]):
            d = np.flatnonzero(a["i420"] != b["i420"])
            pytest.fail("picture with time stamp %d: %d samples differ, first at %d" % (a["pts"], len(d), d[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("seed", range(1, 13))
def test_a_grey_picture_stands_in_for_a_lost_reference_picture(gpu, seed, threads):
    cut, _, _ = lossy(seed, n=30, w=416, h=240, slices=(0, 1, 3)[seed % 3])
    run(cut, threads)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
def test_losses_in_a_stream_of_the_hip_encoder(gpu, threads):
    """a Kvazaar-shaped stream (this project's encoder: WPP, one reference picture): every third P picture lost"""
    from kvazzup_amd.codec import Encoder
    w, h = 640, 384
    e = Encoder(w, h, options=(("qp", 30), ("period", 16), ("me-range", 16)))
    aus = []
    for t in range(34):
        au, _ = e.encode(orc.synth_frame(0, 0x5EED0006, w, h, t))
        aus.append(au)
    e.close()
    cut = [(t, a) for t, a in enumerate(aus) if t % 16 == 0 or t % 3 != 1]
    assert len(cut) < len(aus)
    run(cut, threads)


@pytest.mark.gpu
@pytest.mark.parametrize("every", [4, 7])
def test_the_filter_chain_keeps_delivering_over_a_lossy_wire(gpu, every):
    """KvazaarFilter' -> wire -> OpenHEVCFilter' with every n-th access unit lost between them (harness setting uvgx/wireLossEvery; never an IDR picture): the chain
    delivers every picture that arrived, and they are the checker's decode of the same NAL units"""
    from kvazzup_amd.pipeline import Pipeline
    w, h, frames, period = 416, 240, 40, 16
    pl = Pipeline(w, h, settings={"video/QP": 30, "video/Intra": period, "uvgx/wireLossEvery": every}, custom=(("me-range", 16),))
    od = orc.OracleDecoder()
    try:
        clip = [np.ascontiguousarray(orc.synth_frame(0, 0x5EED0008, w, h, t)) for t in range(frames)]
        for f in clip:                                              # (paced: a uvgComm filter drops inputs that find its buffer full -- not the loss this test is about)
            assert pl.push_host_paced(f, max_backlog=4)
        lost = (frames - len(range(0, frames, period))) // every
        assert lost >= 4 and pl.wait(frames - lost, 60000)
        seen, arrived = 0, []
        for t in range(frames):                                     # (the harness's rule, applied to the access units as they left the encoder)
            au, pts = pl.pop_encoded()
            assert pts == t
            if t % period != 0:
                seen += 1
                if seen % every == 0:
                    continue
            arrived.append((t, au))
        assert len(arrived) == frames - lost
        for t, au in arrived:
            want = od.decode_au(au, t)
            assert len(want) == 1
            d = pl.pop_decoded()
            assert d["pts"] == t and np.array_equal(d["i420"], want[0]["i420"]), "picture %d" % t
        assert od.concealed() > 0
        st = pl.stats()
        assert st["decoded_pictures"] == frames - lost
    finally:
        pl.close()
        od.close()
