"""GPU parity: the HIP decoder behind libOpenHevc* (include/openHevcWrapper.h) against the CPU
checker's decoder and against the encoder's reconstruction (closed loop), bit-exact."""
import numpy as np
import pytest

import orc

SEED = 0x5EED0000


def run_clip(w, h, frames, qp, period, me_range, kind, wpp=1, deblock=1, tile_rows=1, threads=1, sao=0, mv_jitter=0):
    from kvazzup_amd.codec import Decoder
    oe = orc.OracleEncoder(w, h, qp=qp, period=period, me_range=me_range, wpp=wpp, deblock=deblock, tile_rows=tile_rows, sao=sao, mv_jitter=mv_jitter)
    od = orc.OracleDecoder()
    gd = Decoder()
    try:
        for t in range(frames):
            frame = orc.synth_frame(kind, SEED, w, h, t)
            au = oe.encode(frame)
            ref = od.decode_au(au, t)
            got = gd.decode_au(au, t)
            assert len(ref) == 1 and len(got) == 1, (t, len(ref), len(got))
            assert got[0]["width"] == w and got[0]["height"] == h
            assert got[0]["fps"] == (30, 1)
            assert np.array_equal(ref[0]["i420"], oe.recon())           # oracle closed loop
            if not np.array_equal(got[0]["i420"], ref[0]["i420"]):
                d = np.flatnonzero(got[0]["i420"] != ref[0]["i420"])
                pytest.fail("frame %d: %d samples differ, first at %d" % (t, len(d), d[0]))
    finally:
        gd.close()
        od.close()
        oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads,frame", [(1, False), (4, True)])
def test_decoder_follows_a_resolution_change(gpu, threads, frame):
    """a sender that changes its resolution mid-call (uvgComm re-opens Kvazaar: new VPS / SPS / PPS and an IDR picture): the decoder
    re-sizes its buffers at the new SPS -- pictures of the old size still in the frame-threaded ring are output first -- and the
    reported size follows; then back to the first size"""
    from kvazzup_amd.codec import Decoder, Encoder
    gd = Decoder(threads=threads, frame_threads=frame)
    want, got = [], []
    t = 0
    for (w, h) in ((320, 192), (640, 384), (320, 192), (200, 120)):
        ge = Encoder(w, h, options=(("qp", 30), ("period", 64), ("me-range", 8)))
        for k in range(5):
            au, rec = ge.encode(orc.synth_frame(0, SEED + w, w, h, k))
            want.append((w, h, rec))
            got += gd.decode_au(au, t)
            t += 1
        ge.close()
    eos = bytes([0, 0, 0, 1, 36 << 1, 1])
    for _ in range(threads if frame else 0):                       # end-of-sequence NAL units drain the frame-threaded ring
        o = gd.decode_nal(eos)
        if o is not None:
            got.append(o)
    assert len(got) == len(want), (len(got), len(want))
    for i, (g, (w, h, rec)) in enumerate(zip(got, want)):
        assert (g["width"], g["height"]) == (w, h), (i, g["width"], g["height"], w, h)
        assert np.array_equal(g["i420"], rec), i
    gd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 6])
def test_resolution_change_through_the_filter_with_its_output_stage(gpu, threads):
    """the same mid-call resolution changes through OpenHEVCFilter' with its asynchronous output stage (uvgx/asyncOutput = 1: pictures are copied
    out of the decoder's frame memory on a thread of their own while later NAL units are already being decoded): frame memory a resolution
    change retires -- the host output buffers, the pictures completed ahead of their turn -- must stay valid until the copies are done"""
    from kvazzup_amd.codec import Encoder
    from kvazzup_amd.pipeline import Pipeline
    pl = Pipeline(640, 384, settings={"video/OPENHEVC_threads": threads, "video/OH_parallelization": "Frame" if threads > 1 else "Slice", "uvgx/asyncOutput": 1})
    want, t = [], 0
    for rep in range(3):
        for (w, h) in ((320, 192), (640, 384), (200, 120)):
            ge = Encoder(w, h, options=(("qp", 30), ("period", 64), ("me-range", 8)))
            for k in range(4):
                au, rec = ge.encode(orc.synth_frame(0, SEED + w + rep, w, h, k))
                want.append((w, h, rec))
                assert pl.push_encoded(au, t)
                t += 1
            ge.close()
    pl.push_encoded(None)
    assert pl.wait(len(want), 120000), pl.stats()
    for i, (w, h, rec) in enumerate(want):
        g = pl.pop_decoded()
        assert g is not None and (g["width"], g["height"], g["pts"]) == (w, h, i), (i, g and (g["width"], g["height"], g["pts"]))
        assert np.array_equal(g["i420"], rec), i
    pl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=128, h=64, frames=2, qp=32, period=1, me_range=8, kind=0),
    dict(w=256, h=192, frames=5, qp=32, period=64, me_range=16, kind=0),
    dict(w=320, h=240, frames=4, qp=22, period=64, me_range=8, kind=2),
    dict(w=192, h=128, frames=3, qp=10, period=64, me_range=8, kind=2),
    dict(w=192, h=128, frames=3, qp=32, period=64, me_range=8, kind=1),
    dict(w=416, h=240, frames=6, qp=32, period=4, me_range=32, kind=0, wpp=0),
    dict(w=640, h=360, frames=4, qp=27, period=64, me_range=16, kind=0, deblock=0),
    dict(w=130, h=70, frames=3, qp=0, period=64, me_range=1, kind=2),
])
def test_decoder_matches_oracle(gpu, cfg):
    run_clip(**cfg)


@pytest.mark.gpu
def test_gpu_encoder_to_gpu_decoder_closed_loop(gpu):
    """encode on the GPU through kvz_api, decode on the GPU through libOpenHevc*: decoded == recon"""
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = 704, 576
    ge = Encoder(w, h, options=(("qp", 30), ("period", 8), ("me-range", 16)))
    gd = Decoder()
    try:
        for t in range(10):
            frame = orc.synth_frame(0, SEED + 1, w, h, t)
            au, rec = ge.encode(frame)
            got = gd.decode_au(au, t)
            assert len(got) == 1
            assert np.array_equal(got[0]["i420"], rec), "frame %d" % t
    finally:
        gd.close()
        ge.close()


@pytest.mark.gpu
def test_decoder_gates_until_parameter_sets_and_rejects_garbage(gpu):
    from kvazzup_amd.codec import Decoder, split_nals
    oe = orc.OracleEncoder(128, 64, qp=32, period=1, me_range=4)
    au = oe.encode(orc.synth_frame(0, SEED, 128, 64, 0))
    nals = split_nals(au)
    assert [n[4] >> 1 for n in nals] == [32, 33, 34, 19]
    gd = Decoder()
    try:
        assert gd.decode_nal(nals[3]) is None                # VCL before VPS/SPS/PPS: discarded (openhevcfilter.cpp:134-136)
        for n in nals[:3]:
            assert gd.decode_nal(n) is None
        assert gd.decode_nal(nals[3]) is not None
        bad = nals[3][:6] + bytes(len(nals[3]) - 6)
        with pytest.raises(RuntimeError):
            gd.decode_nal(bad)
    finally:
        gd.close()
        oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [2, 4])
def test_frame_threaded_decoder_delays_output_and_drains_on_eos(gpu, threads):
    """libOpenHevcInit(n, OH_THREAD_FRAME): n pictures are parsed concurrently while one more is reconstructed,
    so the output lags n pictures; end-of-sequence NAL units drain the rest; the decoded pictures are
    identical to the synchronous decoder's."""
    from kvazzup_amd import _native as N
    from kvazzup_amd.codec import Decoder
    w, h, frames = 320, 240, 9
    oe = orc.OracleEncoder(w, h, qp=30, period=4, me_range=8)
    aus = [oe.encode(orc.synth_frame(0, SEED, w, h, t)) for t in range(frames)]
    recs = []
    oe2 = orc.OracleEncoder(w, h, qp=30, period=4, me_range=8)
    for t in range(frames):
        oe2.encode(orc.synth_frame(0, SEED, w, h, t)); recs.append(oe2.recon())
    gd = Decoder.__new__(Decoder)
    gd.lib = N.load_library()
    gd.h = gd.lib.libOpenHevcInit(threads, 1)                      # OH_THREAD_FRAME
    assert gd.lib.libOpenHevcStartDecoder(gd.h) == 0
    gd.download = True; gd.vps = gd.sps = gd.pps = False
    out = []
    for t, au in enumerate(aus):
        got = gd.decode_au(au, t)
        assert len(got) == (1 if t >= threads else 0), (t, len(got))
        out += got
    eos = bytes([0, 0, 0, 1, 36 << 1, 1])
    for _ in range(threads):
        out.append(gd.decode_nal(eos))
    assert gd.decode_nal(eos) is None                               # nothing left
    assert [o["pts"] for o in out] == list(range(frames))
    for t in range(frames):
        assert np.array_equal(out[t]["i420"], recs[t]), t
    gd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=320, h=256, frames=5, qp=30, period=3, me_range=16, kind=0, tile_rows=2),
    dict(w=320, h=256, frames=4, qp=27, period=2, me_range=8, kind=2, tile_rows=2, wpp=0),
    dict(w=256, h=448, frames=5, qp=32, period=4, me_range=32, kind=0, tile_rows=3),
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, tile_rows=4),
])
def test_decoder_tile_rows(gpu, cfg):
    """streams with full-width tile rows (uniform spacing, with and without WPP) from the checker's encoder"""
    run_clip(**cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,wpp,tile_rows", [(320, 256, 1, 1), (320, 256, 0, 1), (448, 320, 1, 2), (1280, 720, 1, 1)])
def test_decoder_cu_qp_delta(gpu, w, h, wpp, tile_rows):
    """streams with cu_qp_delta (quantisation group = CTU) from the checker's encoder with a changing delta-QP map: per-CTU
    dequantisation, QpY prediction chain, deblocking with the averaged QpY"""
    from kvazzup_amd.codec import Decoder
    rng = np.random.default_rng(3)
    oe = orc.OracleEncoder(w, h, qp=30, period=4, me_range=16, wpp=wpp, tile_rows=tile_rows, qp_in_cu=1)
    gd = Decoder()
    for t in range(7):
        if t == 1: oe.set_roi(4, 3, rng.integers(-12, 13, 12))
        if t == 3: oe.set_roi(7, 5, rng.integers(-30, 31, 35))
        if t == 5: oe.set_roi(0, 0, None)
        au = oe.encode(orc.synth_frame(0 if t < 6 else 2, 7, w, h, t))
        got = gd.decode_au(au, t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], oe.recon()), t
    gd.close(); oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [dict(), dict(sao=1, tile_rows=2), dict(qp_in_cu=1, wpp=0)])
def test_decoder_survives_corrupted_streams(gpu, extra):
    """bit flips, truncations and garbage in the slice data: every call returns (a picture or an error code), nothing hangs
    or crashes, and the decoder is usable again from the next IDR picture"""
    from kvazzup_amd.codec import Decoder, split_nals
    w, h = 320, 256
    oe = orc.OracleEncoder(w, h, qp=28, period=3, me_range=16, **extra)
    if extra.get("qp_in_cu"):
        oe.set_roi(3, 2, [-6, 0, 5, 9, -3, 2])
    aus = [oe.encode(orc.synth_frame(0 if t % 2 else 2, SEED, w, h, t)) for t in range(6)]
    recs = []
    oe2 = orc.OracleEncoder(w, h, qp=28, period=3, me_range=16, **extra)
    if extra.get("qp_in_cu"):
        oe2.set_roi(3, 2, [-6, 0, 5, 9, -3, 2])
    for t in range(6):
        oe2.encode(orc.synth_frame(0 if t % 2 else 2, SEED, w, h, t)); recs.append(oe2.recon())
    rng = np.random.default_rng(12345)
    gd = Decoder()
    errors = pictures = 0
    for trial in range(int(__import__("os").environ.get("KVZ_FUZZ_TRIALS", "120"))):
        t = int(rng.integers(0, 6))
        au = bytearray(aus[t])
        kind = trial % 4
        body = max(len(au) - 40, 1)                      # leave the first bytes (start code, NAL header) mostly alone
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                i = 40 + int(rng.integers(0, body)) if len(au) > 41 else 0
                au[min(i, len(au) - 1)] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            au = au[:max(8, int(rng.integers(8, len(au))))]
        elif kind == 2:
            i = int(rng.integers(6, len(au)))
            au[i:i + 16] = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
        else:
            au = au + bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        for nal in split_nals(bytes(au)):
            try:
                if gd.decode_nal(nal, t) is not None:
                    pictures += 1
            except RuntimeError:
                errors += 1
    assert errors > 0 and pictures >= 0
    # recovery: a clean IDR and its followers decode exactly
    for t in (3, 4, 5):
        got = gd.decode_au(aus[t], t)
        assert len(got) == 1 and np.array_equal(got[0]["i420"], recs[t]), t
    gd.close(); oe.close(); oe2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=320, h=256, frames=6, qp=32, period=4, me_range=16, kind=0, sao=1),
    dict(w=320, h=256, frames=4, qp=37, period=2, me_range=8, kind=2, sao=1, wpp=0),
    dict(w=448, h=320, frames=5, qp=22, period=3, me_range=16, kind=0, sao=1, tile_rows=2),
    dict(w=130, h=70, frames=3, qp=27, period=64, me_range=8, kind=0, sao=1),
    dict(w=640, h=360, frames=3, qp=30, period=64, me_range=16, kind=0, sao=1, deblock=0),
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, sao=1),
])
def test_decoder_sao(gpu, cfg):
    """streams with sample adaptive offset from the checker's encoder: sao() parsed per CTU (merge left / up, band and edge
    offsets), the filter (8.7.3) after deblocking, filtered pictures as references"""
    run_clip(**cfg)


@pytest.mark.gpu
def test_sao_stream_through_frame_threads_and_filters(gpu):
    """GPU encoder with sao=full -> GPU decoder with frame threads: decoded pictures equal the encoder's reconstruction"""
    from kvazzup_amd import _native as N
    from kvazzup_amd.codec import Decoder, Encoder
    w, h, frames = 640, 384, 8
    ge = Encoder(w, h, options=(("qp", 30), ("period", 4), ("me-range", 16), ("sao", "full"), ("owf", 2)))
    assert not ge.rejected, ge.rejected
    clip = [orc.synth_frame(0, SEED, w, h, t) for t in range(frames)]
    outs = [ge.encode(f) for f in clip] + [ge.encode(None) for _ in range(2)]
    outs = [o for o in outs if o[0] is not None]
    assert len(outs) == frames
    gd = Decoder.__new__(Decoder)
    gd.lib = N.load_library()
    gd.h = gd.lib.libOpenHevcInit(3, 1)
    assert gd.lib.libOpenHevcStartDecoder(gd.h) == 0
    gd.download = True; gd.vps = gd.sps = gd.pps = False
    dec = []
    for t, (au, _) in enumerate(outs):
        dec += gd.decode_au(au, t)
    eos = bytes([0, 0, 0, 1, 36 << 1, 1])
    for _ in range(3):
        d = gd.decode_nal(eos)
        if d is not None:
            dec.append(d)
    assert len(dec) == frames
    for t in range(frames):
        assert np.array_equal(dec[t]["i420"], outs[t][1]), t
    gd.close(); ge.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(w=320, h=256, frames=6, qp=30, period=8, me_range=8, kind=0, mv_jitter=1),
    dict(w=416, h=240, frames=6, qp=24, period=64, me_range=16, kind=2, mv_jitter=1, wpp=0),          # noise: every block coded on top of the interpolation
    dict(w=130, h=70, frames=5, qp=35, period=64, me_range=32, kind=0, mv_jitter=1),                  # long vectors + fractions across the picture edges
    dict(w=640, h=360, frames=4, qp=32, period=64, me_range=16, kind=0, mv_jitter=1, sao=1, deblock=0),
    dict(w=1920, h=1080, frames=3, qp=32, period=64, me_range=16, kind=0, mv_jitter=1),
])
def test_decoder_fractional_motion_vectors(gpu, cfg):
    """streams whose vectors have quarter-sample fractions (the checker's encoder with its test hook: the product's encoder
    searches integer positions only): 8-tap luma / 4-tap chroma interpolation (8.5.3.3.3), all 16 x 64 fraction pairs occur"""
    run_clip(**cfg)


@pytest.mark.gpu
def test_frame_threaded_decoder_survives_corrupted_streams(gpu):
    """the same abuse with libOpenHevcInit(4, OH_THREAD_FRAME): errors surface with the delayed picture they belong to, nothing
    hangs, and after draining a clean IDR and its followers decode exactly"""
    from kvazzup_amd import _native as N
    from kvazzup_amd.codec import Decoder, split_nals
    w, h = 320, 256
    oe = orc.OracleEncoder(w, h, qp=28, period=3, me_range=16)
    aus, recs = [], []
    for t in range(6):
        aus.append(oe.encode(orc.synth_frame(0 if t % 2 else 2, SEED, w, h, t))); recs.append(oe.recon())
    rng = np.random.default_rng(777)
    gd = Decoder.__new__(Decoder)
    gd.lib = N.load_library()
    gd.h = gd.lib.libOpenHevcInit(4, 1)
    assert gd.lib.libOpenHevcStartDecoder(gd.h) == 0
    gd.download = True; gd.vps = gd.sps = gd.pps = False
    errors = 0
    for trial in range(int(__import__("os").environ.get("KVZ_FUZZ_TRIALS", "120"))):
        t = int(rng.integers(0, 6))
        au = bytearray(aus[t])
        if trial % 3 == 0:
            for _ in range(int(rng.integers(1, 6))):
                au[min(40 + int(rng.integers(0, max(len(au) - 40, 1))), len(au) - 1)] ^= 1 << int(rng.integers(0, 8))
        elif trial % 3 == 1:
            au = au[:max(8, int(rng.integers(8, len(au))))]
        for nal in split_nals(bytes(au)):
            try:
                gd.decode_nal(nal, t)
            except RuntimeError:
                errors += 1
    eos = bytes([0, 0, 0, 1, 36 << 1, 1])
    for _ in range(8):                                   # drain whatever is still in flight (errors included)
        try:
            gd.decode_nal(eos)
        except RuntimeError:
            errors += 1
    assert errors > 0
    out = []
    for t in (3, 4, 5):
        out += gd.decode_au(aus[t], t)
    for _ in range(4):
        d = gd.decode_nal(eos)
        if d is not None:
            out.append(d)
    assert len(out) == 3
    for k, t in enumerate((3, 4, 5)):
        assert np.array_equal(out[k]["i420"], recs[t]), t
    gd.close(); oe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("frame_threads", [False, True])
def test_whole_pictures_in_one_segment_with_dependent_segments_enabled(gpu, frame_threads):
    """a PPS with dependent_slice_segments_enabled_flag = 1 does not oblige the stream to use dependent segments: a one-tile picture without WPP
    sent as ONE segment has the same first slice header as one that begins a row-by-row sequence.  The decoder takes it for the latter until the
    next access unit begins, then decodes the segment as the whole picture it is (close_open_picture) instead of dropping it."""
    from kvazzup_amd.codec import Decoder
    w, h = 320, 192
    oe = orc.OracleEncoder(w, h, qp=30, period=4, me_range=8, wpp=0)
    gd = Decoder(threads=4 if frame_threads else 1, frame_threads=frame_threads); od = orc.OracleDecoder()
    got, want = [], []
    for t in range(6):
        au = bytearray(oe.encode(orc.synth_frame(0, SEED, w, h, t)))
        pos = 0
        for nal in orc.split_nals(bytes(au)):
            if (nal[4] >> 1) & 63 == 34:
                au[pos + 6] |= 0x20          # pps_pic_parameter_set_id ue(0), pps_seq_parameter_set_id ue(0), dependent_slice_segments_enabled_flag
            pos += len(nal)
        want.append(oe.recon())
        ref = od.decode_au(bytes(au), t)
        assert len(ref) == 1 and np.array_equal(ref[0]["i420"], want[-1]), t      # the checker reads the segment to its end_of_slice_segment_flag
        got += gd.decode_au(bytes(au), t)
    got += gd.drain() if frame_threads else []
    if not frame_threads:
        eos = bytes([0, 0, 0, 1, 36 << 1, 1])
        f = gd.decode_nal(eos)
        if f is not None:
            got.append(f)
    assert len(got) == 6, len(got)
    for t in range(6):
        assert np.array_equal(got[t]["i420"], want[t]), t
    assert gd.lib.kvzx_decoder_last_error(gd.h) == 0
    for x in (oe, gd, od):
        x.close()


@pytest.mark.gpu
def test_no_cropping_hands_out_the_coded_picture(gpu):
    """libOpenHevcSetNoCropping(h, 1): the conformance window is not applied -- the picture comes at its coded size, the cropped picture is its top left part"""
    from kvazzup_amd.codec import Decoder
    w, h = 200, 136
    oe = orc.OracleEncoder(w, h, qp=32, period=4, me_range=8)
    aus = [oe.encode(orc.synth_frame(0, 0x5EED0009, w, h, t)) for t in range(5)]
    oe.close()
    plain, full = Decoder(), Decoder(no_cropping=True)
    try:
        for t, au in enumerate(aus):
            a, b = plain.decode_au(au, t), full.decode_au(au, t)
            assert len(a) == len(b) == 1
            a, b = a[0], b[0]
            assert (a["width"], a["height"]) == (w, h) and b["width"] >= w and b["height"] >= h and b["width"] % 8 == 0 and b["height"] % 8 == 0 and (b["width"], b["height"]) != (w, h)
            W, H = b["width"], b["height"]
            ya, yb = a["i420"][:w * h].reshape(h, w), b["i420"][:W * H].reshape(H, W)
            assert np.array_equal(ya, yb[:h, :w])
            ua, ub = a["i420"][w * h:w * h * 5 // 4].reshape(h // 2, w // 2), b["i420"][W * H:W * H * 5 // 4].reshape(H // 2, W // 2)
            assert np.array_equal(ua, ub[:h // 2, :w // 2])
    finally:
        plain.close(); full.close()
