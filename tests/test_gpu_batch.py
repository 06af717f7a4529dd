"""Several codec instances in ONE process on one GPU -- uvgComm's multi-party topology: one OpenHEVCFilter per peer beside the shared
KvazaarFilter (/root/reference/src/media/processing/filtergraph.cpp:347-351,561-589).  The instances share the device's HIP streams by role
(csrc/stream_pool.h) and the decoders' pictures are launched by the device's submission layer, same kernels of different decoders' pictures
as ONE launch (csrc/batch.h).  Every decoder's output must equal the checker's reconstruction of its own stream, bit for bit, whatever
shares a launch with it."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import orc

SEED = 0x5EED0040


def _stream(w, h, n, seed, scene_cut=None, **enc):
    """n pictures of the synthetic clip coded by the checker's encoder: (access units, reconstructions)"""
    from kvazzup_amd import synth
    opts = {k: enc.pop(k) for k in ("intra_in_p", "rdoq", "signhide") if k in enc}
    oe = orc.OracleEncoder(w, h, **enc)
    for k, v in opts.items():
        oe.set_option(k.replace("_", "-"), v)
    aus, recs = [], []
    for t in range(n):
        fr = synth.scene_cut_frame(seed, w, h, t, scene_cut) if scene_cut is not None else orc.synth_frame(0, seed, w, h, t)
        aus.append(oe.encode(fr))
        recs.append(oe.recon())
    oe.close()
    return aus, recs


def _batch_stats(lib, reset=False):
    lib.kvzx_batch_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    b, p = C.c_uint64(), C.c_uint64()
    sizes, ms, ln, fr = (C.c_uint64 * 9)(), (C.c_double * 4)(), (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
    k = lib.kvzx_batch_stats(0, C.byref(b), C.byref(p), sizes, ms, ln, fr, int(reset))
    return {"batches": b.value, "pictures": p.value, "sizes": list(sizes), "ms": list(ms)[:k], "launches": list(ln)[:k], "frames": list(fr)[:k]}


def _decode_all(dec, aus, out, err, k):
    try:
        got = []
        for t, au in enumerate(aus):
            got += dec.decode_au(au, t)
        got += dec.drain()
        out[k] = got
    except Exception as e:          # noqa: BLE001 -- reported by the asserting thread
        err[k] = e


STREAMS = [
    dict(w=832, h=480, n=7, qp=30, period=64, me_range=8),
    dict(w=640, h=384, n=7, qp=27, period=4, me_range=8, sao=1, subme=2),                                     # fractional vectors, SAO, a second IDR
    dict(w=1280, h=720, n=6, qp=32, period=64, me_range=8, tile_rows=2, tile_cols=2, qp_in_cu=1, vaq=8, intra_in_p=1, scene_cut=3),   # tiles, cu_qp_delta, intra units in P pictures
]


@pytest.fixture(scope="module")
def coded():
    out = []
    for i, s in enumerate(STREAMS):
        s = dict(s)
        w, h, n = s.pop("w"), s.pop("h"), s.pop("n")
        out.append((w, h) + _stream(w, h, n, SEED + 16 * i, **s))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("frame_threads", [1, 4])
def test_three_decoders_beside_an_encoder_in_one_process(gpu, coded, frame_threads):
    """1 encoder + 3 decoders (a four-party call) running at the same time on the one GPU: three streams of different sizes and tool sets go through
    three decoder instances on three threads while the HIP encoder codes a fourth clip; every decoded picture equals the checker's
    reconstruction and the encoder's access units equal the checker encoder's"""
    from kvazzup_amd.codec import Decoder, Encoder
    decs = [Decoder(threads=frame_threads, frame_threads=frame_threads > 1) for _ in coded]
    lib = decs[0].lib
    _batch_stats(lib, reset=True)
    out, err = [None] * len(coded), [None] * len(coded)
    ths = [threading.Thread(target=_decode_all, args=(decs[k], coded[k][2], out, err, k)) for k in range(len(coded))]
    w, h, n = 832, 480, 6
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=8)
    ge = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 8)))
    for t in ths:
        t.start()
    try:
        for t in range(n):
            fr = orc.synth_frame(0, SEED + 99, w, h, t)
            au_g, rec_g = ge.encode(fr)
            assert au_g == oe.encode(fr), t
            assert np.array_equal(rec_g, oe.recon()), t
    finally:
        for t in ths:
            t.join()
        ge.close(); oe.close()
    st = _batch_stats(lib)
    for d in decs:
        d.close()
    for k, (w, h, aus, recs) in enumerate(coded):
        assert err[k] is None, (k, err[k])
        assert len(out[k]) == len(recs), (k, len(out[k]))
        for t, (g, r) in enumerate(zip(out[k], recs)):
            assert g["width"] == w and g["height"] == h
            assert np.array_equal(g["i420"], r), (k, t)
    # with three decoders open every picture goes through the submission layer
    assert st["pictures"] == sum(len(c[3]) for c in coded), st


@pytest.mark.gpu
def test_full_batches_are_bit_exact(gpu, coded):
    """the submission layer is held while four frame-threaded decoders (two of them on the same stream) queue their pictures, then released: the
    pictures leave in batches of four -- IDR pictures of different sizes in one k_dec_intra launch, P pictures with and without SAO / intra units in one
    k_dec_inter launch -- and every decoder's output still equals the checker's reconstruction"""
    from kvazzup_amd.codec import Decoder
    plan = [0, 1, 2, 0]
    decs = [Decoder(threads=4, frame_threads=True) for _ in plan]
    lib = decs[0].lib
    lib.kvzx_batch_hold.argtypes = [C.c_int, C.c_int]
    _batch_stats(lib, reset=True)
    for d in decs:
        d.set_profiling(1)
    out, err = [None] * len(plan), [None] * len(plan)
    lib.kvzx_batch_hold(0, 1)
    ths = [threading.Thread(target=_decode_all, args=(decs[k], coded[plan[k]][2], out, err, k)) for k in range(len(plan))]
    try:
        for t in ths:
            t.start()
        time.sleep(1.0)                       # every decoder has queued what its ring lets it queue and waits for the first picture
    finally:
        lib.kvzx_batch_hold(0, 0)
        for t in ths:
            t.join()
    st = _batch_stats(lib)
    for d in decs:
        d.close()
    for k, s in enumerate(plan):
        w, h, aus, recs = coded[s]
        assert err[k] is None, (k, err[k])
        assert len(out[k]) == len(recs), (k, len(out[k]))
        for t, (g, r) in enumerate(zip(out[k], recs)):
            assert np.array_equal(g["i420"], r), (k, t)
    assert st["sizes"][4] >= 2, st                       # at least the first pictures (IDR) and the second ones left four at a time
    assert st["launches"][0] > 0 and st["frames"][0] > st["launches"][0], st      # k_dec_inter_n: more pictures than launches


@pytest.mark.gpu
def test_decoder_that_stays_open_when_the_others_close(gpu, coded):
    """a decoder goes on alone after its neighbours have closed (a peer leaves the call): it changes from the submission layer to launching for
    itself in mid-stream, in order"""
    from kvazzup_amd.codec import Decoder
    w, h, aus, recs = coded[0]
    a, b = Decoder(threads=4, frame_threads=True), Decoder(threads=1)
    got = []
    for t, au in enumerate(aus):
        got += a.decode_au(au, t)
        if t == 2:
            b.decode_au(coded[1][2][0], 0)
            b.close()
    got += a.drain()
    a.close()
    assert len(got) == len(recs)
    for t, (g, r) in enumerate(zip(got, recs)):
        assert np.array_equal(g["i420"], r), t
