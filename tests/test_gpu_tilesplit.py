"""Tile-row split of one picture over several encoder instances: band mode in one process, and two
ranks sharing the box's GPU over torch.distributed (gloo) with the halo exchange -- both bit-exact against the CPU checker."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tile_rows", [(320, 256, 2), (256, 448, 3)])
def test_band_mode_single_process(gpu, w, h, tile_rows):
    """one band = the whole picture: the phase1 / phase2 path without any exchange gives the checker's access units"""
    import ctypes as C
    from kvazzup_amd.tilesplit import BandEncoder
    hip = C.CDLL("libamdhip64.so")
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(w * h * 3 // 2)) == 0
    be = BandEncoder(w, h, tile_rows, 0, 1, options=(("qp", 30), ("period", 4), ("me-range", 16)))
    oe = orc.OracleEncoder(w, h, qp=30, period=4, me_range=16, tile_rows=tile_rows)
    for t in range(6):
        frame = orc.synth_frame(0, 11, w, h, t)
        assert hip.hipMemcpy(dptr, C.c_void_p(frame.ctypes.data), C.c_size_t(frame.size), 1) == 0
        assert be.encode(dptr) == oe.encode(frame), t
    be.close(); oe.close()
    hip.hipFree(dptr)


@pytest.mark.gpu
def test_band_mode_rate_control_single_process(gpu):
    """bitrate > 0 in band mode: the band's encoder is told the assembled sizes (kvzx_encoder_band_report_au) and follows the checker's
    controller QP for QP"""
    import ctypes as C
    from kvazzup_amd.tilesplit import BandEncoder
    w, h, tile_rows, bitrate = 320, 256, 2, 300000
    hip = C.CDLL("libamdhip64.so")
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(w * h * 3 // 2)) == 0
    be = BandEncoder(w, h, tile_rows, 0, 1, options=(("qp", 30), ("period", 16), ("me-range", 16), ("bitrate", bitrate)))
    oe = orc.OracleEncoder(w, h, qp=30, period=16, me_range=16, tile_rows=tile_rows, bitrate=bitrate)
    for t in range(24):
        frame = orc.synth_frame(0, 11, w, h, t)
        assert hip.hipMemcpy(dptr, C.c_void_p(frame.ctypes.data), C.c_size_t(frame.size), 1) == 0
        assert be.encode(dptr) == oe.encode(frame), t
    be.close(); oe.close()
    hip.hipFree(dptr)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tile_rows,ranks,mode,bitrate", [(320, 256, 2, 2, "sync", 0), (256, 448, 4, 2, "pipelined", 0), (1920, 1088, 4, 2, "pipelined", 0),
                                                              (320, 256, 2, 2, "sync", 300000), (320, 256, 4, 2, "pipelined", 300000)])
def test_two_ranks_with_halo_exchange(gpu, w, h, tile_rows, ranks, mode, bitrate):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (w + h + tile_rows + bitrate // 1000) % 400), os.path.join(ROOT, "tests", "run_tilesplit.py"), str(w), str(h), str(tile_rows), "20" if bitrate else "6", mode, str(bitrate)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


# ---------------------------------------------------------------------------------------------- the split DECODER
def _nals(au):
    from kvazzup_amd.codec import split_nals
    return list(split_nals(au))


def _band_rows_equal(pic, want, w, h):
    y0, y1 = pic["rows"]
    got = pic["i420"]
    Y, Yw = got[:w * h].reshape(h, w), want[:w * h].reshape(h, w)
    assert np.array_equal(Y[y0:y1], Yw[y0:y1]), "luma rows %d..%d" % (y0, y1)
    for c in range(2):
        o = w * h + c * (w * h // 4)
        P, Pw = got[o:o + w * h // 4].reshape(h // 2, w // 2), want[o:o + w * h // 4].reshape(h // 2, w // 2)
        assert np.array_equal(P[y0 // 2:y1 // 2], Pw[y0 // 2:y1 // 2]), "chroma %d rows %d..%d" % (c, y0 // 2, y1 // 2)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tile_rows,bands,subme", [(320, 256, 2, 2, 0), (256, 448, 4, 2, 2), (256, 448, 3, 3, 4), (1920, 1088, 4, 4, 2), (1920, 1088, 8, 8, 0)])
def test_split_decoder_bands_in_one_process(gpu, w, h, tile_rows, bands, subme):
    """every band decoder gets every NAL unit, parses and reconstructs its own tile rows only; boundary rows go down before the
    deblocking and come back up after it.  Each band's rows equal the checker's reconstruction."""
    from kvazzup_amd.tilesplit import BandDecoder, finish_bands_local
    oe = orc.OracleEncoder(w, h, qp=28, period=4, me_range=16, tile_rows=tile_rows, subme=subme)
    decs = [BandDecoder((h + 63) // 64, tile_rows, r, bands) for r in range(bands)]
    assert sum(d.nrows for d in decs) == (h + 63) // 64
    for t in range(7):
        au = oe.encode(orc.synth_frame(0, 23, w, h, t))
        want = oe.recon()
        for nal in _nals(au):
            ready = [d.feed(nal, t) for d in decs]
        assert all(ready)
        for pic in finish_bands_local(decs):
            assert (pic["width"], pic["height"]) == (w, h)
            _band_rows_equal(pic, want, w, h)
    for d in decs:
        d.close()
    oe.close()


@pytest.mark.gpu
def test_split_decoder_refuses_vectors_that_leave_the_band(gpu):
    """tiles alone do not confine motion vectors (the synthesiser's streams do not): a band decoder says so instead of predicting from
    rows it does not hold"""
    from kvazzup_amd.tilesplit import BandDecoder
    w, h = 256, 256
    gen = orc.OracleGen(width=w, height=h, seed=31, density=40, tile_rows=2, wpp=0, sao=0, tmvp=0, big_mvd=1)
    d = BandDecoder(4, 2, 1, 2)
    refused = False
    for _ in range(6):
        for nal in _nals(gen.picture()):
            try:
                ready = d.feed(nal)
            except RuntimeError:
                refused = True
                break
            if ready is True:                      # (no upper neighbour here: its rows stay as they are -- only the refusal matters)
                d.lib.kvzx_decoder_band_deblock(d.h); d.lib.kvzx_decoder_band_finish(d.h)
        if refused:
            break
    gen.close(); d.close()
    assert refused


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tile_rows,ranks", [(320, 256, 2, 2), (1920, 1088, 4, 2)])
def test_split_decoder_two_ranks(gpu, w, h, tile_rows, ranks):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + (w + h + tile_rows) % 90), os.path.join(ROOT, "tests", "run_tilesplit_dec.py"), str(w), str(h), str(tile_rows), "6"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK" in r.stdout
