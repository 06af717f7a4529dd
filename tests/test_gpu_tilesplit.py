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
