"""GPU parity at the shapes BASELINE.json names (configs[0] .. configs[4]), each through the C ABI:
  configs[0]  1080p all-intra (period = 1), 30 pictures through the filter pipeline
  configs[1]  plain 1080p / period 64 / QP 32 against the checker, and a 130-picture run (two IDRs, POC wrap at 256 is in test_long_run)
  configs[3]  several independent streams at once (two pipelines on the one GPU of the test box)
  configs[4]  7680x4320 with 8 tile rows in band mode against the checker; 8 ranks sharing the GPU with the halo exchange
plus the pipelined encoder (owf 3) with SAO on the device path (the ordering the source-set events protect)."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5EED0002


@pytest.mark.gpu
def test_config1_plain_1080p_p64_qp32_matches_oracle(gpu):
    """BASELINE configs[1] as it stands: 1920x1080, period 64, QP 32, search range 16, nothing else switched on: access units and
    reconstruction of the HIP encoder == checker, HIP decoder == checker's decoder, 3 pictures (IDR + 2 P)"""
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = 1920, 1080
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16)
    od = orc.OracleDecoder()
    ge = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16)))
    gd = Decoder()
    try:
        for t in range(3):
            fr = orc.synth_frame(0, SEED, w, h, t)
            au_o = oe.encode(fr)
            au_g, rec_g = ge.encode(fr)
            assert au_g == au_o, t
            assert np.array_equal(rec_g, oe.recon()), t
            ref = od.decode_au(au_o, t); got = gd.decode_au(au_g, t)
            assert len(ref) == 1 and len(got) == 1 and np.array_equal(got[0]["i420"], ref[0]["i420"]), t
    finally:
        ge.close(); gd.close(); oe.close(); od.close()


@pytest.mark.gpu
@pytest.mark.parametrize("owf,threads", [(3, 12), (6, 32), (12, 24)])
def test_config1_long_run_two_idrs(gpu, owf, threads):
    """130 pictures of the 1080p / period-64 workload through the pipelined filters (owf 3 with 12 frame threads; the benchmark's
    owf 6 with 32: all eight working sets of the encoder in rotation, pictures queued behind the intra pictures' chains):
    three IDRs; EVERY access unit is decoded by the checker's decoder and every picture the pipelined HIP decoder delivered equals the
    checker's (second and third IDR, the rotation of the encoder's eight working sets and the owf queueing included); the first
    pictures' access units also equal the checker encoder's"""
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Decoder
    from kvazzup_amd.pipeline import Pipeline
    w, h, n = 1920, 1080, 130
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/OWF": owf, "video/OPENHEVC_threads": threads, "video/OH_parallelization": "Frame"},
                  custom=(("me-range", 16),))
    clip = [synth.frame(synth.MOVING, SEED, w, h, t) for t in range(4)]
    frames = [orc.synth_frame(0, SEED, w, h, t) for t in range(n)]
    assert all(np.array_equal(clip[t], frames[t]) for t in range(4))          # numpy twin == C twin
    for t in range(n):
        while pl.backlog() >= 8:                      # a uvgComm filter drops inputs when its buffer overflows (filter.cpp:151-222): pace the source
            time.sleep(0.001)
        pl.push(frames[t], t)
    pl.flush()
    assert pl.wait(n, 120000), pl.stats()
    aus = [pl.pop_encoded() for _ in range(n)]
    dec = [pl.pop_decoded() for _ in range(n)]
    pl.close()
    assert all(a is not None for a in aus) and all(d is not None for d in dec)
    assert [a[1] for a in aus] == list(range(n)) and [d["pts"] for d in dec] == list(range(n))
    idr = [t for t in range(n) if (aus[t][0][4] >> 1) == 32]                  # access units that start with a VPS
    assert idr == [0, 64, 128], idr
    od = orc.OracleDecoder()
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16)
    try:
        for t in range(n):
            ref = od.decode_au(aus[t][0], t)
            assert len(ref) == 1 and np.array_equal(ref[0]["i420"], dec[t]["i420"]), "picture %d differs from the checker's decoder" % t
            if t < 3:
                assert oe.encode(frames[t]) == aus[t][0], t
                assert np.array_equal(oe.recon(), dec[t]["i420"]), t
    finally:
        od.close(); oe.close()


@pytest.mark.gpu
def test_config0_all_intra_1080p_30_pictures(gpu):
    """BASELINE configs[0]: 1080p, period = 1, 30 pictures through KvazaarFilter' -> wire -> OpenHEVCFilter'; the first two access units
    against the checker's encoder, ALL thirty decoded by the checker's decoder and compared with what the HIP decoder delivered (PSNR sane)"""
    from kvazzup_amd.codec import Decoder
    from kvazzup_amd.pipeline import Pipeline
    w, h, n = 1920, 1080, 30
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 1}, custom=(("me-range", 16),))
    frames = [orc.synth_frame(0, SEED, w, h, t) for t in range(n)]
    for t in range(n):
        pl.push(frames[t], t)
        assert pl.wait(t + 1, 60000)
    aus = [pl.pop_encoded() for _ in range(n)]
    dec = [pl.pop_decoded() for _ in range(n)]
    pl.close()
    oe = orc.OracleEncoder(w, h, qp=32, period=1, me_range=16)
    od = orc.OracleDecoder()
    gd = od                                                                   # (closed in the finally clause below)
    try:
        for t in range(n):
            assert (aus[t][0][4] >> 1) == 32, t                               # every picture an IDR with parameter sets (vps-period 1)
            if t < 2:
                assert oe.encode(frames[t]) == aus[t][0], t
                assert np.array_equal(oe.recon(), dec[t]["i420"]), t
            ref = od.decode_au(aus[t][0], t)
            assert len(ref) == 1 and np.array_equal(ref[0]["i420"], dec[t]["i420"]), "picture %d differs from the checker's decoder" % t
            mse = np.mean((frames[t][:w * h].astype(float) - dec[t]["i420"][:w * h]) ** 2)
            assert 10 * np.log10(255.0 ** 2 / mse) > 33, t
    finally:
        gd.close(); oe.close()


@pytest.mark.gpu
def test_config3_two_streams_concurrently_on_one_gpu(gpu):
    """BASELINE configs[3] in miniature: two independent 1080p streams (a two-party call), each with its own encoder and decoder
    instance, running at the same time on the one GPU; both bit-exact against the checker"""
    from kvazzup_amd.pipeline import Pipeline
    w, h, n = 1920, 1080, 6
    seeds = (SEED, SEED + 16)
    pls = [Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/OWF": 2, "video/OPENHEVC_threads": 3, "video/OH_parallelization": "Frame"},
                    custom=(("me-range", 16),)) for _ in seeds]
    clips = [[orc.synth_frame(0, s, w, h, t) for t in range(n)] for s in seeds]
    for t in range(n):
        for k, pl in enumerate(pls):
            pl.push(clips[k][t], t)
    for pl in pls:
        pl.flush()
    for pl in pls:
        assert pl.wait(n, 120000), pl.stats()
    for k, pl in enumerate(pls):
        oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16)
        for t in range(n):
            au, pts = pl.pop_encoded()
            d = pl.pop_decoded()
            assert pts == t and d["pts"] == t
            assert au == oe.encode(clips[k][t]), (k, t)
            assert np.array_equal(d["i420"], oe.recon()), (k, t)
        oe.close()
        pl.close()


@pytest.mark.gpu
def test_config3_eight_party_call_on_one_gpu(gpu):
    """BASELINE configs[3] in its own shape, inside ONE process as uvgComm runs it (filtergraph.cpp:561-589: one OpenHEVCFilter per peer): eight independent
    1080p streams, each with its own KvazaarFilter' -> WireAdapter -> OpenHEVCFilter' chain, all at once on the one GPU -- eight decoders posting to the
    submission layer (csrc/batch.h: their pictures leave in shared launches), eight encoders on the shared streams.  Every access unit of every stream equals
    the checker encoder's, every decoded picture its reconstruction."""
    from kvazzup_amd.pipeline import Pipeline
    w, h, n, K = 1920, 1080, 5, 8
    seeds = [SEED + 16 * k for k in range(K)]
    pls = [Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/OWF": 2, "video/OPENHEVC_threads": 2, "video/OH_parallelization": "Frame", "video/kvzThreads": 2},
                    custom=(("me-range", 16),)) for _ in seeds]
    clips = [[orc.synth_frame(0, s, w, h, t) for t in range(n)] for s in seeds]
    try:
        for t in range(n):
            for k, pl in enumerate(pls):
                pl.push(clips[k][t], t)
        for pl in pls:
            pl.flush()
        for pl in pls:
            assert pl.wait(n, 240000), pl.stats()
        for k, pl in enumerate(pls):
            oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16)
            try:
                for t in range(n):
                    au, pts = pl.pop_encoded()
                    d = pl.pop_decoded()
                    assert pts == t and d["pts"] == t, (k, t)
                    assert au == oe.encode(clips[k][t]), "stream %d: access unit %d differs from the checker's" % (k, t)
                    assert np.array_equal(d["i420"], oe.recon()), "stream %d: decoded picture %d differs from the reconstruction" % (k, t)
            finally:
                oe.close()
    finally:
        for pl in pls:
            pl.close()


@pytest.mark.gpu
def test_config3_bench_command_with_two_ranks(gpu):
    """BASELINE configs[3] the way the driver runs it: `bench.py --gpus 2` (launch_ranks -> StreamRanks -> run_stream -> max over ranks), the two
    ranks sharing the test box's one GPU over gloo, four intra periods each at full rate (the processes time-slice the GPU: no wait in the intra chains may
    give up over that); the line must say two streams and carry a sane whole-job rate"""
    import json
    # (Rounds 4-5 saw this command crawl at 8 - 30 frames/s beside pytest -n 3 workers and retried it.  Root cause, round 6 (tools/measure/crawl_root_cause.sh,
    # profiles/r06_crawl_root_cause.txt): hardware-queue oversubscription ACROSS PROCESSES -- beside three foreign processes with eight busy streams each the
    # two ranks fall from 3 300 to 520 frames/s (short foreign kernels) or 46 (long ones), beside the same three with ONE stream each, chain kernels or not,
    # to nothing less; the spin-waits of the intra chains are not it.  GPU_MAX_HW_QUEUES=2 (the HIP runtime's own knob) is what helps on a GPU shared between
    # processes: 1 894 instead of 717 frames/s beside short foreign kernels, 444 instead of 41 beside long ones, at 22 % of the rate alone.  The
    # driver runs the suite alone: one attempt, the rate asserted.)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--repeats", "1",
                        "--no-cpu-baseline", "--no-host-boundary"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "device error" not in r.stderr, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["streams"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["metric"] == "hevc_encode_decode_fps" and line["unit"] == "frames/s" and line["steps"] == 4
    # two ranks time-slicing one GPU run at a third of one rank's rate, not at a three-hundredth
    assert 100.0 < line["value"] < 100000.0, (line["value"], line["config"].get("host_cpu_cores_busy"), line["config"].get("host_cpu_throttled_ms"))
    assert "error flags" not in r.stderr, r.stderr[-3000:]
    assert abs(line["value"] - 2 * 64 / (line["ms_per_step"] / 1e3)) < 1.0          # whole-job frames per second: both ranks' pictures over the slowest rank's time
    assert 30.0 < line["config"]["psnr_y"] < 50.0


@pytest.mark.gpu
def test_config4_8k_eight_tile_rows_band_mode(gpu):
    """BASELINE configs[4] at its own shape in one process: 7680x4320, tiles = 1x8, the band path (phase 1 / halo / phase 2) over the
    whole picture; an IDR and a P picture against the checker's encoder with the same tiling"""
    import ctypes as C
    from kvazzup_amd.tilesplit import BandEncoder
    w, h = 7680, 4320
    hip = C.CDLL("libamdhip64.so")
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(w * h * 3 // 2)) == 0
    be = BandEncoder(w, h, 8, 0, 1, options=(("qp", 32), ("period", 64), ("me-range", 16)))
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16, tile_rows=8)
    try:
        for t in range(2):
            fr = orc.synth_frame(0, 0x5EED0005, w, h, t)
            assert hip.hipMemcpy(dptr, C.c_void_p(fr.ctypes.data), C.c_size_t(fr.size), 1) == 0
            assert be.encode(dptr) == oe.encode(fr), t
    finally:
        be.close(); oe.close(); hip.hipFree(dptr)


@pytest.mark.gpu
def test_config4_eight_ranks_share_the_gpu(gpu):
    """8 ranks (gloo, all on the test box's one GPU), one tile row each, with the halo exchange: 1920x1088, tiles 1x8"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29871", os.path.join(ROOT, "tests", "run_tilesplit.py"), "1920", "1088", "8", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("owf", [3, 6])
def test_default_mode_pipelined_with_intra_pictures_on_the_side_stream(gpu, owf):
    """uvgComm's default mode (preset veryfast: SAO, subme 2, 16x16 intra units in P pictures; 1 Mbit/s with rc-algorithm lambda: per-CTU QPs and the row groups
    of rate control v2) with pictures in flight and an intra period of 12: every intra picture's chain runs on the side stream BESIDE the P pictures in front
    of it -- its own progress counters, edge columns and SAO work picture, the rate control state updated in picture order on the main stream -- and the access
    units are still the synchronous checker's (rc-delay = pictures in flight + 1)"""
    from kvazzup_amd import synth
    from kvazzup_amd.codec import Decoder, Encoder
    w, h, n, period, br = 1280, 720, 40, 12, 1000000
    ge = Encoder(w, h, options=(("preset", "veryfast"), ("qp", 32), ("period", period), ("me-range", 16), ("owf", owf), ("bitrate", br), ("rc-algorithm", "lambda")),
                 fields={"target_bitrate": br})
    assert not ge.rejected, ge.rejected
    oe = orc.OracleEncoder(w, h, qp=32, period=period, me_range=16, sao=1, subme=2, bitrate=br, rc_bands=4)
    oe.set_option("intra-in-p", 1); oe.set_option("rc-delay", min(owf, 6) + 1); oe.set_option("me-source", 1)      # (preset veryfast: the search on the input picture)
    od = orc.OracleDecoder(); gd = Decoder(threads=4, frame_threads=True)
    frames = [synth.scene_cut_frame(SEED, w, h, t, 17) for t in range(n)]
    aus = []
    for t in range(n + owf + 1):
        out = ge.encode(frames[t] if t < n else None, want_recon=False)      # (None: pic_in == NULL hands out a picture still in flight)
        if out[0]:
            aus.append(out[0])
    assert len(aus) == n, len(aus)
    got = []
    for t in range(n):
        want = oe.encode(frames[t])
        assert aus[t] == want, (t, len(aus[t]), len(want))
        ref = od.decode_au(aus[t], t)
        assert len(ref) == 1 and np.array_equal(ref[0]["i420"], oe.recon()), t
        got += gd.decode_au(aus[t], t)
    got += gd.drain()
    assert len(got) == n
    for x in (ge, gd, oe, od):
        x.close()


@pytest.mark.gpu
def test_pipelined_encoder_with_sao_on_device_path_matches_oracle(gpu):
    """owf = 3 with sao = full at 1080p, pictures handed over in HBM back to back: k_sao of picture t reads the source picture while
    the input stage already prepares picture t + 2 -- the access units must still be the checker's (synchronous) ones"""
    import ctypes as C
    from kvazzup_amd.codec import Encoder
    w, h, n = 1920, 1080, 8
    ge = Encoder(w, h, options=(("qp", 32), ("period", 64), ("me-range", 16), ("sao", "full"), ("owf", 3)))
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16, sao=1)
    hip = C.CDLL("libamdhip64.so")
    frames = [orc.synth_frame(0, SEED, w, h, t) for t in range(n)]
    dptrs = []
    for fr in frames:
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(fr.size)) == 0
        assert hip.hipMemcpy(p, C.c_void_p(fr.ctypes.data), C.c_size_t(fr.size), 1) == 0
        dptrs.append(p)
    try:
        aus = []
        for t in range(n):
            au = ge.encode_device(dptrs[t])
            if au:
                aus.append(au)
        while len(aus) < n:
            au = ge.encode_device(None)
            assert au
            aus.append(au)
        for t in range(n):
            assert aus[t] == oe.encode(frames[t]), t
    finally:
        ge.close(); oe.close()
        for p in dptrs:
            hip.hipFree(p)


@pytest.mark.gpu
def test_config2_4k_decode_matches_the_checkers_decoder(gpu):
    """BASELINE configs[2] size on the DECODER side (the bench's `secondary` leg times it): 3840x2160, IDR + 2 P of the plain workload, then one
    default-mode picture pair (preset veryfast: SAO, fractional search, intra units in P pictures) -- every picture of the HIP decoder equals
    oracle/hevc_dec.c's, sample for sample (and the encoder's reconstruction: closed loop at 4K)"""
    from kvazzup_amd.codec import Decoder, Encoder
    w, h = 3840, 2160
    for name, opts, frames in (("plain", (("qp", 32), ("period", 64), ("me-range", 16)), 3),
                               ("default-mode", (("preset", "veryfast"), ("qp", 32), ("period", 64), ("me-range", 16)), 2)):
        ge = Encoder(w, h, options=opts)
        assert not ge.rejected, ge.rejected
        od = orc.OracleDecoder()
        gd = Decoder()
        try:
            for t in range(frames):
                au, rec = ge.encode(orc.synth_frame(0, SEED, w, h, t))
                ref = od.decode_au(au, t); got = gd.decode_au(au, t)
                assert len(ref) == 1 and len(got) == 1, (name, t)
                assert np.array_equal(got[0]["i420"], ref[0]["i420"]), "%s: 4K picture %d differs from the checker's decoder" % (name, t)
                assert np.array_equal(got[0]["i420"], rec), "%s: 4K picture %d differs from the encoder's reconstruction" % (name, t)
        finally:
            ge.close(); gd.close(); od.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,name", [(1, "flat"), (2, "noise")])
def test_bound_clips_1080p_through_the_filter_chain(gpu, kind, name):
    """SURVEY 8(d)'s two bound clips at BASELINE configs[1]'s shape through KvazaarFilter' -> WireAdapter -> OpenHEVCFilter': `flat` (everything
    skipped: substreams of a few bytes) and `noise` (every block searched, dense coefficients, ~10x the bins of the moving-objects clip).  Access
    units == the checker encoder's, decoded pictures == its reconstruction, 4 pictures (IDR + 3 P)"""
    from kvazzup_amd.pipeline import Pipeline
    w, h, frames = 1920, 1080, 4
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=16)
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64}, custom=(("me-range", 16),))
    try:
        clip = [orc.synth_frame(kind, SEED, w, h, t) for t in range(frames)]
        for f in clip:
            pl.push(f)
        assert pl.wait(frames, 120000)
        for t in range(frames):
            au_o = oe.encode(clip[t])
            au_g, pts = pl.pop_encoded()
            assert pts == t and au_g == au_o, "%s: AU %d differs (%d vs %d bytes)" % (name, t, len(au_g), len(au_o))
            d = pl.pop_decoded()
            assert d["pts"] == t and np.array_equal(d["i420"], oe.recon()), "%s: decoded picture %d differs from the reconstruction" % (name, t)
    finally:
        pl.close(); oe.close()
