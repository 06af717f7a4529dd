#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: HEVC encode+decode fps on synthetic YUV420 (uvgx-synth-v1).

A "step" is ONE INTRA PERIOD -- 64 pictures, the first an IDR -- through the hot path: kvz_api-side
encode (HIP kernels; input pictures already resident in HBM) and libOpenHevc-side decode of the access
units (host CABAC parse + HIP reconstruction, output left in HBM).  The two codecs sit in the C++ mirrors
of uvgComm's KvazaarFilter and OpenHEVCFilter, each on its own thread as in the reference's filter graph.
Whatever --steps says, the timed region therefore holds IDR and P pictures in the workload's own
proportion (1 : 63).  The clock starts on an EMPTY, flushed pipeline and stops when the last timed picture
has left the decoder (the pipeline is flushed again): nothing is in flight across either end.
`value` stays frames/s; `config.pictures_per_step` = 64.

--gpus N (N > 1) without a torch.distributed environment: this process launches N fresh rank processes
(before anything here touches torch or the GPU) and forwards rank 0's line.  N > 1 runs one independent
stream per GPU (BASELINE configs[3]: multi-party call, no collective on the data path), weak scaling;
with fewer devices than ranks the ranks share devices and the barrier / max-over-ranks go through gloo.

Prints ONE JSON line on rank 0 (see the contract in the task description / DESIGN.md section 6).
"""
import argparse
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PERIOD = 64                    # pictures per step = the intra period of the named workloads
CLIP_FRAMES = 128              # the named clips are 128 pictures long; longer runs cycle them (the wrap falls on an IDR)
WORKLOADS = {
    # BASELINE.json configs[1]: 1080p, preset=ultrafast, intra period 64, encode + decode on one GPU
    "1080p": dict(w=1920, h=1080, name="1080p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=2),
    # configs[2]: 4K encode (decode is run too; reported in the same fps)
    "4k": dict(w=3840, h=2160, name="2160p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=3),
    "720p": dict(w=1280, h=720, name="720p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=2),
    # configs[4]: ONE 8K stream, its 8 tile rows split over the ranks (strong scaling; see tilesplit_main)
    "8k-tilesplit": dict(w=7680, h=4320, name="4320p-yuv420-ultrafast-p64-qp32-encode-tile-row-split", cfg_index=5),
}
# custom parameters of the host-boundary legs (uvgComm's INI list "parameters", kvazaarfilter.cpp:351-371): the reconstruction is not downloaded
# (uvgComm frees it unread, :476) and encoder_encode(NULL) only returns pictures that are finished (the loop at :440-448 then keeps video/OWF
# pictures in flight instead of emptying the pipeline after every picture) -- INTEGRATION.md
HOST_CUSTOM = (("recon-output", "0"), ("null-input", "poll"))
HBM_PEAK_GBS = 8000.0          # replaced by the device's own figure in main(); this is the guide's (MI355X_MICROARCH.md) and the fallback
HBM_PEAK_SOURCE = "MI355X_MICROARCH.md (the runtime reported no memory clock / bus width)"          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes(kernel, cw, ch, me_range):
    """Compulsory bytes of ONE launch, P = coded luma samples."""
    P = cw * ch
    if kernel == "k_me":                          # current block once + its search window once, per 32x32 block
        return (P // 1024) * (1024 + (32 + 2 * me_range) ** 2)
    if kernel in ("k_inter_recon", "k_dec_inter", "k_inter_recon<dec>"):
        return int(4.5 * P) if kernel == "k_inter_recon" else int(3.0 * P)
    if kernel in ("k_intra_recon", "k_dec_intra", "k_intra_recon<dec>"):  # source in + reconstruction out (+ the level words, counted with k_tokenize)
        return int(3.0 * P) if kernel == "k_intra_recon" else int(1.5 * P)
    if kernel == "k_intra_analyse":
        return P
    if kernel in ("k_deblock", "k_dec_deblock"):
        return int(3.0 * P)
    if kernel == "k_tokenize":                    # every level of the picture once (int16) + the per-8x8 CU records; tokens out not counted
        return int(3.0 * P) + (P // 64) * 11
    if kernel == "k_tok_compact":                 # the piece table of every CTU ([16 units][17 pieces] {offset, length}); tokens not counted
        return (P // 4096) * 16 * 17 * 8
    if kernel in ("k_sao", "k_dec_sao", "k_sao<dec>"):          # deblocked picture in, filtered picture out (+ the source picture for the statistics)
        return int(4.5 * P) if kernel == "k_sao" else int(3.0 * P)
    if kernel == "k_pad_input":
        return int(3.0 * P)
    if kernel == "k_inter_signal":
        return (P // 64) * 16
    return P


def cpu_budget(world):
    """CPU cores this rank may use: the container's CFS quota (cgroup v2 cpu.max) or the visible cores, shared by the ranks of
    the node."""
    cores = float(len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, float(q) / float(per))
    except Exception:
        pass
    return cores / max(1, world)


def _throttled_us():
    """time the container's CPU quota has stalled the job so far (cgroup v2 cpu.stat), microseconds; 0 when unknown"""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1])
    except Exception:
        pass
    return 0


def _thread_cpu():
    """{tid: (name, CPU seconds)} of this process's threads"""
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (f[f.index("(") + 1:f.rindex(")")], (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:
            pass
    return out


# ---------------------------------------------------------------------------------------------------------------
# cpu_baseline: the CPU checker (oracle/, a scalar C port of the same algorithm) on ALL host cores -- one
# independent clip per core, each in its own process (the port has no threads of its own; a multi-party call is
# independent streams anyway).  This is the only place bench.py touches oracle/.
# ---------------------------------------------------------------------------------------------------------------
def cpu_worker(w, h, frames, me_range, seed):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=me_range)
    od = orc.OracleDecoder()
    clip = [orc.synth_frame(0, seed, w, h, t) for t in range(frames)]
    print("ready", flush=True)
    sys.stdin.readline()                      # all workers start together
    t0 = time.time()
    n = 0
    for t, fr in enumerate(clip):
        au = oe.encode(fr)
        n += len(od.decode_au(au, t))
    print("done %d %.6f" % (n, time.time() - t0), flush=True)


def cpu_baseline(w, h, frames, me_range, cores):
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d,%d,%d,%d,%d" % (w, h, frames, me_range, 0x5EED0002 + 16 * i)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for i in range(cores)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu worker failed to start")
        t0 = time.time()
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        n = 0
        for p in procs:
            tok = p.stdout.readline().split()
            if len(tok) != 3 or tok[0] != "done":
                raise RuntimeError("cpu worker failed")
            n += int(tok[1])
        dt = time.time() - t0
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            p.wait()
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d clips (one per core, one process each) x %d pictures %dx%d (1 intra + %d inter, search range %d), encode+decode by oracle/ (scalar C port)"
                      % (cores, frames, w, h, frames - 1, me_range)}


# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """--gpus N without a torch.distributed environment: N fresh processes, one per rank (this process has not imported torch
    or touched the GPU); rank 0 prints the line.  A rank that dies takes the others with it: they would otherwise sit in a
    barrier until the process group's timeout."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = max(rc, abs(code))
        time.sleep(0.05)
    for p in live:                                   # a rank failed: stop the rest
        p.terminate()
    for p in live:
        try:
            p.wait(10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def stream_seed(cfg_index, rank):
    """every rank codes its own synthetic stream (uvgx-synth-v1: seed = 0x5EED0000 + configuration, shifted per stream)"""
    return 0x5EED0000 + cfg_index + 16 * rank


def init_dist(world, local_rank):
    """device of this rank and the process group that brackets the timed region (no collective on the data path) -- tile-row split
    workload: device tensors travel between the ranks, so this one runs on torch / RCCL"""
    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("no GPU visible: this library has no CPU fallback")
    dev_index = local_rank % ndev
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ndev >= world:
            backend = "nccl"
            dist.init_process_group("nccl", device_id=dev)
        else:                                                     # ranks share devices: RCCL wants one device per rank
            backend = "gloo"
            dist.init_process_group("gloo")
    return torch, dist, dev, dev_index, backend


def barrier_max(dist, backend, dev, torch, value=None):
    """barrier (value None) or max over ranks of `value`"""
    if backend is None:
        return value
    if value is None:
        dist.barrier()
        return None
    tt = torch.tensor([value], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


class StreamRanks:
    """The stream workloads (one independent stream per rank, BASELINE configs[1..3]): nothing travels between the ranks, so the only
    thing the process group does is bracket the timed region -- a barrier and a max over ranks, on CPU tensors over gloo.  torch.cuda is
    never initialised here: the library is loaded FIRST and runs on the system's HIP runtime; the copy of the runtime that torch ships
    stays dormant (with torch.cuda initialised this library would run on torch's copy, whose device-to-host copies are blit kernels that
    slow every kernel beside them -- DESIGN.md section 6).  With one rank torch is not imported at all."""

    def __init__(self, world, local_rank, need_device=True):
        from kvazzup_amd import _native
        self.lib = _native.load_library()                     # before any import of torch
        ndev = self.lib.kvzx_device_count()
        if ndev < 1 and need_device:                          # (need_device=False: the CPU test of the process-group plumbing)
            raise RuntimeError("no GPU visible: this library has no CPU fallback")
        self.dev_index = local_rank % max(1, ndev)
        self.backend, self.dist, self.torch = None, None, None
        if world > 1:
            import datetime
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
            self.backend, self.dist, self.torch = "gloo", dist, torch

    def sync(self, value=None):
        if self.backend is None:
            return value
        if value is None:
            self.dist.barrier()
            return None
        tt = self.torch.tensor([value], dtype=self.torch.float64)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class DeviceClip:
    """the synthetic clip in device memory (kvzx_harness_*: generated on the GPU, no tensor library)"""

    def __init__(self, lib, dev_index, seed, w, h, frames):
        import ctypes as C
        lib.kvzx_harness_alloc.restype = C.c_void_p
        lib.kvzx_harness_alloc.argtypes = [C.c_int, C.c_size_t]
        lib.kvzx_harness_free.argtypes = [C.c_void_p]
        lib.kvzx_harness_synth_frame.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]
        lib.kvzx_harness_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.lib, self.dev, self.w, self.h, self.n = lib, dev_index, w, h, w * h * 3 // 2
        self.ptr = []
        for t in range(frames):
            p = lib.kvzx_harness_alloc(dev_index, self.n)
            if not p or not lib.kvzx_harness_synth_frame(p, 0, seed & 0xFFFFFFFF, w, h, t):
                raise RuntimeError("device clip: allocation or synthesis failed")
            self.ptr.append(p)
        if not lib.kvzx_harness_sync(dev_index):
            raise RuntimeError("device clip: synthesis failed")

    def host(self, t):
        import numpy as np
        a = np.empty(self.n, dtype=np.uint8)
        if not self.lib.kvzx_harness_download(a.ctypes.data, self.ptr[t], self.n):
            raise RuntimeError("device clip: download failed")
        return a

    def close(self):
        for p in self.ptr:
            self.lib.kvzx_harness_free(p)
        self.ptr = []


def tilesplit_main(args):
    """BASELINE configs[4]: a single 8K picture stream, 8 full-width tile rows, split over the ranks (whole tile rows per
    rank); the only exchange on the data path is the deblock halo (kvazzup_amd/tilesplit.py).  Strong scaling.  `value` is the split
    encoder; the split decoder on the same stream follows as `secondary`.
    A step = one picture here (the configuration is 'one 7680x4320 frame, repeated for timing')."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch, dist, dev, dev_index, backend = init_dist(world, local_rank)
    from kvazzup_amd import synth
    from kvazzup_amd.tilesplit import BandEncoder
    wl = WORKLOADS["8k-tilesplit"]
    w, h, tile_rows = wl["w"], wl["h"], 8
    total = args.warmup + args.steps
    nclip = min(total, 32)
    clip = [synth.frame_torch(synth.MOVING, 0x5EED0005, w, h, t, dev) for t in range(nclip)]     # every rank holds the stream (a band only reads its rows)
    torch.cuda.synchronize()
    coder_threads = max(1, min(16, int(cpu_budget(world)) - 1))     # the ranks of a node share its cores: each sizes its arithmetic-coder pool to its share
    be = BandEncoder(w, h, tile_rows, rank, world, options=(("qp", 32), ("period", 64), ("me-range", args.me_range), ("threads", coder_threads)), device=dev_index,
                     dist=dist if world > 1 else None, pipelined=True)     # the gather of picture t completes during picture t + 1
    nbytes = 0
    aus = []                                                          # rank 0: the access units, for the split decoder's leg
    for t in range(args.warmup):
        au = be.encode(clip[t % nclip].data_ptr())
        if au is not None:
            aus.append(au)
    torch.cuda.synchronize()
    barrier_max(dist, backend, dev, torch)
    if hasattr(be, "times"):
        be.times.clear()                                              # (KVAZZUP_BENCH_TILESPLIT_TIMES: the timed pictures only)
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        au = be.encode(clip[t % nclip].data_ptr())
        if au is not None:
            nbytes += len(au); aus.append(au)
    au = be.flush()
    if au is not None:
        nbytes += len(au); aus.append(au)
    torch.cuda.synchronize()
    barrier_max(dist, backend, dev, torch)
    elapsed = barrier_max(dist, backend, dev, torch, time.perf_counter() - t0)
    decode = None if args.no_split_decode else tilesplit_decode(args, aus, h, tile_rows, rank, world, torch, dist, dev, dev_index, backend)
    if rank == 0:
        print(json.dumps({
            "metric": "hevc_encode_fps_one_stream_tile_row_split", "value": round(args.steps / elapsed, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"], "width": w, "height": h, "tile_rows": tile_rows, "ranks": world, "ctu_rows_rank0": be.nrows,
                       "pictures_per_step": 1, "collective_backend": backend,
                       "intra_period": 64, "qp": 32, "me_range": args.me_range, "bytes_per_frame": round(nbytes / args.steps, 1),
                       "halo_bytes_per_picture_and_rank": round(be.halo_bytes_exchanged / max(1, total), 1),
                       "exchange": "2 halo blocks per internal boundary and picture (4 luma + 2x2 chroma rows + CU records), send/recv, in flight during the tokenizer and the arithmetic coder; substreams: fixed-size all_gather of the headers + padded gather of the payloads, completed during the next picture"},
            "secondary": decode, "roofline": None, "cpu_baseline": None}), flush=True)
    if os.environ.get("KVAZZUP_BENCH_TILESPLIT_TIMES"):
        print("rank %d seconds per phase over %d pictures: %s" % (rank, args.steps, {k: round(v, 4) for k, v in getattr(be, "times", {}).items()}), file=sys.stderr, flush=True)
    be.close()
    if world > 1:
        dist.destroy_process_group()


def tilesplit_decode(args, aus, h, tile_rows, rank, world, torch, dist, dev, dev_index, backend):
    """the split DECODER on the stream just coded: every rank gets every access unit (untimed: rank 0 broadcasts them, as the network
    would deliver them to every rank), parses and reconstructs its tile rows, exchanges the two boundary blocks per picture
    (tilesplit.BandDecoder).  Timed: all pictures after the first `warmup`; pictures stay in HBM."""
    import numpy as np
    from kvazzup_amd.codec import split_nals
    from kvazzup_amd.tilesplit import BandDecoder
    if world > 1:
        stage = dev if backend == "nccl" else "cpu"
        meta = torch.zeros(1, dtype=torch.int64, device=stage)
        if rank == 0:
            meta[0] = len(aus)
        dist.broadcast(meta, 0)
        sizes = torch.zeros(int(meta[0]), dtype=torch.int64, device=stage)
        if rank == 0:
            sizes.copy_(torch.tensor([len(a) for a in aus], dtype=torch.int64))
        dist.broadcast(sizes, 0)
        blob = torch.zeros(int(sizes.sum()), dtype=torch.uint8, device=stage)
        if rank == 0:
            blob.copy_(torch.from_numpy(np.frombuffer(b"".join(aus), dtype=np.uint8).copy()))
        dist.broadcast(blob, 0)
        raw, offs = blob.cpu().numpy().tobytes(), np.concatenate([[0], np.cumsum(sizes.cpu().numpy())])
        aus = [raw[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    nals = [list(split_nals(a)) for a in aus]
    bd = BandDecoder((h + 63) // 64, tile_rows, rank, world, device=dev_index, dist=dist if world > 1 else None, download=False)
    warm = max(1, min(args.warmup, len(aus) - 1))
    done = 0
    t0 = None
    for t, units in enumerate(nals):
        if t == warm:
            torch.cuda.synchronize()
            barrier_max(dist, backend, dev, torch)
            t0 = time.perf_counter()
        got = None
        for nal in units:
            got = bd.feed(nal, t)
        if world > 1:
            assert got is True, "the band of picture %d did not complete" % t
            got = bd.finish_exchange()
        assert got is not None and got["height"] == h
        done += t >= warm
    torch.cuda.synchronize()
    barrier_max(dist, backend, dev, torch)
    elapsed = barrier_max(dist, backend, dev, torch, time.perf_counter() - t0)
    bd.close()
    return {"metric": "hevc_decode_fps_one_stream_tile_row_split", "value": round(done / elapsed, 3), "unit": "frames/s", "pictures": done,
            "ms_per_picture": round(elapsed / done * 1e3, 4), "exchange": "2 blocks of 8 x W bytes per internal boundary and picture, around the deblocking"}


def run_stream(args, wl, steps, warmup, ranks, rank, world, quality, host_io=False, extra_custom=(), extra_settings=None):
    """K = steps intra periods of one stream through KvazaarFilter' -> WireAdapter -> OpenHEVCFilter' on this rank's GPU, timed
    args.repeats times (median of 3 runs); every repetition starts and ends on an empty, flushed pipeline.
    host_io: the reference's own boundary -- pictures enter as HOST I420 through kvz_api->encoder_encode(kvz_picture*) (the filter's
    memcpy into the kvz_picture included, kvazaarfilter.cpp:410-438) and leave through libOpenHevcGetOutput + the filter's row copy into
    host memory (openhevcfilter.cpp:192-239): PCIe both ways inside the timed region.
    Returns a dict of measurements.  sync(value=None) = barrier / max over ranks."""
    from kvazzup_amd.pipeline import Pipeline
    import ctypes as C
    dev_index, sync = ranks.dev_index, ranks.sync
    w, h = wl["w"], wl["h"]
    # pictures parsed concurrently (video/OPENHEVC_threads): the ring has to cover the parse of an intra picture -- ~10 ms on one core at
    # 4K, where twelve pictures pass in 6 ms (measured: 2090 frames/s with 12, 2390 with 24; 1080p, 3.5 ms per intra picture: 5900-6170 / 6500-6700)
    D = max(1, args.decoder_frame_threads or 32)      # (round 3, end: 32 against 24: 8 430-8 580 against 8 240-8 350 frames/s at 1080p, 3 150-3 170 against 2 910-3 100 at 4K, two alternating runs each)
    budget = float(os.environ.get("KVAZZUP_BENCH_CPU_BUDGET", 0)) or cpu_budget(world)
    if budget < 13.0:                                # not enough host CPU for the full thread complement: shrink the pools
        D = max(1, min(D, int(budget * 0.45 + 0.5)))
        os.environ.setdefault("KVAZZUP_AMD_ENTROPY_THREADS", str(max(2, min(16, int(budget * 0.4)))))
        os.environ.setdefault("KVAZZUP_AMD_PARSE_THREADS", str(max(1, min(16, int(budget * 0.4)))))       # (row-parallel parser of the synchronous decoder)
    seed = stream_seed(wl["cfg_index"], rank)
    # synthetic clip generated directly in HBM (inputs resident before the timed region)
    nclip = CLIP_FRAMES
    dclip = DeviceClip(ranks.lib, dev_index, seed, w, h, nclip)
    clip = dclip.ptr
    host_clip = [dclip.host(t) for t in range(nclip)] if host_io else None       # pageable host memory, as a camera filter's frames are

    def device_sync():
        if not ranks.lib.kvzx_harness_sync(dev_index):
            raise RuntimeError("device synchronisation failed")

    def make(keep, download):
        st = {"video/QP": 32, "video/Intra": PERIOD, "video/VPS": 1, "uvgx/gpu": dev_index, "uvgx/decoderDownload": int(download),
              "video/OWF": args.owf, "video/OPENHEVC_threads": D, "video/OH_parallelization": "Frame" if D > 1 else "Slice"}
        st.update(extra_settings or {})
        if os.environ.get("KVAZZUP_BENCH_COPY_THREADS"):
            st["uvgx/copyThreads"] = os.environ["KVAZZUP_BENCH_COPY_THREADS"]       # (measurement aid: helpers of the filters' picture copies, default 4)
        return Pipeline(w, h, settings=st,
                        custom=(("me-range", args.me_range), ("gpu", dev_index)) + ((("sao", "full"),) if args.sao else ()) + ((("me-early-termination", "off"),) if args.full_search else ()) + ((("intra-satd", "0"),) if args.intra_sad else ()) + ((("gpu-entropy", "1"),) if args.gpu_entropy else ()) + ((("subme", str(args.subme)),) if args.subme else ()) + tuple(tuple(kv.split("=", 1)) for kv in args.custom) + tuple(extra_custom),
                        loopback=True, keep_outputs=keep)

    # source -> KvazaarFilter -> WireAdapter -> OpenHEVCFilter -> sink, one thread per filter (csrc/filters.hip)
    pl = make(False, host_io)
    lib = pl.lib
    enc_h, dec_h = pl.encoder_handle(), pl.decoder_handle()
    cw, ch = C.c_int(), C.c_int()
    lib.kvzx_encoder_coded_size(enc_h, C.byref(cw), C.byref(ch))
    cw, ch = cw.value, ch.value

    def run(npic):
        """push `npic` more pictures, flush the pipeline, wait until every one of them has been decoded.  The feeder keeps the
        encoder filter's input buffer short of its overflow threshold (a uvgComm filter drops inputs at 10 buffered, filter.cpp:151-222)."""
        last = pl.pushed + npic
        while pl.pushed < last:
            ok = pl.push_host_paced(host_clip[pl.pushed % nclip], 6, 120000) if host_io else pl.push_device_paced(clip[pl.pushed % nclip], 6, 120000)
            if not ok:
                raise RuntimeError("pipeline stalled")
        pl.flush()
        if not pl.wait(last, 120000):
            raise RuntimeError("pipeline did not deliver %d pictures: %r" % (last, pl.stats()))

    def times(reset):
        ms = (C.c_double * 16)()
        n = (C.c_uint64 * 16)()
        out = {}
        k = lib.kvzx_encoder_kernel_times(enc_h, ms, n, int(reset))
        for i in range(k):
            out[lib.kvzx_encoder_kernel_name(i).decode()] = (ms[i], n[i])
        k = lib.kvzx_decoder_kernel_times(dec_h, ms, n, int(reset))
        for i in range(k):
            name = lib.kvzx_decoder_kernel_name(i).decode()
            a = out.get(name, (0.0, 0))
            out[name] = (a[0] + ms[i], a[1] + n[i])
        return out

    run(max(1, warmup) * PERIOD if warmup > 0 else 8)         # warm-up: whole periods, so that the first timed picture is an IDR (8 pictures when --warmup 0: the pipeline must at least be built)
    if pl.pushed % PERIOD:
        run(PERIOD - pl.pushed % PERIOD)
    # HIP events around every kernel of every 8th picture of the timed region (IDR pictures fall on multiples of 8)
    prof = 0 if os.environ.get("KVAZZUP_BENCH_NOPROF") else args.profile_every
    lib.kvzx_encoder_set_profiling(enc_h, prof)
    lib.kvzx_decoder_set_profiling(dec_h, prof)
    busy0 = pl.busy_ms()
    st0 = pl.stats()
    times(True)
    _thr0 = _thread_cpu() if os.environ.get("KVAZZUP_BENCH_THREADS") else None
    _sampler = None
    if os.environ.get("CPU_SAMPLER_REGION"):                  # tools/cpu_sampler.c preloaded: sample the timed regions only
        _sampler = C.CDLL(None)
    reps = []
    for rep in range(max(1, args.repeats)):
        device_sync()                                          # the pipeline is empty: everything pushed so far has been decoded
        sync()
        cpu0 = time.process_time()
        thr0 = _throttled_us()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        if _sampler is not None:
            _sampler.cpu_sampler_begin()
        t0 = time.perf_counter()
        run(steps * PERIOD)
        device_sync()
        sync()
        el = time.perf_counter() - t0
        if _sampler is not None:
            _sampler.cpu_sampler_end()
        # throttled: summed over the job's threads: > 0 means the CPU quota, not the GPU, set the pace for a while; host cores: CPU
        # seconds of all threads of this rank per second of the timed region
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        reps.append({"elapsed": sync(el), "throttled_ms": (_throttled_us() - thr0) / 1e3, "host_cores": (time.process_time() - cpu0) / el,
                     "minflt": (ru1.ru_minflt - ru0.ru_minflt) / (steps * PERIOD), "sys_share": (ru1.ru_stime - ru0.ru_stime) / max(1e-9, (ru1.ru_stime - ru0.ru_stime) + (ru1.ru_utime - ru0.ru_utime))})
    if _thr0 is not None:                                       # KVAZZUP_BENCH_THREADS=1: CPU time per thread over the timed regions (stderr)
        _thr1 = _thread_cpu()
        tot = sum(r["elapsed"] for r in reps)
        rows = sorted(((_thr1[t][1] - _thr0.get(t, ("", 0.0))[1], t, _thr1[t][0]) for t in _thr1), reverse=True)
        for dt, t, name in rows[:40]:
            if dt > 0:
                print("thread %7d %-16s %.3f s (%.2f cores)" % (t, name, dt, dt / tot), file=sys.stderr)
    med = sorted(reps, key=lambda r: r["elapsed"])[len(reps) // 2]
    elapsed, throttled_ms, host_cores = med["elapsed"], med["throttled_ms"], med["host_cores"]
    kt = times(False)
    npic = steps * PERIOD
    nall = npic * len(reps)
    busy = [round((b - a) / nall, 4) for a, b in zip(busy0, pl.busy_ms())]
    st = pl.stats()
    nbytes = st["encoded_bytes"] - st0["encoded_bytes"]
    if st["decoded_pictures"] != pl.pushed or st["dropped"] or st["encoded_pictures"] - st0["encoded_pictures"] != nall:
        raise RuntimeError("pipeline lost pictures: %r" % (st,))
    pl.close()

    out = {"elapsed": elapsed, "pictures": npic, "cw": cw, "ch": ch, "kt": kt, "busy": busy, "bytes_per_picture": nbytes / nall, "D": D,
           "host_cores": host_cores, "budget": budget, "throttled_ms": throttled_ms, "psnr_y": None, "minflt": med["minflt"], "sys_share": med["sys_share"],
           "runs_fps": [round(world * npic / r["elapsed"], 1) for r in reps]}
    if quality:
        # Quality of what was just timed (untimed pass): one intra period through a second pipeline with the decoded pictures
        # downloaded; luma PSNR of the decoder's output against the source, mean over the period's 64 pictures.
        import numpy as np
        q = make(True, True)
        for t in range(PERIOD):
            if not q.push_device_paced(clip[t], 6, 120000):
                raise RuntimeError("quality pass stalled")
        q.flush()
        if not q.wait(PERIOD, 120000):
            raise RuntimeError("quality pass did not deliver")
        ps = []
        for t in range(PERIOD):
            d = q.pop_decoded()
            src = dclip.host(t)[:w * h].astype(np.int32)
            mse = float(((src - d["i420"][:w * h].astype(np.int32)) ** 2).mean(dtype=np.float64))
            ps.append(99.0 if mse == 0 else 10.0 * float(np.log10(255.0 * 255.0 / mse)))
        q.close()
        out["psnr_y"] = round(sum(ps) / len(ps), 3)
    dclip.close()
    return out


def multi_stream(args, wl, K, steps, ranks):
    """K independent streams (a K-party call) on ONE GPU at the same time, each through its own KvazaarFilter' -> WireAdapter -> OpenHEVCFilter' chain
    (own encoder, decoder, HIP streams, host threads): what the GPU sustains when it is not waiting for one stream's chain of dependent kernels.
    The host side is divided between the streams (decoder frame threads, coder threads).  Returns the aggregate frames/s and each stream's."""
    import threading
    from kvazzup_amd.pipeline import Pipeline
    w, h = wl["w"], wl["h"]
    budget = float(os.environ.get("KVAZZUP_BENCH_CPU_BUDGET", 0)) or cpu_budget(1)
    # the parse ring has to cover an intra picture's parse (3.5 ms at 1080p): 32 pictures at one stream's full rate, 32 / K at a K-th of it
    D = int(os.environ.get("KVAZZUP_BENCH_MULTI_D", 0)) or max(2, min(32, 32 // K))
    threads = max(2, min(16, int(budget * 0.5 / K)))
    clips = [DeviceClip(ranks.lib, ranks.dev_index, stream_seed(wl["cfg_index"], k), w, h, PERIOD) for k in range(K)]
    pls = [Pipeline(w, h, settings={"video/QP": 32, "video/Intra": PERIOD, "video/VPS": 1, "uvgx/gpu": ranks.dev_index, "uvgx/decoderDownload": 0, "video/kvzThreads": threads,
                                    "video/OWF": args.owf, "video/OPENHEVC_threads": D, "video/OH_parallelization": "Frame"},
                    custom=(("me-range", args.me_range), ("gpu", ranks.dev_index), ("input-hold", "1")), loopback=True, keep_outputs=False) for _ in range(K)]
    gate = threading.Barrier(K + 1)
    elapsed = [0.0] * K

    def body(k):
        pl, clip = pls[k], clips[k].ptr

        def run(npic):
            last = pl.pushed + npic
            while pl.pushed < last:
                if not pl.push_device_paced(clip[pl.pushed % PERIOD], 6, 120000):
                    raise RuntimeError("pipeline stalled")
            pl.flush()
            if not pl.wait(last, 120000):
                raise RuntimeError("pipeline did not deliver")
        run(PERIOD)                                     # warm-up: one period
        gate.wait()
        t0 = time.perf_counter()
        run(steps * PERIOD)
        elapsed[k] = time.perf_counter() - t0
        gate.wait()

    import ctypes as C
    lib = ranks.lib
    lib.kvzx_batch_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    lib.kvzx_batch_kernel_name.restype = C.c_char_p
    ths = [threading.Thread(target=body, args=(k,)) for k in range(K)]
    for t in ths:
        t.start()
    gate.wait()
    if not os.environ.get("KVAZZUP_BENCH_NOPROF"):
        for pl in pls:
            lib.kvzx_decoder_set_profiling(pl.decoder_handle(), args.profile_every)
    lib.kvzx_batch_stats(ranks.dev_index, None, None, None, None, None, None, 1)
    cpu0, t0 = time.process_time(), time.perf_counter()
    gate.wait()
    wall = time.perf_counter() - t0
    cores = (time.process_time() - cpu0) / wall
    nb, npics = C.c_uint64(), C.c_uint64()
    sizes, bms, bln, bfr = (C.c_uint64 * 9)(), (C.c_double * 4)(), (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
    nk = lib.kvzx_batch_stats(ranks.dev_index, C.byref(nb), C.byref(npics), sizes, bms, bln, bfr, 0)
    # the decoders' submission layer (csrc/batch.h): pictures per launch, and the batched kernels against the HBM roofline -- algorithmic bytes of
    # the pictures in a launch over the launch's duration, beside the single-picture kernels' fractions in `roofline.frac_by_kernel`
    cw, ch = (w + 63) // 64 * 64, (h + 63) // 64 * 64
    batched = {}
    for i in range(nk):
        if bln[i]:
            name = lib.kvzx_batch_kernel_name(i).decode()
            us, per = bms[i] / bln[i] * 1e3, bfr[i] / bln[i]
            single = {"k_dec_inter_n": "k_dec_inter", "k_dec_intra_n": "k_dec_intra", "k_dec_deblock_n": "k_dec_deblock", "k_dec_sao_n": "k_dec_sao"}[name]
            batched[name] = {"avg_launch_us": round(us, 2), "pictures_per_launch": round(per, 2), "us_per_picture": round(us / per, 2),
                             "hbm_frac": round(algorithmic_bytes(single, cw, ch, args.me_range) * per / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)}
    for t in ths:
        t.join()
    for pl in pls:
        st = pl.stats()
        if st["decoded_pictures"] != pl.pushed or st["dropped"]:
            raise RuntimeError("a stream lost pictures: %r" % (st,))
        pl.close()
    for c in clips:
        c.close()
    npic = steps * PERIOD
    return {"streams": K, "value": round(K * npic / max(elapsed), 1), "unit": "frames/s (all streams together)", "per_stream": [round(npic / e, 1) for e in elapsed],
            "steps_per_stream": steps, "decoder_frame_threads_per_stream": D, "coder_threads_per_stream": threads, "host_cpu_cores_busy": round(cores, 2),
            "decoder_batches": {"launches": nb.value, "pictures": npics.value, "pictures_per_batch": round(npics.value / max(1, nb.value), 3),
                                "batches_by_size": {str(n): sizes[n] for n in range(1, 9) if sizes[n]}, "kernels": batched}}


def roofline_of(m, steps, me_range, workload_key):
    """dominant kernel = largest share of the timed region: average launch time x launches in the region (the events sample
    every n-th picture, so the launch counts come from the picture types, not from the samples)"""
    kt, cw, ch = m["kt"], m["cw"], m["ch"]
    if not any(v[1] for v in kt.values()):
        return None, {}, {}            # KVAZZUP_BENCH_NOPROF=1: throughput-only run
    n_idr, npic = steps, steps * PERIOD

    def launches(k):
        if k in ("k_intra_analyse", "k_intra_recon", "k_dec_intra", "k_intra_recon<dec>"):
            return n_idr
        if k in ("k_me", "k_inter_recon", "k_inter_signal", "k_dec_inter", "k_inter_recon<dec>"):
            return npic - n_idr
        return npic
    kern = [k for k in kt if kt[k][1] > 0 and k.startswith("k_")]
    dom = max(kern, key=lambda k: kt[k][0] / kt[k][1] * launches(k))
    avg_s = kt[dom][0] / kt[dom][1] / 1e3
    ab = algorithmic_bytes(dom, cw, ch, me_range)
    achieved = ab / avg_s / 1e9
    # HBM traffic per launch: from the committed PMC passes of this same command (rocprofv3 --pmc cannot run inside the bench);
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.  null when no pass exists.
    traffic, traffic_src = None, None
    mfma = None
    for rnd in ("r03", "r02", "r01"):
        try:
            name = "profiles/%s_pmc_traffic_%s.json" % (rnd, workload_key)
            pmc = json.load(open(os.path.join(ROOT, name)))
            traffic = pmc["kernels"][dom]["traffic_bytes"]
            traffic_src = name
            # matrix-core utilisation of the kernels that use them (same counter passes: SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x SIMDs))
            mfma = {k: v["mfma_util"] for k, v in pmc["kernels"].items() if v.get("mfma_util")} or None
            break
        except Exception:
            pass
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "peak_source": HBM_PEAK_SOURCE, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": ab, "avg_launch_us": round(avg_s * 1e6, 2)}
    kernels_us = {k: round(v[0] / v[1] * 1e3, 2) for k, v in kt.items() if v[1]}
    share = {k: round(v[0] / v[1] * launches(k) / (m["elapsed"] * 1e3), 4) for k, v in kt.items() if v[1]}
    # every kernel against the HBM roofline (algorithmic bytes of one launch / its average duration)
    per_kernel = {k: round(algorithmic_bytes(k, cw, ch, me_range) / (kt[k][0] / kt[k][1] / 1e3) / 1e9 / HBM_PEAK_GBS, 5) for k in kern}
    roof["frac_by_kernel"] = per_kernel
    roof["mfma_util_by_kernel"] = mfma                 # north_star: "MFMA utilisation against gfx950 peak" -- the transforms and Hadamard sums are small products between LDS phases
    return roof, kernels_us, share


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24, help="timed steps; a step = one intra period = 64 pictures")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up steps (intra periods)")
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--me-range", type=int, default=16)
    ap.add_argument("--repeats", type=int, default=3, help="the K-step timed region is run this many times; `value` is the median run (BASELINE.md: median of 3)")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the `host_boundary` legs (host I420 in through kvz_api->encoder_encode, decoded I420 out into host memory)")
    ap.add_argument("--host-io", action="store_true", help="profiling aid: the MAIN run goes through the host boundary (then `value` is the host-boundary rate and no separate leg is run)")
    ap.add_argument("--streams-per-gpu", default="2,4", help="comma-separated K, e.g. 2,4: K independent streams at once on the one GPU, each with its own filter chain in this process (aggregate frames/s); off by default: "
                         "with HIP's four hardware queues per priority level the streams of several pipelines share queues and serialise (DESIGN.md section 6; GPU_MAX_HW_QUEUES=8 lifts two streams from 3100 to 5700 frames/s)")
    ap.add_argument("--custom", action="append", default=[], metavar="KEY=VALUE", help="extra kvazaar option for the encoder of every leg (uvgComm's custom-parameter list, kvazaarfilter.cpp:355-368), e.g. --custom intra-in-p=1")
    ap.add_argument("--no-preset-line", action="store_true", help="skip the `default_mode` line: uvgComm's own default encoder settings for this size (defaultsettings.cpp:287-316: preset veryfast, 1 Mbit/s) instead of the benchmark's fixed-QP ultrafast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-decode", action="store_true", help="8k-tilesplit: skip the split decoder's leg (reported as `secondary`)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 4K line (configs[2], the north-star target) that a 1080p run appends as `secondary`")
    ap.add_argument("--secondary-steps", type=int, default=8)
    ap.add_argument("--cpu-frames", type=int, default=12, help="pictures per core in the cpu_baseline sample")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--decoder-frame-threads", type=int, default=0,
                    help="OpenHEVC 'Frame' parallelisation (uvgComm setting video/OH_parallelization): pictures parsed concurrently; 1 = off; 0 (default) = 32")
    ap.add_argument("--profile-every", type=int, default=8, help="kernel timing with HIP events on every n-th picture")
    ap.add_argument("--intra-sad", action="store_true", help="intra-satd=0: the intra mode search compares SADs instead of 8x8 Hadamard sums (for the quality / rate comparison in DESIGN.md)")
    ap.add_argument("--full-search", action="store_true", help="me-early-termination=off: every 32x32 block is searched exhaustively (the k_me issue-rate roofline is reported for this case)")
    ap.add_argument("--subme", type=int, default=0, help="kvazaar subme 0..4: fractional-sample motion refinement (0 at the ultrafast preset the headline workload uses; 2 / 4 at the presets above)")
    ap.add_argument("--gpu-entropy", action="store_true", help="gpu-entropy=1: the arithmetic coder on the GPU (k_cabac_rows) instead of the host thread pool (A/B measurement, DESIGN.md section 5)")
    ap.add_argument("--sao", action="store_true", help="kvazaar sao=full (off at the ultrafast preset the headline workload uses)")
    ap.add_argument("--owf", type=int, default=6,
                    help="uvgComm setting video/OWF (kvazaar owf): 1 = host arithmetic coding of picture t overlaps the kernels of t + 1; "
                         "2 = it runs on background threads and the output lags two pictures; 3 .. 8 = that many pictures in flight "
                         "(the settings UI offers 0 .. core count, videosettings.cpp:488-493)")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(*[int(v) for v in args.cpu_worker.split(",")])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.workload == "8k-tilesplit":
        return tilesplit_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ranks = StreamRanks(world, local_rank)
    backend = ranks.backend
    # HBM peak: from the device (hipDeviceProp: bus width x memory clock x 4, harness_kernels.hip); the guide's 8 TB/s only when the runtime does not report them
    global HBM_PEAK_GBS, HBM_PEAK_SOURCE
    import ctypes as C
    ranks.lib.kvzx_harness_hbm_peak_gbs.restype = C.c_double
    ranks.lib.kvzx_harness_hbm_peak_gbs.argtypes = [C.c_int]
    name, cus, clk, mclk, bus = C.create_string_buffer(128), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ranks.lib.kvzx_harness_device_info(ranks.dev_index, name, 128, C.byref(cus), C.byref(clk), C.byref(mclk), C.byref(bus))
    pk = ranks.lib.kvzx_harness_hbm_peak_gbs(ranks.dev_index)
    if pk > 0:
        HBM_PEAK_GBS = round(pk, 1)
        HBM_PEAK_SOURCE = "hipDeviceProp of %s: memoryBusWidth %d bit x memoryClockRate %d MHz x 4 transfers per clock (HBM3E) / 8" % (name.value.decode(), bus.value, mclk.value)
    device_info = {"name": name.value.decode(), "compute_units": cus.value, "clock_mhz": clk.value, "memory_clock_mhz": mclk.value, "memory_bus_bits": bus.value}

    wl = WORKLOADS[args.workload]
    w, h = wl["w"], wl["h"]
    resident = (("input-hold", "1"),)       # the clip's device pictures stay untouched: encode_device returns without waiting for its input stage
    if args.host_io:
        args.no_host_boundary = True
        m = run_stream(args, wl, args.steps, args.warmup, ranks, rank, world, quality=False, host_io=True, extra_custom=HOST_CUSTOM)
    else:
        m = run_stream(args, wl, args.steps, args.warmup, ranks, rank, world, quality=(rank == 0), extra_custom=resident)
    def host_leg(wl_, steps_, warm_, resident):
        """the same steps through the reference's own boundary (run_stream host_io); a dict for the JSON line"""
        try:
            hb = run_stream(args, wl_, steps_, warm_, ranks, rank, world, quality=False, host_io=True, extra_custom=HOST_CUSTOM)
        except Exception as e:
            return {"error": str(e)}
        fps_h = world * hb["pictures"] / hb["elapsed"]
        pic = wl_["w"] * wl_["h"] * 3 // 2
        return {"value": round(fps_h, 3), "unit": "frames/s", "runs": hb["runs_fps"], "of_resident": round(fps_h / resident, 4),
                "h2d_GBps": round(fps_h * pic / 1e9 / world, 2), "d2h_GBps": round(fps_h * pic / 1e9 / world, 2),
                "host_cpu_cores_busy": round(hb["host_cores"], 2), "minor_page_faults_per_picture": round(hb["minflt"], 1), "cpu_time_in_kernel": round(hb["sys_share"], 3),
                "filter_busy_ms_per_picture": {"KvazaarFilter": hb["busy"][0], "WireAdapter": hb["busy"][1], "OpenHEVCFilter": hb["busy"][2]},
                "boundary": "host I420 -> KvazaarFilter' (memcpy into a page-locked kvz_picture, kvz_api->encoder_encode; custom parameters recon-output=0: uvgComm frees the "
                            "reconstruction unread, kvazaarfilter.cpp:476; null-input=poll: the loop at :440-448 collects finished pictures without emptying the pipeline) -> access units -> OpenHEVCFilter' (libOpenHevcDecode / GetOutput, row copy into host "
                            "memory, openhevcfilter.cpp:192-239); uploads and downloads on their own HIP streams beside the kernels"}

    hostb = None
    if not args.no_host_boundary:
        hostb = host_leg(wl, args.steps, args.warmup, world * args.steps * PERIOD / m["elapsed"])
    multi = None
    if world == 1 and args.streams_per_gpu and not args.host_io:
        multi = []
        for K in [int(v) for v in args.streams_per_gpu.split(",") if v.strip()]:
            try:
                multi.append(multi_stream(args, wl, K, max(2, args.steps // 2), ranks))
            except Exception as e:
                multi.append({"streams": K, "error": str(e)})
    sec = None
    sec_host = None
    if world == 1 and args.workload == "1080p" and not args.no_secondary:
        # (a second pipeline in this process inherits the first one's HIP streams -- csrc/stream_pool.h -- and with them its hardware-queue
        # layout; before that pool the 4K leg ran 15-20 % slower here than in a process of its own)
        try:
            sec = run_stream(args, WORKLOADS["4k"], max(1, min(args.steps, args.secondary_steps)), min(2, max(1, args.warmup)), ranks, rank, world, quality=True, extra_custom=resident)
            if not args.no_host_boundary:
                ssteps_ = max(1, min(args.steps, args.secondary_steps))
                sec_host = host_leg(WORKLOADS["4k"], ssteps_, 1, sec["pictures"] / sec["elapsed"])
        except Exception as e:       # the headline line must not be lost to the secondary one
            sec = {"error": str(e)}

    preset_line = None
    if world == 1 and args.workload == "1080p" and not args.no_secondary and not args.no_preset_line:
        # uvgComm's default mode for a stream of this complexity class (defaultsettings.cpp:300-316): preset veryfast (here: sao full, subme 2, intra-in-p),
        # rate control at 1 Mbit/s (kvazaarfilter.cpp:223-228 -> rc-algorithm lambda: "uvgx rate control v2")
        try:
            psteps = max(1, min(args.steps, args.secondary_steps))
            pm = run_stream(args, wl, psteps, min(2, max(1, args.warmup)), ranks, rank, world, quality=True, extra_custom=resident,
                            extra_settings={"video/Preset": "veryfast", "video/bitrate": 1000000})
            preset_line = {"settings": {"video/Preset": "veryfast", "video/bitrate": 1000000}, "value": round(pm["pictures"] / pm["elapsed"], 3), "unit": "frames/s",
                           "steps": psteps, "runs_fps": pm["runs_fps"], "bits_per_picture": round(8 * pm["bytes_per_picture"], 1),
                           "kbit_per_s_at_30fps": round(8 * pm["bytes_per_picture"] * 30 / 1e3, 1), "psnr_y": pm["psnr_y"],
                           "host_cpu_cores_busy": round(pm["host_cores"], 2), "kernels_us": roofline_of(pm, psteps, args.me_range, args.workload)[1]}
        except Exception as e:
            preset_line = {"error": str(e)}

    if rank == 0:
        npic = args.steps * PERIOD
        fps = world * npic / m["elapsed"]
        roof, kernels_us, share = roofline_of(m, args.steps, args.me_range, args.workload)
        out = {
            "metric": "hevc_encode_decode_fps", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"], "width": w, "height": h, "coded_width": m["cw"], "coded_height": m["ch"],
                       "pictures_per_step": PERIOD, "step": "one intra period: 1 IDR + 63 P pictures, encode + decode", "pictures_per_gpu": npic,
                       "ms_per_picture": round(m["elapsed"] / npic * 1e3, 5),
                       "timed_region": "empty flushed pipeline -> last timed picture decoded and flushed out",
                       "intra_period": PERIOD, "qp": 32, "me_range": args.me_range, "streams": world, "collective_backend": backend,
                       "bits_per_picture": round(8 * m["bytes_per_picture"], 1), "psnr_y": m["psnr_y"],
                       "decoder_frame_threads": m["D"], "owf": args.owf, "gpu_entropy": bool(args.gpu_entropy), "subme": args.subme, "sao": bool(args.sao), "me_early_termination": not args.full_search, "intra_satd": not args.intra_sad,
                       "host_cpu_cores_busy": round(m["host_cores"], 2), "minor_page_faults_per_picture": round(m["minflt"], 1), "cpu_time_in_kernel": round(m["sys_share"], 3), "host_cpu_budget_cores": round(m["budget"], 1), "host_cpu_throttled_ms": round(m["throttled_ms"], 1),
                       "input": "host I420 through kvz_api->encoder_encode" if args.host_io else "I420 resident in HBM",
                       "output": "Annex-B AU on host + decoded I420 in " + ("host memory" if args.host_io else "HBM"),
                       "repeats": args.repeats, "runs_fps": m["runs_fps"], "value_is": "median run of `repeats` (BASELINE.md timing rule)"},
            "host_boundary": hostb,
            "streams_per_gpu": multi,
            "default_mode": preset_line,
            "device": device_info,
            "roofline": roof,
            "kernels_us": kernels_us,
            "filter_busy_ms_per_picture": {"KvazaarFilter": m["busy"][0], "WireAdapter": m["busy"][1], "OpenHEVCFilter": m["busy"][2]},
            "kernel_share_of_step": share,
        }
        if sec is not None:
            if "error" in sec:
                out["secondary"] = {"workload": WORKLOADS["4k"]["name"], "error": sec["error"]}
            else:
                ssteps = max(1, min(args.steps, args.secondary_steps))
                sroof, skern, _ = roofline_of(sec, ssteps, args.me_range, "4k")
                out["secondary"] = {"workload": WORKLOADS["4k"]["name"], "value": round(sec["pictures"] / sec["elapsed"], 3), "unit": "frames/s",
                                    "steps": ssteps, "pictures_per_step": PERIOD, "ms_per_step": round(sec["elapsed"] / ssteps * 1e3, 4),
                                    "bits_per_picture": round(8 * sec["bytes_per_picture"], 1), "psnr_y": sec["psnr_y"],
                                    "runs_fps": sec["runs_fps"], "host_boundary": sec_host,
                                    "host_cpu_cores_busy": round(sec["host_cores"], 2), "roofline": sroof, "kernels_us": skern}
        if args.full_search and "k_me" in m["kt"] and m["kt"]["k_me"][1]:
            # The motion search is integer VALU work, not streaming: its own ceiling is the issue rate of v_qsad_pk_u16_u8
            # (four 4-sample SADs per lane; measured ~24 cycles per wave instruction on gfx950, tools/qsad_bench.hip ->
            # profiles/r02_qsad_bench.txt): 1024 SIMDs x 2.4 GHz / 24 x 64 lanes x 16 sample differences.
            W = 2 * args.me_range + 1
            sads = (m["cw"] * m["ch"] // 1024) * W * W * 1024
            me_s = m["kt"]["k_me"][0] / m["kt"]["k_me"][1] / 1e3
            peak = 1024 * 2.4e9 / 24 * 64 * 16
            out["roofline_valu"] = {"kernel": "k_me", "bound": "valu (v_qsad_pk_u16_u8 issue rate)", "achieved": round(sads / me_s / 1e12, 2),
                                    "peak": round(peak / 1e12, 2), "unit": "T sample-differences/s", "frac": round(sads / me_s / peak, 4)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(w, h, args.cpu_frames, args.me_range, max(1, int(cpu_budget(1))))
            except Exception as e:       # the checker library is test infrastructure; report, do not fail the bench
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
