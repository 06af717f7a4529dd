#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: HEVC encode+decode fps on synthetic YUV420 (uvgx-synth-v1).

A "step" is one picture through the hot path: kvz_api-side encode (HIP kernels; input picture
already resident in HBM) and libOpenHevc-side decode of the access unit it produced (host CABAC
parse + HIP reconstruction, output left in HBM).  The two codecs sit in the C++ mirrors of
uvgComm's KvazaarFilter and OpenHEVCFilter, each on its own thread as in the reference's filter
graph, so picture t+1 is encoded while picture t is decoded; the timed region starts with the first
push and ends when the last decoded picture has left the decoder.  N > 1 runs one independent stream
per GPU (BASELINE configs[3]: multi-party call, no collective on the data path), weak scaling.

Prints ONE JSON line on rank 0 (see the contract in the task description / DESIGN.md section 6).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: 1080p, preset=ultrafast, intra period 64, encode + decode on one GPU
    "1080p": dict(w=1920, h=1080, name="1080p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=2),
    # configs[2]: 4K encode (decode is run too; reported in the same fps)
    "4k": dict(w=3840, h=2160, name="2160p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=3),
    "720p": dict(w=1280, h=720, name="720p-yuv420-ultrafast-p64-qp32-encode+decode", cfg_index=2),
    # configs[4]: ONE 8K stream, its 8 tile rows split over the ranks (strong scaling; see tilesplit_main)
    "8k-tilesplit": dict(w=7680, h=4320, name="4320p-yuv420-ultrafast-p64-qp32-encode-tile-row-split", cfg_index=5),
}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes(kernel, cw, ch, me_range):
    """Compulsory bytes of ONE launch, P = coded luma samples."""
    P = cw * ch
    if kernel == "k_me":                          # current block once + its search window once, per 32x32 block
        return (P // 1024) * (1024 + (32 + 2 * me_range) ** 2)
    if kernel in ("k_inter_recon", "k_inter_recon<dec>"):
        return int(4.5 * P) if kernel == "k_inter_recon" else int(3.0 * P)
    if kernel in ("k_intra_recon", "k_intra_recon<dec>"):
        return int(3.0 * P) if kernel == "k_intra_recon" else int(1.5 * P)
    if kernel == "k_intra_analyse":
        return P
    if kernel == "k_deblock":
        return int(3.0 * P)
    if kernel == "k_tokenize":                    # every level of the picture once (int16) + the per-8x8 CU records; tokens out not counted
        return int(3.0 * P) + (P // 64) * 11
    if kernel == "k_tok_compact":                 # the piece table of every CTU ([16 units][17 pieces] {offset, length}); tokens not counted
        return (P // 4096) * 16 * 17 * 8
    if kernel in ("k_sao", "k_sao<dec>"):         # deblocked picture in, filtered picture out (+ the source picture for the statistics)
        return int(4.5 * P) if kernel == "k_sao" else int(3.0 * P)
    if kernel == "k_pad_input":
        return int(3.0 * P)
    if kernel == "k_inter_signal":
        return (P // 64) * 16
    if kernel == "k_scatter_levels":
        return int(3.0 * P)
    return P


def cpu_budget(world):
    """CPU cores this rank may use: the container's CFS quota (cgroup v2 cpu.max) or the visible cores, shared by the ranks of
    the node.  One stream needs ~13 cores at full rate (8 CABAC parse workers, 16 arithmetic-coder workers, the filter and
    synchronisation threads); with less, the pools are sized down instead of letting the kernel throttle the whole job."""
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


def cpu_baseline(w, h, frames, me_range):
    """The CPU checker (oracle/, a scalar C port of the same algorithm) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    oe = orc.OracleEncoder(w, h, qp=32, period=64, me_range=me_range)
    od = orc.OracleDecoder()
    clip = [orc.synth_frame(0, 0x5EED0002, w, h, t) for t in range(frames)]
    t0 = time.time()
    n = 0
    for t, fr in enumerate(clip):
        au = oe.encode(fr)
        n += len(od.decode_au(au, t))
    dt = time.time() - t0
    oe.close()
    od.close()
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d pictures %dx%d (1 intra + %d inter, search range %d), encode+decode by oracle/ (scalar C, one thread)" % (frames, w, h, frames - 1, me_range)}


def tilesplit_main(args):
    """BASELINE configs[4]: a single 8K picture stream, 8 full-width tile rows, split over the ranks (whole tile rows per
    rank); the only exchange on the data path is the deblock halo (kvazzup_amd/tilesplit.py).  Encode only; strong scaling."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from kvazzup_amd import synth
    from kvazzup_amd.tilesplit import BandEncoder
    wl = WORKLOADS["8k-tilesplit"]
    w, h, tile_rows = wl["w"], wl["h"], 8
    total = args.warmup + args.steps
    clip = [synth.frame_torch(synth.MOVING, 0x5EED0005, w, h, t, dev) for t in range(total)]     # every rank holds the stream (a band only reads its rows)
    torch.cuda.synchronize()
    be = BandEncoder(w, h, tile_rows, rank, world, options=(("qp", 32), ("period", 64), ("me-range", args.me_range)), device=local_rank,
                     dist=dist if world > 1 else None)
    nbytes = 0
    for t in range(args.warmup):
        be.encode(clip[t].data_ptr())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        au = be.encode(clip[t].data_ptr())
        if au is not None:
            nbytes += len(au)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": "hevc_encode_fps_one_stream_tile_row_split", "value": round(args.steps / elapsed, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"], "width": w, "height": h, "tile_rows": tile_rows, "ranks": world, "ctu_rows_rank0": be.nrows,
                       "intra_period": 64, "qp": 32, "me_range": args.me_range, "bytes_per_frame": round(nbytes / args.steps, 1),
                       "halo_bytes_per_picture_and_rank": round(be.halo_bytes_exchanged / max(1, total), 1),
                       "exchange": "2 halo blocks per internal boundary and picture (4 luma + 2x2 chroma rows + CU records), RCCL send/recv"},
            "roofline": None, "cpu_baseline": None}))
    be.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1536)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--me-range", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--decoder-frame-threads", type=int, default=12,
                    help="OpenHEVC 'Frame' parallelisation (uvgComm setting video/OH_parallelization): pictures parsed concurrently; 1 = off")
    ap.add_argument("--profile-every", type=int, default=8, help="kernel timing with HIP events on every n-th picture")
    ap.add_argument("--full-search", action="store_true", help="me-early-termination=off: every 32x32 block is searched exhaustively (the k_me issue-rate roofline is reported for this case)")
    ap.add_argument("--sao", action="store_true", help="kvazaar sao=full (off at the ultrafast preset the headline workload uses)")
    ap.add_argument("--owf", type=int, default=3,
                    help="uvgComm setting video/OWF (kvazaar owf): 1 = host arithmetic coding of picture t overlaps the kernels of t + 1; "
                         "2 = it runs on background threads and the output lags two pictures; 3 = one more picture in flight "
                         "(the settings UI offers 0 .. core count, videosettings.cpp:488-493)")
    args = ap.parse_args()
    if args.workload == "8k-tilesplit":
        return tilesplit_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from kvazzup_amd import synth
    from kvazzup_amd import _native as N
    from kvazzup_amd.pipeline import Pipeline
    import ctypes as C

    wl = WORKLOADS[args.workload]
    w, h = wl["w"], wl["h"]
    total = args.warmup + args.steps
    D = max(1, args.decoder_frame_threads)
    budget = float(os.environ.get("KVAZZUP_BENCH_CPU_BUDGET", 0)) or cpu_budget(world)
    if budget < 13.0:                                # not enough host CPU for the full thread complement: shrink the pools
        D = max(1, min(D, int(budget * 0.45 + 0.5)))
        os.environ.setdefault("KVAZZUP_AMD_ENTROPY_THREADS", str(max(2, min(16, int(budget * 0.4)))))
        os.environ.setdefault("KVAZZUP_AMD_PARSE_THREADS", str(max(1, min(16, int(budget * 0.4)))))       # (row-parallel parser of the synchronous decoder)
    extra = (D if D > 1 else 0) + min(max(args.owf, 0), 3)   # pictures pushed after the timed ones: the encoder (owf) and the frame-threaded decoder deliver with a lag
    seed = 0x5EED0000 + wl["cfg_index"] + 16 * rank
    # synthetic clip generated directly in HBM (inputs resident before the timed region)
    clip = [synth.frame_torch(synth.MOVING, seed, w, h, t, dev) for t in range(total + extra)]
    torch.cuda.synchronize()

    # source -> KvazaarFilter -> WireAdapter -> OpenHEVCFilter -> sink, one thread per filter (csrc/filters.hip)
    pl = Pipeline(w, h, settings={"video/QP": 32, "video/Intra": 64, "video/VPS": 1, "uvgx/gpu": local_rank, "uvgx/decoderDownload": 0,
                                  "video/OWF": args.owf, "video/OPENHEVC_threads": D, "video/OH_parallelization": "Frame" if D > 1 else "Slice"},
                  custom=(("me-range", args.me_range), ("gpu", local_rank)) + ((("sao", "full"),) if args.sao else ()) + ((("me-early-termination", "off"),) if args.full_search else ()), loopback=True, keep_outputs=False)
    lib = pl.lib
    enc_h, dec_h = pl.encoder_handle(), pl.decoder_handle()
    cw, ch = C.c_int(), C.c_int()
    lib.kvzx_encoder_coded_size(enc_h, C.byref(cw), C.byref(ch))
    cw, ch = cw.value, ch.value

    def run(first, count):
        """push pictures until `first + count` have been DECODED.  The feeder keeps the encoder filter's input buffer
        short of its overflow threshold (a uvgComm filter drops inputs at 10 buffered, filter.cpp:151-222)."""
        g = pl.pushed
        last = min(first + count + extra, total + extra)
        while g < last:
            if not pl.push_device_paced(clip[g].data_ptr(), 6, 120000):
                raise RuntimeError("pipeline stalled")
            g += 1
        if not pl.wait(first + count, 120000):
            raise RuntimeError("pipeline did not deliver %d pictures" % (first + count))

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

    run(0, args.warmup)
    # HIP events around every kernel of every 8th picture of the timed region (IDR pictures fall on multiples of 8)
    prof = 0 if os.environ.get("KVAZZUP_BENCH_NOPROF") else args.profile_every
    lib.kvzx_encoder_set_profiling(enc_h, prof)
    lib.kvzx_decoder_set_profiling(dec_h, prof)
    busy0 = pl.busy_ms()
    times(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    cpu0 = time.process_time()
    thr0 = _throttled_us()
    _thr0 = _thread_cpu() if os.environ.get("KVAZZUP_BENCH_THREADS") else None
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    throttled_ms = (_throttled_us() - thr0) / 1e3              # summed over the job's threads: > 0 means the CPU quota, not the GPU, set the pace for a while
    host_cores = (time.process_time() - cpu0) / elapsed        # CPU seconds of all threads of this rank per second of the timed region
    if _thr0 is not None:                                       # KVAZZUP_BENCH_THREADS=1: CPU time per thread over the timed region (stderr)
        _thr1 = _thread_cpu()
        rows = sorted(((_thr1[t][1] - _thr0.get(t, ("", 0.0))[1], t, _thr1[t][0]) for t in _thr1), reverse=True)
        for dt, t, name in rows[:40]:
            if dt > 0:
                print("thread %7d %-16s %.3f s (%.2f cores)" % (t, name, dt, dt / elapsed), file=sys.stderr)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kt = times(False)
    busy = [round((b - a) / args.steps, 4) for a, b in zip(busy0, pl.busy_ms())]
    st = pl.stats()
    nbytes = st["encoded_bytes"] * args.steps / max(1, st["encoded_pictures"])
    if st["decoded_pictures"] < total or st["dropped"]:
        raise RuntimeError("pipeline lost pictures: %r" % (st,))
    pl.close()

    if rank == 0:
        fps = world * args.steps / elapsed
        if not any(v[1] for v in kt.values()):
            kt = {"k_none": (1e-9, 1)}          # KVAZZUP_BENCH_NOPROF=1: throughput-only run (no roofline)
        # dominant kernel = largest share of the timed region: average launch time x launches in the region (the
        # events sample every n-th picture, so the launch counts come from the picture types, not from the samples)
        n_idr = sum(1 for t in range(args.warmup, total) if t % 64 == 0)
        def launches(k):
            if k.startswith("k_intra"):
                return n_idr
            if k in ("k_me", "k_inter_recon", "k_inter_signal", "k_inter_recon<dec>"):
                return args.steps - n_idr
            return args.steps
        dom = max((k for k in kt if kt[k][1] > 0 and k.startswith("k_")), key=lambda k: kt[k][0] / kt[k][1] * launches(k))
        avg_s = kt[dom][0] / kt[dom][1] / 1e3
        ab = algorithmic_bytes(dom, cw, ch, args.me_range)
        achieved = ab / avg_s / 1e9
        # HBM traffic per launch: from the committed PMC passes of this same command (rocprofv3 --pmc cannot run inside
        # the bench); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.  null when no pass exists.
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_%s.json" % args.workload)))
            traffic = pmc["kernels"][dom]["traffic_bytes"]
            traffic_src = "profiles/r01_pmc_traffic_%s.json" % args.workload
        except Exception:
            pass
        out = {
            "metric": "hevc_encode_decode_fps", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"], "width": w, "height": h, "coded_width": cw, "coded_height": ch,
                       "frames_per_gpu": args.steps, "intra_period": 64, "qp": 32, "me_range": args.me_range,
                       "streams": world, "bytes_per_frame": round(nbytes / args.steps, 1), "decoder_frame_threads": D, "owf": args.owf, "sao": bool(args.sao), "me_early_termination": not args.full_search, "host_cpu_cores_busy": round(host_cores, 2), "host_cpu_budget_cores": round(budget, 1), "host_cpu_throttled_ms": round(throttled_ms, 1),
                       "input": "I420 resident in HBM", "output": "Annex-B AU on host + decoded I420 in HBM"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ab, "avg_launch_us": round(avg_s * 1e6, 2)},
            "kernels_us": {k: round(v[0] / v[1] * 1e3, 2) for k, v in kt.items() if v[1]},
            "filter_busy_ms_per_step": {"KvazaarFilter": busy[0], "WireAdapter": busy[1], "OpenHEVCFilter": busy[2]},
            "kernel_share_of_step": {k: round(v[0] / v[1] * launches(k) / (elapsed * 1e3), 4) for k, v in kt.items() if v[1]},
        }
        if args.full_search and "k_me" in kt and kt["k_me"][1]:
            # The motion search is integer VALU work, not streaming: its own ceiling is the issue rate of v_qsad_pk_u16_u8
            # (four 4-sample SADs per lane; measured ~24 cycles per wave instruction on gfx950, scratch/qsad_bench2.hip):
            # 1024 SIMDs x 2.4 GHz / 24 x 64 lanes x 16 sample differences.  Reported beside the HBM roofline the contract asks for.
            W = 2 * args.me_range + 1
            sads = (cw * ch // 1024) * W * W * 1024
            me_s = kt["k_me"][0] / kt["k_me"][1] / 1e3
            peak = 1024 * 2.4e9 / 24 * 64 * 16
            out["roofline_valu"] = {"kernel": "k_me", "bound": "valu (v_qsad_pk_u16_u8 issue rate)", "achieved": round(sads / me_s / 1e12, 2),
                                    "peak": round(peak / 1e12, 2), "unit": "T sample-differences/s", "frac": round(sads / me_s / peak, 4)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(w, h, args.cpu_frames, args.me_range)
            except Exception as e:       # the checker library is test infrastructure; report, do not fail the bench
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
