"""bench.py, part 4 of 5 -- BASELINE configs[4]: ONE 8K picture stream, its tile rows split over the ranks (kvazzup_amd/tilesplit.py); strong scaling."""
import json
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from .workloads import WORKLOADS
from .host import cpu_budget, init_dist, barrier_max


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
