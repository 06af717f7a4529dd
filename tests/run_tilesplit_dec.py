#!/usr/bin/env python3
"""Worker for the split-decoder test: launched by torch.distributed.run with N ranks.  Every rank feeds the same access units (the CPU
checker's encoder, tile rows, vectors confined to their tile) to its BandDecoder; halos go over torch.distributed; every rank compares its
band's rows with the checker's reconstruction and rank 0 collects the verdicts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    w, h, tile_rows, frames = (int(x) for x in sys.argv[1:5])
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)
    torch.cuda.set_device(dev)
    backend = "nccl" if ndev >= world else "gloo"
    dist.init_process_group(backend)
    from kvazzup_amd.codec import split_nals
    from kvazzup_amd.tilesplit import BandDecoder
    import orc
    oe = orc.OracleEncoder(w, h, qp=28, period=4, me_range=16, tile_rows=tile_rows, subme=2)
    bd = BandDecoder((h + 63) // 64, tile_rows, rank, world, device=dev, dist=dist)
    bad = 0
    for t in range(frames):
        au = oe.encode(orc.synth_frame(0, 23, w, h, t))
        want = oe.recon()
        ready = False
        for nal in split_nals(au):
            ready = bd.feed(nal, t)
        assert ready is True
        pic = bd.finish_exchange()
        y0, y1 = pic["rows"]
        got = pic["i420"]
        ok = np.array_equal(got[:w * h].reshape(h, w)[y0:y1], want[:w * h].reshape(h, w)[y0:y1])
        for c in range(2):
            o = w * h + c * (w * h // 4)
            ok = ok and np.array_equal(got[o:o + w * h // 4].reshape(h // 2, w // 2)[y0 // 2:y1 // 2], want[o:o + w * h // 4].reshape(h // 2, w // 2)[y0 // 2:y1 // 2])
        if not ok:
            bad += 1
            print("rank %d picture %d: band rows %d..%d differ from the checker" % (rank, t, y0, y1), flush=True)
    v = torch.tensor([bad], dtype=torch.int64, device="cuda:%d" % dev if backend == "nccl" else "cpu")
    dist.all_reduce(v)
    if rank == 0:
        print("split decoder: %d ranks (%s), %dx%d, %d tile rows, %d pictures: %s" % (world, backend, w, h, tile_rows, frames, "OK" if int(v.item()) == 0 else "%d MISMATCHES" % int(v.item())), flush=True)
    bd.close(); oe.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if int(v.item()) else 0)


if __name__ == "__main__":
    main()
