#!/usr/bin/env python3
"""Worker for the tile-row split tests: launched by torch.distributed.run with N ranks.  Every rank codes its band of one
picture stream; rank 0 assembles the access units and compares them with the CPU checker's encoder for the same tiling.
All ranks use GPU (LOCAL_RANK mod device count), so two ranks can share the one GPU of a test box (backend gloo)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    w, h, tile_rows, frames = (int(x) for x in sys.argv[1:5])
    pipelined = len(sys.argv) > 5 and sys.argv[5] == "pipelined"        # access units arrive one picture late (the gather completes during the next picture)
    bitrate = int(sys.argv[6]) if len(sys.argv) > 6 else 0               # > 0: rate control, every rank's controller fed with the assembled sizes
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)
    torch.cuda.set_device(dev)
    backend = "nccl" if ndev >= world else "gloo"
    dist.init_process_group(backend)
    from kvazzup_amd.tilesplit import BandEncoder
    import numpy as np
    import orc
    opts = (("qp", 30), ("period", 4), ("me-range", 16)) + ((("bitrate", bitrate),) if bitrate else ())
    be = BandEncoder(w, h, tile_rows, rank, world, options=opts, device=dev, dist=dist, pipelined=pipelined)
    oe = orc.OracleEncoder(w, h, qp=30, period=4, me_range=16, tile_rows=tile_rows, bitrate=bitrate) if rank == 0 else None
    od = orc.OracleDecoder() if rank == 0 else None
    bad = 0
    got = []
    for t in range(frames):
        frame = orc.synth_frame(0, 11, w, h, t)
        d = torch.from_numpy(frame).to("cuda:%d" % dev)
        au = be.encode(d.data_ptr())
        if au is not None:
            got.append(au)
    au = be.flush()
    if au is not None:
        got.append(au)
    if rank == 0:
        assert len(got) == frames, (len(got), frames)
        for t in range(frames):
            frame = orc.synth_frame(0, 11, w, h, t)
            want = oe.encode(frame)
            au = got[t]
            if au != want:
                bad += 1
                print("picture %d: split encoder %d bytes, checker %d bytes, equal=%s" % (t, len(au), len(want), au == want), flush=True)
            else:
                pics = od.decode_au(au, t)
                assert len(pics) == 1 and np.array_equal(pics[0]["i420"], oe.recon()), t
    if rank == 0:
        print("tilesplit: %d ranks (%s), %dx%d, %d tile rows, %d pictures, halo bytes per picture and rank %.0f: %s" % (
            world, backend, w, h, tile_rows, frames, be.halo_bytes_exchanged / max(1, frames), "OK" if not bad else "%d MISMATCHES" % bad), flush=True)
    be.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
