"""Synthetic clips (numpy / torch twins of oracle/synth.c) and the one-stream-per-rank sharding that
bench.py uses for N > 1 (BASELINE configs[3]: independent streams, no collective on the data path)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_synth_twins_agree(kind):
    import torch
    from kvazzup_amd import synth
    for (w, h, t) in ((128, 64, 0), (320, 240, 3), (416, 240, 17)):
        a = synth.frame(kind, 0x5EED0001, w, h, t)
        assert np.array_equal(a, orc.synth_frame(kind, 0x5EED0001, w, h, t))
        assert np.array_equal(a, synth.frame_torch(kind, 0x5EED0001, w, h, t, torch.device("cpu")).numpy())
    if kind == 0:
        assert a.min() >= 16 and a[:416 * 240].max() <= 235


WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
import numpy as np, hashlib
import orc
from kvazzup_amd import sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
seed = sharding.stream_seed(2, rank)
w, h, frames = 192, 128, 4
e = orc.OracleEncoder(w, h, qp=32, period=64, me_range=4); d = orc.OracleDecoder()
t0 = time.perf_counter(); n = 0; dig = hashlib.md5()
for t in range(frames):
    au = e.encode(orc.synth_frame(0, seed, w, h, t)); dig.update(au)
    n += len(d.decode_au(au, t))
elapsed = time.perf_counter() - t0
tot, worst = sharding.aggregate(n, elapsed, dist)
digs = [None] * world
dist.all_gather_object(digs, dig.hexdigest())
if rank == 0:
    print("RESULT", tot, worst >= elapsed - 1e-9, len(set(digs)), world)
dist.destroy_process_group()
'''


def test_two_rank_stream_sharding_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 8            # 2 ranks x 4 pictures: whole-job count
    assert line[2] == "True"            # the reported time is the slowest rank's
    assert int(line[3]) == 2            # the two ranks coded different streams
