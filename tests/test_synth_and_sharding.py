"""Synthetic clips (numpy / torch twins of oracle/synth.c) and the one-stream-per-rank sharding that
bench.py uses for N > 1 (BASELINE configs[3]: independent streams, no collective on the data path): bench.py's own code, on CPU."""
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
import os, sys
sys.path.insert(0, sys.argv[1])
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ranks = bench.StreamRanks(world, int(os.environ["LOCAL_RANK"]), need_device=False)     # bench.py's own process-group plumbing, without a GPU
assert ranks.backend == "gloo"
ranks.sync()                                                    # the barrier that brackets the timed region
worst = ranks.sync(1.0 + rank)                                  # max over ranks of the elapsed time
seeds = [bench.stream_seed(2, r) for r in range(world)]
ranks.sync()
if rank == 0:
    print("RESULT", worst, len(set(seeds)), world)
ranks.close()
'''


def test_two_rank_stream_sharding_gloo(tmp_path):
    """the path `bench.py --gpus 2` takes between its ranks (StreamRanks: gloo process group, barrier, max over ranks; one stream seed per
    rank), world size 2 on CPU.  The same command end to end, codecs included, runs in tests/test_gpu_configs.py."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) == 2.0        # the reported time is the slowest rank's
    assert int(line[2]) == 2            # the two ranks code different streams
    assert int(line[3]) == 2


def test_launch_ranks_stops_the_job_when_a_rank_dies(tmp_path):
    """bench.py --gpus N starts its ranks itself; a rank that fails must end the job (exit code != 0) instead of leaving the others in a
    barrier.  Here (no GPU) every rank fails at once: the launcher returns promptly with their exit code."""
    import time
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=120)
    import kvazzup_amd
    if kvazzup_amd.load_library().kvzx_device_count() == 0:
        assert out.returncode != 0 and time.time() - t0 < 100, (out.returncode, out.stderr[-500:])
