"""bench.py, part 2 of 5 -- the host side of a run: CPU budget and per-thread CPU time, the rank launcher and the process group that brackets the
timed region, and the `cpu_baseline` leg (the only place the bench touches oracle/)."""
import subprocess
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BENCH = os.path.join(ROOT, "bench.py")


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
    procs = [subprocess.Popen([sys.executable, BENCH, "--cpu-worker", "%d,%d,%d,%d,%d" % (w, h, frames, me_range, 0x5EED0002 + 16 * i)],
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
        procs.append(subprocess.Popen([sys.executable, BENCH] + argv, env=env))
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
