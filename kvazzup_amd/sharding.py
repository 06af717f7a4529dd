"""How the path shards over GPUs: whole streams, one per rank -- the unit of a
multi-party call (/root/reference/src/media/processing/filtergraph.cpp:561-589 builds one receive graph
per peer).  No collective touches the data path; torch.distributed only brackets the timed region."""

BASE_SEED = 0x5EED0000


def stream_seed(cfg_index, rank):
    """every rank encodes its own synthetic stream (uvgx-synth-v1 seed = 0x5EED0000 + cfg, shifted per stream)"""
    return BASE_SEED + cfg_index + 16 * rank


def aggregate(units, elapsed, dist=None):
    """whole-job units (sum over ranks) and the time of the slowest rank (max over ranks)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units, elapsed
    import torch
    t = torch.tensor([float(units), 0.0], dtype=torch.float64)
    m = torch.tensor([float(elapsed)], dtype=torch.float64)
    dev = None
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        t, m = t.to(dev), m.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(round(t[0].item())), float(m.item())
