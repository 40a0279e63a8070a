"""Batch sharding across ranks (one process per GPU).

Transforms of a batch are independent (the reference's `DFT` methods act on one slice at a
time, src/dft.rs:6-11), so multi-GPU execution is a contiguous partition of the batch with no
data-path collective.  `torch.distributed` is used only to line ranks up around a timed region
and to combine their clocks.
"""


def shard_range(batch, world_size, rank):
    """Contiguous [first, first + count) of `batch` transforms owned by `rank`: ceil(batch / world)
    per rank, the last ranks possibly short or empty (same rule as nttx_*_host's device split)."""
    if batch < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad shard request")
    per = -(-batch // world_size)
    first = min(rank * per, batch)
    return first, min(per, batch - first)


def max_over_ranks(seconds, dist=None, device=None):
    """Elapsed time of the slowest rank (the job's time).  `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank, world_size, steps, seconds):
    """Whole-job units per second when every rank processed `units_per_rank` units per step."""
    return units_per_rank * world_size * steps / seconds
