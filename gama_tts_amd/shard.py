"""Utterance sharding for multi-GPU runs (one process per GPU, no data-path collective).

Utterances are independent: rank r owns a contiguous block of the batch and
its own device; the only inter-rank traffic is a barrier and the MAX of the elapsed time.
"""


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` utterances owned by `rank` (sizes differ by <= 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(value, dist=None, device=None):
    """MAX-reduce a python float over the process group (identity without one)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
