"""Multi-GPU layout of the bootstrap path: one process per GPU, the batch is cut into contiguous
slices (an integer's blocks never straddle two GPUs), the server keys are generated once and
replicated with ONE broadcast per key image; there is no collective on the data path."""
from __future__ import annotations


def shard_bounds(total: int, world: int, rank: int) -> tuple:
    """Contiguous slice [lo, hi) of `total` independent units owned by `rank`; slices differ in
    size by at most one unit and cover every unit exactly once."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def replicate(dist, tensors: list, src: int = 0) -> None:
    """Broadcast each key image (a flat byte tensor, already allocated on every rank) from src."""
    for t in tensors:
        dist.broadcast(t, src=src)


def replicate_blob(dist, blob, src: int = 0):
    """Small host-side objects (the client key of a benchmark run)."""
    box = [blob]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def max_over_ranks(dist, value: float, device=None) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
