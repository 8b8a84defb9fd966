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


class _DevMem:
    """Zero-copy view of library-owned device memory for torch (cuda array interface)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def replicated_context(dist, rank: int, world: int, device, seed: bytes, params_id: int):
    """A context with the same key set on every rank: rank 0 generates the keys, the two device key
    images go out with one broadcast each (straight out of / into device memory), the client key
    follows as a small host object.  Returns (context, tensors the context points at): keep the
    tensors alive as long as the context."""
    import torch
    from .native import Context, KEY_CLIENT
    ctx = Context(seed=seed, device=device.index or 0, params_id=params_id)
    keep = []
    blob = None
    if rank == 0:
        ctx.keygen()
        blob = ctx.export_keys(KEY_CLIENT)
    if world > 1:
        bsk_bytes, ksk_bytes = ctx.device_keys_size()
        if rank == 0:
            bp, _, kp, _ = ctx.device_keys()
            tb = torch.as_tensor(_DevMem(bp, bsk_bytes), device=device)
            tk = torch.as_tensor(_DevMem(kp, ksk_bytes), device=device)
        else:
            tb = torch.empty(bsk_bytes, dtype=torch.uint8, device=device)
            tk = torch.empty(ksk_bytes, dtype=torch.uint8, device=device)
        replicate(dist, [tb, tk])
        torch.cuda.synchronize()
        if rank != 0:
            ctx.attach_device_keys(tb.data_ptr(), tk.data_ptr())
            keep += [tb, tk]
        blob = replicate_blob(dist, blob if rank == 0 else None)
        if rank != 0:
            ctx.import_keys(KEY_CLIENT, blob)
    return ctx, keep
