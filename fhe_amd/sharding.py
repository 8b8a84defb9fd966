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


def key_image_tensors(ctx, device, owner: bool) -> list:
    """The two key images of a context as flat byte tensors on `device`: on the rank that owns the
    keys, zero-copy views of the library's own memory; elsewhere, fresh buffers of the same sizes to
    receive into.  (A host device is served too -- views of host memory -- which is how the CPU
    tests run this path on a stand-in context.)"""
    import torch
    bsk_bytes, ksk_bytes = ctx.device_keys_size()
    if not owner:
        return [torch.empty(bsk_bytes, dtype=torch.uint8, device=device),
                torch.empty(ksk_bytes, dtype=torch.uint8, device=device)]
    bp, _, kp, _ = ctx.device_keys()
    if device.type == "cuda":
        return [torch.as_tensor(_DevMem(bp, bsk_bytes), device=device),
                torch.as_tensor(_DevMem(kp, ksk_bytes), device=device)]
    import ctypes
    return [torch.frombuffer((ctypes.c_uint8 * n).from_address(p), dtype=torch.uint8)
            for p, n in ((bp, bsk_bytes), (kp, ksk_bytes))]


def replicated_context(dist, rank: int, world: int, device, seed: bytes, params_id: int, context_cls=None,
                       always_broadcast: bool = False):
    """A context with the same key set on every rank: rank 0 generates the keys, the two device key
    images go out with one broadcast each (straight out of / into device memory), the client key
    follows as a small host object.  Returns (context, tensors the context points at): keep the
    tensors alive as long as the context.  always_broadcast: run the collectives with one rank too
    (a communicator of size one), which exercises the wrapping of library memory on a single GPU."""
    import torch
    from .native import KEY_CLIENT
    if context_cls is None:
        from .native import Context as context_cls
    ctx = context_cls(seed=seed, device=device.index or 0, params_id=params_id)
    keep = []
    blob = None
    if rank == 0:
        ctx.keygen()
        blob = ctx.export_keys(KEY_CLIENT)
    if world > 1 or (always_broadcast and dist is not None):
        images = key_image_tensors(ctx, device, owner=rank == 0)
        if rank == 0 and device.type == "cuda":
            # The collective gets memory torch allocated itself on every rank: the owner's images are copied out
            # of the library's memory first (a device-to-device copy of a few hundred megabytes, once per key set),
            # so that the communication library never has to register or export memory another runtime allocated.
            images = [t.clone() for t in images]
        # The images were written on the library's own non-blocking stream, which no torch stream is
        # ordered against.  Every key load ends in a host-side wait on that stream, so they are complete
        # here; the two waits below make that ordering explicit at the hand-over instead of leaving it to
        # the loaders (torch's side too: the receive buffers' allocation and any pending work on them).
        if hasattr(ctx, "sync"):
            ctx.sync()
        if device.type == "cuda":
            torch.cuda.synchronize()
        replicate(dist, images)
        if device.type == "cuda":
            torch.cuda.synchronize()
        if rank != 0:
            ctx.attach_device_keys(images[0].data_ptr(), images[1].data_ptr())
            keep += images
        blob = replicate_blob(dist, blob if rank == 0 else None)
        if rank != 0:
            ctx.import_keys(KEY_CLIENT, blob)
    return ctx, keep
