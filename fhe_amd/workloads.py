"""The operator workloads of the benchmark configurations, written against the engine surface
(`evaluate_batch`, `encrypt`) so that they run on the GPU context and, in the CPU tests, on a
plaintext stand-in.  Sharding follows fhe_amd/sharding.py: every rank owns a contiguous slice of
the independent instances, there is no collective on the data path; only the argmax needs one
exchange, of one (value, index) pair per rank, at the very end.

  * transfers: the confidential-ERC20 transfer (tests/fhevm-suite/e2e/contracts/EncryptedERC20.sol:
    87-93, 208-220): ok = amount <= from; v = ok ? amount : 0; from -= v; to += v.
  * argmax: a blind-auction tournament over encrypted bids: per pair gt + select(value) +
    select(index).
"""
from __future__ import annotations

from .sharding import shard_bounds


def _cat(a, b):
    """Concatenation of two ciphertext columns (lists, or (n, size) byte arrays)."""
    if isinstance(a, list):
        return a + list(b)
    import numpy as np
    return np.concatenate([a, b]) if len(b) else a


# the transfer as one fused schedule: values 0 = from, 1 = to, 2 = amount; 3 = amount <= from;
# 4 = select(3, amount, 0); 5 = from - 4; 6 = to + 4.  Subtraction and addition share their layers.
def transfer_steps(width: int = 64):
    return [("le", width, 2, 0, None), ("select", width, 2, ("clear", 0), 3), ("sub", width, 0, 4, None),
            ("add", width, 1, 4, None)]


def transfers(backend, bal_from, bal_to, amount, zero, width: int = 64, fused: bool = True):
    """(new_from[], new_to[]) for n independent transfers: one fused schedule when the backend has
    `evaluate_fused`, else four batched operator calls.  Columns are lists of ciphertexts or, for
    the GPU context, (n, size) byte arrays, which cross the binding without being copied."""
    n = len(amount)
    if n == 0:
        return [], []
    if fused and hasattr(backend, "evaluate_fused"):
        nf, nt = backend.evaluate_fused(transfer_steps(width), [width] * 3, [bal_from, bal_to, amount], [5, 6],
                                        [width, width])
        return nf, nt
    ok = backend.evaluate_batch("le", width, amount, bal_from)
    v = backend.evaluate_batch("select", width, amount, [zero] * n, ok)
    return backend.evaluate_batch("sub", width, bal_from, v), backend.evaluate_batch("add", width, bal_to, v)


def argmax(backend, vals, idxs, width: int = 32, idx_width: int = 16):
    """(max value, its index) by a binary tournament; ties keep the right-hand entry's rival, i.e.
    the strictly greater left value wins."""
    if isinstance(vals, (list, tuple)):
        vals, idxs = list(vals), list(idxs)
    if not len(vals):
        raise ValueError("empty tournament")
    fused = hasattr(backend, "evaluate_fused")
    # one round as one schedule: values 0 = L, 1 = R, 2 = LI, 3 = RI; 4 = L > R; 5, 6 = the two selects
    steps = [("gt", width, 0, 1, None), ("select", width, 0, 1, 4), ("select", idx_width, 2, 3, 4)]
    while len(vals) > 1:
        half = len(vals) // 2
        L, R = vals[:half], vals[half:2 * half]
        LI, RI = idxs[:half], idxs[half:2 * half]
        if fused:
            nv, ni = backend.evaluate_fused(steps, [width, width, idx_width, idx_width], [L, R, LI, RI], [5, 6],
                                            [width, idx_width])
            vals, idxs = _cat(nv, vals[2 * half:]), _cat(ni, idxs[2 * half:])
            continue
        g = backend.evaluate_batch("gt", width, L, R)
        nv = backend.evaluate_batch("select", width, L, R, g)
        ni = backend.evaluate_batch("select", idx_width, LI, RI, g)
        vals, idxs = nv + vals[2 * half:], ni + idxs[2 * half:]
    return vals[0], idxs[0]


def sharded_argmax(dist, backend, vals, idxs, rank: int, world: int, width: int = 32,
                   idx_width: int = 16, local: bool = False):
    """Every rank reduces its own slice of the bids, the (at most `world`) shard winners are gathered
    host-side and reduced on rank 0.  `vals` / `idxs` are the full lists on every rank (a rank only
    touches its slice) or, with local=True, just this rank's slice (lists or byte arrays).
    Returns the winner on rank 0, None elsewhere."""
    if local:
        mine = argmax(backend, vals, idxs, width, idx_width) if len(vals) else None
    else:
        lo, hi = shard_bounds(len(vals), world, rank)
        mine = argmax(backend, vals[lo:hi], idxs[lo:hi], width, idx_width) if hi > lo else None
    if dist is None or world == 1:
        return mine
    box = [None] * world
    dist.all_gather_object(box, mine)
    if rank != 0:
        return None
    winners = [w for w in box if w is not None]
    return argmax(backend, [bytes(w[0]) for w in winners], [bytes(w[1]) for w in winners], width, idx_width)
