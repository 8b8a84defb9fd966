#!/usr/bin/env python3
"""Operator-level workloads of BASELINE.json (configs 3-5): euint64 mul / compare latency and
throughput, the confidential-ERC20 transfer batch, the blind-auction argmax tournament.
Every result is decrypted and checked against the clear model.  Prints one JSON object on rank 0.

One GPU: `python bench_workloads.py`.  N GPUs of one node:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_workloads.py --gpus N`
-- the independent instances (multiplications, transfers, bids) are cut into contiguous slices, one
per rank; keys are generated on rank 0 and broadcast once; the only data exchange is the argmax's
final round over one (value, index) pair per rank."""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def column(cts):
    """Ciphertexts of one width as an (n, size) byte array."""
    import numpy as np
    rows = [np.frombuffer(c, dtype=np.uint8) for c in cts]
    return np.stack(rows) if rows else np.zeros((0, 0), dtype=np.uint8)


def main(argv=None, backend=None):
    """backend: what bench.GpuBackend provides (device of a rank, context with replicated keys);
    the CPU tests inject a plaintext stand-in to run this control flow under gloo."""
    if backend is None:
        from bench import GpuBackend
        backend = GpuBackend()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--transfers", type=int, default=0,
                    help="total over all GPUs; default 1250 per GPU (10 000 over a node of eight)")
    ap.add_argument("--bids", type=int, default=1000, help="total over all GPUs")
    ap.add_argument("--mul-batch", type=int, default=256,
                    help="random pairs, total over all GPUs (the reference's vectors are added)")
    ap.add_argument("--bsk-modulus", choices=["q64", "prime", "grouped", "q64_grouped", "grouped3", "q64_grouped3"], default="grouped3",
                    help="bootstrapping key: over 2^64, over the NTT prime, or over the prime with the key bits in pairs")
    args = ap.parse_args(argv)

    import torch
    from fhe_amd import workloads
    from fhe_amd.sharding import max_over_ranks, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = backend.device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend.collective == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend.collective, rank=rank, world_size=world)
    seed = bytes((11 * i + 5) & 0xFF for i in range(32))
    ctx, keep = backend.context(dist, rank, world, dev, seed, {"q64": 1, "prime": 2, "grouped": 3, "q64_grouped": 4, "grouped3": 5, "q64_grouped3": 6}[args.bsk_modulus])

    def timed(fn):
        """Wall time of fn over all ranks: barrier on both sides, maximum over ranks."""
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = max_over_ranks(dist, dt, dev)
        return out, dt

    def total(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        return t.item()

    rnd = random.Random(2026)          # same stream on every rank: each encrypts its own slice only
    M64 = (1 << 64) - 1
    res = {"n_gpus": world, "bsk_modulus": args.bsk_modulus}

    # ---- config 3: euint64 mul + compare (reference vector first: fhevmOperations6.ts:647-660)
    ka, kb, kp = 4293652672, 4293367238, 18434227713315959936
    a, b = ctx.encrypt(ka, 64), ctx.encrypt(kb, 64)
    ctx.evaluate("mul", 64, a, b)  # warm-up: builds and uploads the schedules
    ctx.evaluate("gt", 64, a, b)
    ctx.evaluate("add", 64, a, b)
    n0 = ctx.pbs_count()
    t0 = time.perf_counter(); out = ctx.evaluate("mul", 64, a, b); t_mul = time.perf_counter() - t0
    pbs_mul = ctx.pbs_count() - n0
    assert ctx.decrypt(out) == kp
    t0 = time.perf_counter(); out = ctx.evaluate("gt", 64, a, b); t_gt = time.perf_counter() - t0
    assert ctx.decrypt(out) == 1
    t0 = time.perf_counter(); out = ctx.evaluate("add", 64, a, b); t_add = time.perf_counter() - t0
    assert ctx.decrypt(out) == (ka + kb) & M64
    nb = args.mul_batch
    # the reference's euint64 vectors first (fhevmOperations6.ts:647-705 mul, :1067-1125 gt), then random pairs
    kat = [(4293652672, 4293367238), (4293367238, 4293367238),
           (18442678758156446977, 18443460298071844659), (18442678758156446973, 18442678758156446977),
           (18442678758156446977, 18442678758156446977), (18442678758156446977, 18442678758156446973)]
    xs = [p[0] for p in kat] + [rnd.randrange(1 << 64) for _ in range(nb)]
    ys = [p[1] for p in kat] + [rnd.randrange(1 << 64) for _ in range(nb)]
    nb = len(xs)
    lo, hi = shard_bounds(nb, world, rank)
    # operand columns as (n, size) byte arrays: the C ABI's own layout, no copies in the binding
    A, B = column(ctx.encrypt(x, 64) for x in xs[lo:hi]), column(ctx.encrypt(y, 64) for y in ys[lo:hi])
    outs, t_mulb = timed(lambda: ctx.evaluate_batch("mul", 64, A, B))
    assert [ctx.decrypt(bytes(o)) for o in outs] == [(x * y) & M64 for x, y in zip(xs[lo:hi], ys[lo:hi])]
    outs, t_gtb = timed(lambda: ctx.evaluate_batch("gt", 64, A, B))
    assert [ctx.decrypt(bytes(o)) for o in outs] == [int(x > y) for x, y in zip(xs[lo:hi], ys[lo:hi])]
    res["euint64"] = {"mul_latency_ms": t_mul * 1e3, "mul_pbs": pbs_mul, "gt_latency_ms": t_gt * 1e3,
                      "add_latency_ms": t_add * 1e3, "mul_batch": nb, "mul_per_s": nb / t_mulb,
                      "mul_batch_pbs_per_s": nb * pbs_mul / t_mulb, "gt_per_s": nb / t_gtb}

    # ---- config 4: ERC20 transfers (le64 + select64 + sub64 + add64 each), sharded by transfer
    n = args.transfers or 1250 * world
    bal_from = [rnd.randrange(1 << 40) for _ in range(n)]
    bal_to = [rnd.randrange(1 << 40) for _ in range(n)]
    amount = [rnd.randrange(1 << 41) for _ in range(n)]   # about half exceed the balance
    lo, hi = shard_bounds(n, world, rank)
    F, T, AM = (column(ctx.encrypt(v, 64) for v in vs[lo:hi]) for vs in (bal_from, bal_to, amount))
    zero = ctx.encrypt(0, 64, trivial=True)
    n0 = ctx.pbs_count()
    (nf, nt), t_erc = timed(lambda: workloads.transfers(ctx, F, T, AM, zero))
    pbs_erc = total(ctx.pbs_count() - n0)
    for i in range(0, hi - lo, max(1, (hi - lo) // 64)):
        g = lo + i
        v = amount[g] if amount[g] <= bal_from[g] else 0
        assert (ctx.decrypt(bytes(nf[i])), ctx.decrypt(bytes(nt[i]))) == (bal_from[g] - v, (bal_to[g] + v) & M64), g
    res["erc20"] = {"transfers": n, "seconds": t_erc, "transfers_per_s": n / t_erc, "pbs": pbs_erc,
                    "pbs_per_transfer": pbs_erc / n, "pbs_per_s": pbs_erc / t_erc}

    # ---- config 5: blind-auction argmax over euint32 bids, sharded by bid, winners reduced on rank 0
    nbid = args.bids
    bids = rnd.sample(range(1 << 32), nbid)
    lo, hi = shard_bounds(nbid, world, rank)
    vals = column(ctx.encrypt(bids[i], 32) for i in range(lo, hi))
    idxs = column(ctx.encrypt(i, 16, trivial=True) for i in range(lo, hi))
    n0 = ctx.pbs_count()
    win, t_arg = timed(lambda: workloads.sharded_argmax(dist, ctx, vals, idxs, rank, world, local=True))
    pbs_arg = total(ctx.pbs_count() - n0)
    if rank == 0:
        assert ctx.decrypt(bytes(win[0])) == max(bids) and ctx.decrypt(bytes(win[1])) == bids.index(max(bids))
    res["argmax"] = {"bids": nbid, "seconds": t_arg, "pbs": pbs_arg, "pbs_per_s": pbs_arg / t_arg}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    del keep


if __name__ == "__main__":
    main()
