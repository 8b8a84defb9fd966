#!/usr/bin/env python3
"""Operator-level workloads of BASELINE.json (configs 3-5) on one GPU: euint64 mul / compare latency
and throughput, the confidential-ERC20 transfer batch, the blind-auction argmax tournament.
Every result is decrypted and checked against the clear model.  Prints one JSON object."""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def timed(fn):
    t0 = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transfers", type=int, default=512)
    ap.add_argument("--bids", type=int, default=1000)
    ap.add_argument("--mul-batch", type=int, default=64)
    ap.add_argument("--bsk-modulus", choices=["q64", "prime"], default="prime")
    args = ap.parse_args()
    from fhe_amd.native import Context
    ctx = Context(seed=bytes((11 * i + 5) & 0xFF for i in range(32)), params_id=2 if args.bsk_modulus == "prime" else 1)
    ctx.keygen()
    rnd = random.Random(2026)
    M64 = (1 << 64) - 1
    res = {}

    # ---- config 3: euint64 mul + compare (reference vectors first: fhevmOperations6.ts:647-705, :1067)
    kat = [(4293652672, 4293367238, 18434227713315959936)]
    a, b = ctx.encrypt(kat[0][0], 64), ctx.encrypt(kat[0][1], 64)
    ctx.evaluate("mul", 64, a, b)  # warm-up: builds and uploads the schedule
    ctx.evaluate("gt", 64, a, b)
    n0 = ctx.pbs_count()
    out, t_mul = timed(lambda: ctx.evaluate("mul", 64, a, b))
    pbs_mul = ctx.pbs_count() - n0
    assert ctx.decrypt(out) == kat[0][2]
    out, t_gt = timed(lambda: ctx.evaluate("gt", 64, a, b))
    assert ctx.decrypt(out) == 1
    out, t_add = timed(lambda: ctx.evaluate("add", 64, a, b))
    assert ctx.decrypt(out) == (kat[0][0] + kat[0][1]) & M64
    nb = args.mul_batch
    xs = [rnd.randrange(1 << 64) for _ in range(nb)]
    ys = [rnd.randrange(1 << 64) for _ in range(nb)]
    A, B = [ctx.encrypt(x, 64) for x in xs], [ctx.encrypt(y, 64) for y in ys]
    outs, t_mulb = timed(lambda: ctx.evaluate_batch("mul", 64, A, B))
    assert [ctx.decrypt(o) for o in outs] == [(x * y) & M64 for x, y in zip(xs, ys)]
    outs, t_gtb = timed(lambda: ctx.evaluate_batch("gt", 64, A, B))
    assert [ctx.decrypt(o) for o in outs] == [int(x > y) for x, y in zip(xs, ys)]
    res["euint64"] = {"mul_latency_ms": t_mul * 1e3, "mul_pbs": pbs_mul, "gt_latency_ms": t_gt * 1e3,
                      "add_latency_ms": t_add * 1e3, "mul_batch": nb, "mul_per_s": nb / t_mulb,
                      "mul_batch_pbs_per_s": nb * pbs_mul / t_mulb, "gt_per_s": nb / t_gtb}

    # ---- config 4: ERC20 transfers: le64 + select64 + add64 + sub64 each
    # (tests/fhevm-suite/e2e/contracts/EncryptedERC20.sol:90,210-215)
    n = args.transfers
    bal_from = [rnd.randrange(1 << 40) for _ in range(n)]
    bal_to = [rnd.randrange(1 << 40) for _ in range(n)]
    amount = [rnd.randrange(1 << 41) for _ in range(n)]   # about half exceed the balance
    F, T, AM = ([ctx.encrypt(v, 64) for v in vs] for vs in (bal_from, bal_to, amount))
    zero = [ctx.encrypt(0, 64, trivial=True)] * n
    n0 = ctx.pbs_count()

    def transfers():
        ok = ctx.evaluate_batch("le", 64, AM, F)
        v = ctx.evaluate_batch("select", 64, AM, zero, ok)
        return ctx.evaluate_batch("sub", 64, F, v), ctx.evaluate_batch("add", 64, T, v)
    (nf, nt), t_erc = timed(transfers)
    pbs_erc = ctx.pbs_count() - n0
    for i in range(0, n, max(1, n // 64)):
        v = amount[i] if amount[i] <= bal_from[i] else 0
        assert (ctx.decrypt(nf[i]), ctx.decrypt(nt[i])) == (bal_from[i] - v, (bal_to[i] + v) & M64), i
    res["erc20"] = {"transfers": n, "seconds": t_erc, "transfers_per_s": n / t_erc, "pbs": pbs_erc,
                    "pbs_per_transfer": pbs_erc / n, "pbs_per_s": pbs_erc / t_erc}

    # ---- config 5: blind-auction argmax over euint32 bids: per pair gt32 + select32 (value) + select16 (index)
    nbid = args.bids
    bids = rnd.sample(range(1 << 32), nbid)
    vals = [ctx.encrypt(v, 32) for v in bids]
    idxs = [ctx.encrypt(i, 16, trivial=True) for i in range(nbid)]
    n0 = ctx.pbs_count()

    def argmax(vals, idxs):
        while len(vals) > 1:
            half = len(vals) // 2
            L, R = vals[:half], vals[half:2 * half]
            LI, RI = idxs[:half], idxs[half:2 * half]
            g = ctx.evaluate_batch("gt", 32, L, R)
            nv = ctx.evaluate_batch("select", 32, L, R, g)
            ni = ctx.evaluate_batch("select", 16, LI, RI, g)
            vals, idxs = nv + vals[2 * half:], ni + idxs[2 * half:]
        return vals[0], idxs[0]
    (mv, mi), t_arg = timed(lambda: argmax(vals, idxs))
    pbs_arg = ctx.pbs_count() - n0
    assert ctx.decrypt(mv) == max(bids) and ctx.decrypt(mi) == bids.index(max(bids))
    res["argmax"] = {"bids": nbid, "seconds": t_arg, "pbs": pbs_arg, "pbs_per_s": pbs_arg / t_arg}
    print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
