#!/usr/bin/env python3
"""Soak of the integer operators: random operands, every operator and width, batches through
lfhe_evaluate_batch, every result decrypted and compared with the clear model (oracle/radix.py).
  tools/soak_ops.py [rounds] [batch] [params id]      (default 20 rounds of 64 instances, key 3)
Needs a GPU; prints one JSON line."""
from __future__ import annotations

import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fhe_amd.native import Context
    from oracle import radix
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    pid = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = Context(seed=os.urandom(32), params_id=pid)
    ctx.keygen()
    rnd = random.Random(int.from_bytes(os.urandom(8), "little"))
    binary = ["add", "sub", "mul", "eq", "ne", "ge", "gt", "le", "lt", "min", "max", "and", "or", "xor"]
    unary = ["not", "neg"]
    shifts = ["shl", "shr", "rotl", "rotr"]
    done, wrong, t0, p0 = 0, 0, time.time(), ctx.pbs_count()
    for it in range(rounds):
        for w in (8, 16, 32, 64):
            xs = [rnd.randrange(1 << w) if rnd.random() < 0.8 else rnd.choice([0, 1, (1 << w) - 1, 1 << (w - 1)]) for _ in range(batch)]
            ys = [rnd.randrange(1 << w) if rnd.random() < 0.8 else rnd.choice([0, 1, (1 << w) - 1, x]) for x in xs]
            A = [ctx.encrypt(x, w) for x in xs]
            B = [ctx.encrypt(y, w) for y in ys]
            for op in binary:
                out = ctx.evaluate_batch(op, w, A, B)
                got = [ctx.decrypt(o) for o in out]
                wrong += sum(g != radix.clear_op(op, w, x, y) for g, x, y in zip(got, xs, ys))
                done += batch
            for op in unary:
                out = ctx.evaluate_batch(op, w, A, None)
                got = [ctx.decrypt(o) for o in out]
                wrong += sum(g != radix.clear_op(op, w, x, 0) for g, x in zip(got, xs))
                done += batch
            amounts = [rnd.randrange(256) for _ in range(batch)]
            S = [ctx.encrypt(t, 8) for t in amounts]
            for op in shifts:
                out = ctx.evaluate_batch(op, w, A, S)
                got = [ctx.decrypt(o) for o in out]
                wrong += sum(g != radix.clear_op(op, w, x, t) for g, x, t in zip(got, xs, amounts))
                done += batch
            conds = [rnd.randrange(2) for _ in range(batch)]
            Cn = [ctx.encrypt(c, 1) for c in conds]
            out = ctx.evaluate_batch("select", w, A, B, Cn)
            got = [ctx.decrypt(o) for o in out]
            wrong += sum(g != (x if c else y) for g, x, y, c in zip(got, xs, ys, conds))
            done += batch
            # operands at noise level 2 (results of a select) into the operators that take them as they are
            sel = ctx.evaluate_batch("select", w, A, B, Cn)
            sv = [x if c else y for x, y, c in zip(xs, ys, conds)]
            for op in ("lt", "ge", "eq", "min", "max", "add", "sub"):
                out = ctx.evaluate_batch(op, w, sel, B)
                got = [ctx.decrypt(o) for o in out]
                wrong += sum(g != radix.clear_op(op, w, v, y) for g, v, y in zip(got, sv, ys))
                done += batch
            out = ctx.evaluate_batch("select", w, sel, ctx.evaluate_batch("select", w, B, A, Cn), Cn)
            got = [ctx.decrypt(o) for o in out]
            wrong += sum(g != x for g, x in zip(got, xs))      # c ? (c ? a : b) : (c ? b : a) = a
            done += batch
        print(f"round {it + 1} of {rounds}: {done} evaluations, {wrong} wrong", file=sys.stderr, flush=True)
    print(json.dumps({"params_id": pid, "batch": batch, "evaluations": done, "wrong": wrong, "bootstraps": ctx.pbs_count() - p0,
                      "slow_path_bootstraps": ctx.lift_fallbacks(), "seconds": round(time.time() - t0, 1)}), flush=True)
    ctx.close()
    if wrong:
        raise SystemExit("wrong results")


if __name__ == "__main__":
    main()
