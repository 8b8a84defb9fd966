#!/usr/bin/env python3
"""Latency of one operation and rate of a batch, for every integer operator at 8 .. 64 bits (and the
wide types for add / compare): tools/operator_table.py [params id] [batch] > table.json.
Operands and results in host memory (the shape of POST /evaluate), and the same batch again on
ciphertexts resident in the library's device store (handles in, handles out: the shape of the
executor events); every result of either route is decrypted and compared with the clear model.
Needs a GPU."""
from __future__ import annotations

import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fhe_amd.native import Context, program
    from oracle import radix
    pid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    ctx = Context(seed=os.urandom(32), params_id=pid)
    ctx.keygen()
    rnd = random.Random(11)
    rows = []
    cases = [(op, w, None) for w in (8, 16, 32, 64) for op in
             ("add", "sub", "mul", "lt", "eq", "min", "select", "and", "neg", "shl")]
    cases += [("add", 128, None), ("add", 256, None), ("lt", 256, None), ("mul", 128, None)]
    cases += [("mul", 64, 3), ("mul", 64, 1000003), ("div", 64, 10), ("rem", 64, 10), ("shl", 64, 13), ("and", 64, 0xFFFF)]
    for op, w, scalar in cases:
        m = (1 << w) - 1
        xs = [rnd.randrange(m + 1) for _ in range(batch)]
        ys = [rnd.randrange(m + 1) for _ in range(batch)]
        A = [ctx.encrypt(x, w) for x in xs]
        conds = None
        if scalar is not None:
            Bv, ys = [scalar] * batch, [scalar] * batch
        elif op == "shl":
            ys = [y % 256 for y in ys]
            Bv = [ctx.encrypt(y, 8) for y in ys]
        elif op == "neg":
            Bv = None
        else:
            Bv = [ctx.encrypt(y, w) for y in ys]
        if op == "select":
            cbits = [rnd.randrange(2) for _ in range(batch)]
            conds = [ctx.encrypt(c, 1) for c in cbits]
        want = [(x if c else y) for x, y, c in zip(xs, ys, cbits)] if op == "select" else \
               [radix.clear_op(op, w, x, y) for x, y in zip(xs, ys)]
        one = lambda i: ctx.evaluate(op, w, A[i], None if Bv is None else Bv[i], None if conds is None else conds[i])
        ok = ctx.decrypt(one(0)) == want[0]                       # warm-up: builds and uploads the schedule
        best = None
        for i in range(1, 4):
            t0 = time.perf_counter()
            out = one(i)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            ok = ok and ctx.decrypt(out) == want[i]
        n0 = ctx.pbs_count()
        ctx.evaluate_batch(op, w, A[:2], None if Bv is None else Bv[:2], None if conds is None else conds[:2])
        t0 = time.perf_counter()
        outs = ctx.evaluate_batch(op, w, A, Bv, conds)
        dt = time.perf_counter() - t0
        ok = ok and [ctx.decrypt(o) for o in outs] == want
        # the same batch on resident ciphertexts: only handles cross the bus inside the timed region
        hd = lambda tag, i: bytes([tag]) + i.to_bytes(4, "big") + bytes(27)
        hA = [hd(1, i) for i in range(batch)]
        for h, ct in zip(hA, A):
            ctx.store_put(h, ct)
        hB = Bv
        if Bv is not None and not isinstance(Bv[0], int):
            hB = [hd(2, i) for i in range(batch)]
            for h, ct in zip(hB, Bv):
                ctx.store_put(h, ct)
        hC = None
        if conds is not None:
            hC = [hd(3, i) for i in range(batch)]
            for h, ct in zip(hC, conds):
                ctx.store_put(h, ct)
        hO = [hd(4, i) for i in range(batch)]
        ctx.evaluate_batch_handles(op, w, hA[:2], None if hB is None else hB[:2], None if hC is None else hC[:2], hO[:2])
        t0 = time.perf_counter()
        ctx.evaluate_batch_handles(op, w, hA, hB, hC, hO)
        dt_res = time.perf_counter() - t0
        ok = ok and [ctx.decrypt(ctx.store_get(h)) for h in hO] == want
        for h in hA + hO + (hB if hB is not None and not isinstance(hB[0], int) else []) + (hC or []):
            ctx.store_drop(h)
        kw = {} if scalar is None else {"scalar": scalar}
        p = program(op, w, **kw)
        rows.append({"op": op, "width": w, "clear_operand": scalar, "bootstraps": p["pbs"], "layers": len(p["layers"]),
                     "latency_ms": round(best * 1e3, 2), "batch": batch, "per_second": round(batch / dt, 1),
                     "pbs_per_second": round(batch * p["pbs"] / dt),
                     "resident_per_second": round(batch / dt_res, 1), "resident_pbs_per_second": round(batch * p["pbs"] / dt_res),
                     "verified": ok})
        print(rows[-1], file=sys.stderr, flush=True)
    print(json.dumps({"params_id": pid, "rows": rows}), flush=True)
    ctx.close()
    if not all(r["verified"] for r in rows):
        raise SystemExit("wrong results")


if __name__ == "__main__":
    main()
