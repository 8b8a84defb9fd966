#!/usr/bin/env python3
"""Single-operation latency probe: a few euint64 operations one at a time (run under
rocprofv3 --kernel-trace --stats to see how a bootstrap layer's time splits between kernels)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fhe_amd.native import Context

ctx = Context(seed=bytes(range(32)), params_id=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
ctx.keygen()
a, b = ctx.encrypt(0x123456789ABCDEF0, 64), ctx.encrypt(0x0FEDCBA987654321, 64)
for op in ("gt", "add", "mul"):
    ctx.evaluate(op, 64, a, b)          # program build + warm-up
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.evaluate(op, 64, a, b)
    print(op, (time.perf_counter() - t0) / 3 * 1e3, "ms")
ctx.close()
