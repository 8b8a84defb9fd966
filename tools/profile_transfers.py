#!/usr/bin/env python3
"""ERC20 transfer batch alone (for rocprofv3 --kernel-trace --stats): wall time split into host
encryption, the fused evaluation call and decryption."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fhe_amd import workloads
from fhe_amd.native import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = Context(seed=bytes(range(32)), params_id=5)
ctx.keygen()
rnd = random.Random(1)
t0 = time.perf_counter()
F = [ctx.encrypt(rnd.randrange(1 << 40), 64) for _ in range(n)]
T = [ctx.encrypt(rnd.randrange(1 << 40), 64) for _ in range(n)]
A = [ctx.encrypt(rnd.randrange(1 << 41), 64) for _ in range(n)]
zero = ctx.encrypt(0, 64, trivial=True)
t1 = time.perf_counter()
workloads.transfers(ctx, F[:8], T[:8], A[:8], zero)   # warm-up: schedules built and uploaded
t2 = time.perf_counter()
n0 = ctx.pbs_count()
nf, nt = workloads.transfers(ctx, F, T, A, zero)
t3 = time.perf_counter()
pbs = ctx.pbs_count() - n0
print(f"encrypt {t1 - t0:.2f} s, warm-up {t2 - t1:.2f} s, {n} transfers {t3 - t2:.3f} s = {n / (t3 - t2):.1f}/s, "
      f"{pbs} bootstraps = {pbs / (t3 - t2):.0f}/s")
ctx.close()
