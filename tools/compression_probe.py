#!/usr/bin/env python3
"""Times lfhe_compress / lfhe_decompress on 4096 blocks (128 euint64) through the C ABI, host buffers."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fhe_amd.native import Context

ctx = Context(seed=bytes(range(32)), params_id=5)
ctx.keygen()
t0 = time.perf_counter()
ctx.compression_keygen()
t1 = time.perf_counter()
cts = [ctx.encrypt((0x9E3779B97F4A7C15 * (i + 1)) & ((1 << 64) - 1), 64) for i in range(128)]
ctx.decompress(ctx.compress(cts[:4]))                       # warm-up
t2 = time.perf_counter()
blob = ctx.compress(cts)
t3 = time.perf_counter()
back = ctx.decompress(blob)
t4 = time.perf_counter()
assert [ctx.decrypt(c) for c in back[:8]] == [ctx.decrypt(c) for c in cts[:8]]
plain = sum(len(c) for c in cts)
print(f"compression keys {t1 - t0:.2f} s; 4096 blocks: {plain} -> {len(blob)} bytes (x{plain / len(blob):.0f}), "
      f"compress {1e3 * (t3 - t2):.1f} ms, decompress {1e3 * (t4 - t3):.1f} ms")
ctx.close()
