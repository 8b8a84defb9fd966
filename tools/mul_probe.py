import sys, os, time
sys.path.insert(0, os.getcwd())
from fhe_amd.native import Context
ctx = Context(seed=bytes(range(32)), params_id=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
ctx.keygen()
a, b = ctx.encrypt(0x123456789ABCDEF0, 64), ctx.encrypt(0x0FEDCBA987654321, 64)
ctx.evaluate("mul", 64, a, b)
t0 = time.perf_counter()
for _ in range(8):
    ctx.evaluate("mul", 64, a, b)
print("mul", (time.perf_counter() - t0) / 8 * 1e3, "ms")
ctx.close()
