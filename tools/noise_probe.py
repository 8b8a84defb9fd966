#!/usr/bin/env python3
"""Measured output noise of the bootstrap for each bootstrapping-key flavour: encrypts random
blocks with the library's own keys, bootstraps them with the identity table on the GPU and prints
the spread of  phase - delta * m  (phase = b - <a, s> with the exported GLWE secret key)."""
from __future__ import annotations

import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fhe_amd.native import Context, KEY_CLIENT
    N, n, B = 2048, 918, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    res = {}
    for name, pid in (("q64", 1), ("prime", 2), ("grouped", 3), ("q64_grouped", 4)):
        ctx = Context(seed=bytes((13 * i + 1) & 0xFF for i in range(32)), params_id=pid)
        ctx.keygen()
        blob = ctx.export_keys(KEY_CLIENT)
        sk = np.frombuffer(blob[-(n + N) * 8:], dtype=np.uint64)[n:].copy()
        rng = np.random.default_rng(5)
        vals = rng.integers(0, 256, size=B // 4)
        blocks = np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1)
                                 for v in vals])
        msgs = np.array([(int(vals[i // 4]) >> (2 * (i % 4))) & 3 for i in range(B)], dtype=np.uint64)
        out = ctx.pbs_batch(blocks, ctx.make_lut(lambda x: x)[None])

        def err(cts):
            with np.errstate(over="ignore"):
                phase = cts[:, N] - (cts[:, :N] * sk[None, :]).sum(axis=1, dtype=np.uint64)
                e = (phase - (msgs << np.uint64(59))).astype(np.int64)
            return e.astype(np.float64)
        e_in, e_out = err(blocks), err(out)
        res[name] = {"bootstraps": B, "input_noise_log2_std": math.log2(e_in.std()),
                     "output_noise_log2_std": math.log2(e_out.std()),
                     "output_noise_log2_max": math.log2(np.abs(e_out).max())}
        ctx.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
