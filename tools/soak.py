#!/usr/bin/env python3
"""Soak of the batched bootstrap: fresh random ciphertexts and tables every round, EVERY output
decrypted and compared with the table's value, the largest output noise recorded.
  tools/soak.py [rounds] [params ids...]      (default 30 rounds of 4096 bootstraps under keys 3 and 4)
Needs a GPU; prints one JSON line per key."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fhe_amd.native import Context, KEY_CLIENT
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ids = [int(a) for a in sys.argv[2:]] or [3, 4]
    N, n, B = 2048, 918, 4096
    for pid in ids:
        ctx = Context(seed=os.urandom(32), params_id=pid)
        ctx.keygen()
        sk = np.frombuffer(ctx.export_keys(KEY_CLIENT)[-(n + N) * 8:], dtype=np.uint64)[n:].copy()
        rng = np.random.default_rng(int.from_bytes(os.urandom(8), "little"))
        wrong, worst, t0 = 0, 0, time.time()
        for it in range(rounds):
            if it % 50 == 49:
                print(f"key {pid}: round {it + 1} of {rounds}, {wrong} wrong", file=sys.stderr, flush=True)
            vals = rng.integers(0, 256, size=B // 4)
            blocks = np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1)
                                     for v in vals])
            msgs = np.array([(int(vals[i // 4]) >> (2 * (i % 4))) & 3 for i in range(B)])
            # message + carry: add a random carry part in the clear so that all 16 table entries are used
            carry = rng.integers(0, 4, size=B).astype(np.uint64)
            blocks[:, N] += carry << np.uint64(61)
            x = msgs + 4 * carry.astype(np.int64)
            tables = rng.integers(0, 16, size=(16, 16))
            luts = np.stack([ctx.make_lut(lambda v, t=t: int(t[v])) for t in tables])
            idx = rng.integers(0, 16, size=B).astype(np.uint32)
            out = ctx.pbs_batch(blocks, luts, idx)
            with np.errstate(over="ignore"):
                phase = out[:, N] - (out[:, :N] * sk[None, :]).sum(axis=1, dtype=np.uint64)
            m = ((phase + np.uint64(1 << 58)) >> np.uint64(59)) & np.uint64(15)
            want = tables[idx, x].astype(np.uint64)
            wrong += int((m != want).sum())
            err = np.abs((phase - (want << np.uint64(59))).astype(np.int64))
            worst = max(worst, int(err[m == want].max()))
        print(json.dumps({"params_id": pid, "key_image": "factored" if ctx.bsk_factored() else "as generated",
                          "bootstraps": rounds * B, "wrong": wrong, "worst_noise_log2": float(np.log2(worst)),
                          "half_step_log2": 58, "seconds": round(time.time() - t0, 1)}), flush=True)
        ctx.close()
        if wrong:
            raise SystemExit("decryption errors")


if __name__ == "__main__":
    main()
