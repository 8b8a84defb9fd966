#!/usr/bin/env python3
"""Does a blind-rotation launch run slower after the GPU has been idle?  Batches of 1024 bootstraps
back to back, then with pauses between them: tools/idle_gap_probe.py [batch]   (needs a GPU)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from fhe_amd.native import Context
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda", 0)
    ctx = Context(seed=bytes(range(32)), params_id=5)
    ctx.keygen()
    N = ctx.N
    vals = np.random.default_rng(1).integers(0, 256, size=(B + 3) // 4)
    blocks = np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1) for v in vals])[:B]
    luts = np.stack([ctx.make_lut(lambda x: x)])
    d_in = torch.from_numpy(blocks.view(np.int64)).to(dev)
    d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    step = lambda: ctx.pbs_batch_device_timed(B, d_in.data_ptr(), d_luts.data_ptr(), None, d_out.data_ptr(), 1)[1]
    step()
    for pause in (0.0, 0.003, 0.03, 0.3):
        ms = []
        for _ in range(6):
            time.sleep(pause)
            ms.append(step())
        print(f"pause {pause * 1e3:6.1f} ms before each launch: blind rotation of {B} takes "
              f"{min(ms):.2f} .. {max(ms):.2f} ms (median {sorted(ms)[len(ms) // 2]:.2f})", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
