#!/usr/bin/env python3
"""Which workgroup shape serves which launch size (key bits in triples): blind rotation of B ciphertexts with two
ciphertexts per workgroup, one per workgroup, and 512 threads per ciphertext, back to back (the library's own choice
also has two CUs per ciphertext for the keys over 2^64, up to one ciphertext per two CUs).
  tools/launch_size_probe.py [params id = 5] [sizes ...]      (needs a GPU)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from fhe_amd.native import Context
    pid = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    sizes = [int(a) for a in sys.argv[2:]] or [128, 256, 288, 320, 384, 448, 512, 640, 768, 896, 1024, 1536, 2048]
    dev = torch.device("cuda", 0)
    variants = (("two per workgroup", {"LFHE_PAIR_ABOVE": "0", "LFHE_WIDE_BELOW": "0", "LFHE_G64_SPLIT_BELOW": "0"}),
                ("one per workgroup", {"LFHE_PAIR_ABOVE": "-1", "LFHE_WIDE_BELOW": "0", "LFHE_G64_SPLIT_BELOW": "0"}),
                ("512 threads each", {"LFHE_WIDE_BELOW": "1000000", "LFHE_G64_SPLIT_BELOW": "0"}),
                ("library's choice", {}))
    rows = {}
    for name, env in variants:
        for k in ("LFHE_PAIR_ABOVE", "LFHE_WIDE_BELOW", "LFHE_G64_SPLIT_BELOW"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = Context(seed=bytes(range(32)), params_id=pid)
        ctx.keygen()
        N = ctx.N
        B = max(sizes)
        vals = np.random.default_rng(1).integers(0, 256, size=(B + 3) // 4)
        blocks = np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1) for v in vals])[:B]
        luts = np.stack([ctx.make_lut(lambda x: x)])
        d_in = torch.from_numpy(blocks.view(np.int64)).to(dev)
        d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
        d_out = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        for b in sizes:
            if name == "512 threads each" and b > 1024:
                continue
            step = lambda: ctx.pbs_batch_device_timed(b, d_in.data_ptr(), d_luts.data_ptr(), None, d_out.data_ptr(), 1)[1]
            step()
            ms = sorted(step() for _ in range(5))
            rows.setdefault(b, {})[name] = ms[2]
        ctx.close()
    print("ciphertexts  " + "  ".join(f"{n:>18s}" for n, _ in variants))
    for b in sizes:
        print(f"{b:11d}  " + "  ".join(f"{rows[b].get(n, float('nan')):18.2f}" for n, _ in variants), flush=True)


if __name__ == "__main__":
    main()
