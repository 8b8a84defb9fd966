"""One rank, RCCL: the product's key replication (fhe_amd.sharding.replicated_context) with its real
collectives on a communicator of size one, then the part of a non-zero rank played in the same
process.  Run as a fresh process by tests/test_zz_two_ranks.py; prints a line per stage."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from fhe_amd.native import Context, KEY_CLIENT
    from fhe_amd.sharding import key_image_tensors, max_over_ranks, replicate, replicated_context
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print("process group up", flush=True)
    seed = bytes(range(32))
    ctx, keep = replicated_context(dist, 0, 1, dev, seed, 3, always_broadcast=True)
    print("keys generated, images broadcast over RCCL from library memory", flush=True)
    dst = Context(seed=seed, params_id=3)
    try:
        recv = key_image_tensors(dst, dev, owner=False)
        for r, s in zip(recv, key_image_tensors(ctx, dev, owner=True)):
            r.copy_(s)                      # what the broadcast delivers on a rank other than 0
        replicate(dist, recv)               # the receiving side's call, on the receiving side's buffers
        torch.cuda.synchronize()
        dst.attach_device_keys(recv[0].data_ptr(), recv[1].data_ptr())
        dst.import_keys(KEY_CLIENT, ctx.export_keys(KEY_CLIENT))
        print("images attached on the second context", flush=True)
        vals = list(range(100, 180))
        blocks = np.concatenate([np.frombuffer(ctx.encrypt(v, 8)[16:], dtype=np.uint64).reshape(4, 2049) for v in vals])
        lut = np.stack([ctx.make_lut(lambda x: (5 * x + 3) % 16)])
        a, b = ctx.pbs_batch(blocks, lut), dst.pbs_batch(blocks, lut)
        assert np.array_equal(a, b), "attached images bootstrap differently"
        assert dst.decrypt_blocks(b[:8]) == [(5 * ((v >> (2 * k)) & 3) + 3) % 16 for v in vals[:2] for k in range(4)]
        assert max_over_ranks(dist, 1.5, dev) == 1.5
        flag = torch.tensor([1], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        assert int(flag.item()) == 1
        dist.barrier()
        print("ok", flush=True)
    finally:
        dst.close()
        ctx.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
