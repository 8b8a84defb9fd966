"""Key replication through torch.distributed on one GPU: two ranks over gloo (host-staged
broadcasts), and one rank over RCCL (the product's own collectives on a communicator of size one).
Both start fresh processes and watch them with a bound: a rank that dies or stalls inside a
collective leaves its peer waiting forever, and a silent wait is what ends a GPU run.
gloo is used on host tensors only: gloo broadcasts directly on device tensors, two ranks on one GPU,
stall inside torch.distributed.broadcast on both ranks (tests/gloo_device_probe.py shows the frame;
neither stream ordering nor the wrapping of library memory is involved).  The product never does
that either -- its device-tensor collectives are RCCL's."""
import pytest

pytestmark = pytest.mark.gpu


def _two_rank_worker(rank, world, port, q):
    """One rank of the bench's key replication, both ranks on the one GPU of the box; gloo with host
    staging stands in for RCCL (two ranks cannot share a device under RCCL), everything else is the
    bench's path: library-owned device memory seen as a torch tensor, one broadcast per key image,
    attach."""
    import faulthandler
    import os
    import sys
    # a rank that stalls names the frame it sits in (all threads) and exits, instead of leaving the
    # parent to guess from a silent queue
    faulthandler.dump_traceback_later(240, exit=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import torch
    import torch.distributed as dist
    from fhe_amd.sharding import max_over_ranks, replicated_context, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class HostStaged:
        """torch.distributed with device tensors staged through the host: gloo moves host memory."""
        ReduceOp = dist.ReduceOp
        broadcast_object_list = staticmethod(dist.broadcast_object_list)
        all_reduce = staticmethod(dist.all_reduce)

        @staticmethod
        def broadcast(t, src=0):
            if not t.is_cuda:
                return dist.broadcast(t, src=src)
            h = t.cpu()
            dist.broadcast(h, src=src)
            if rank != src:
                t.copy_(h)

    ctx, keep = replicated_context(HostStaged, rank, world, dev, bytes(range(32)), 3)
    try:
        vals = list(range(40, 72))
        lo, hi = shard_bounds(len(vals), world, rank)
        blocks = np.concatenate([np.frombuffer(ctx.encrypt(v, 8)[16:], dtype=np.uint64).reshape(4, 2049) for v in vals[lo:hi]])
        out = ctx.pbs_batch(blocks, np.stack([ctx.make_lut(lambda x: (x + 1) % 16)]), None)
        boots_ok = ctx.decrypt_blocks(out) == [(((v >> (2 * k)) & 3) + 1) % 16 for v in vals[lo:hi] for k in range(4)]
        summ = ctx.decrypt(ctx.evaluate("add", 8, ctx.encrypt(vals[lo], 8), ctx.encrypt(vals[hi - 1], 8)))
        t = max_over_ranks(HostStaged, 1.0 + rank)
        q.put((rank, (lo, hi), summ, (vals[lo] + vals[hi - 1]) & 0xFF, boots_ok, t, len(keep)))
        dist.barrier()
    finally:
        ctx.close()
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()


def test_two_ranks_replicate_keys_through_torch_distributed(capsys):
    import queue
    import socket
    import time
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    # wait with a sign of life every 20 s, give up after 10 min
    got, t0 = [], time.time()
    try:
        while len(got) < len(procs):
            try:
                got.append(q.get(timeout=20))
            except queue.Empty:
                assert all(p.is_alive() or p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
                assert time.time() - t0 < 600, "ranks did not report"
                with capsys.disabled():
                    print(f"[two-rank test: waiting for the ranks, {time.time() - t0:.0f} s]", flush=True)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    got.sort()
    (r0, cut0, s0, w0, d0, t0, k0), (r1, cut1, s1, w1, d1, t1, k1) = got
    assert cut0 == (0, 16) and cut1 == (16, 32)
    assert d0 and d1                      # every bootstrapped block decrypts to table(message) on both ranks
    assert s0 == w0 and s1 == w1          # rank 1 decrypts with the replicated client key, computes with the replicated server keys
    assert t0 == t1 == 2.0
    assert k0 == 0 and k1 == 2            # rank 1 keeps the two received key images alive


def test_key_replication_over_rccl_on_one_rank():
    """fhe_amd.sharding.replicated_context as the benchmark calls it, with the nccl backend: library
    memory wrapped as tensors, one RCCL broadcast per key image, the object broadcast of the client
    key, attach on a second context, barrier / max / min reductions on device tensors.  One GPU
    allows one RCCL rank, so the communicator has size one; a fresh process, watched with a bound."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "-u", os.path.join(root, "tests", "rccl_one_rank.py")], env=env, cwd=root,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout, p.stderr[-2000:])
