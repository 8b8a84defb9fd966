"""What a rank other than 0 does in a multi-GPU run, on one GPU: it receives the two device key
images (here: a device-to-device copy stands in for the RCCL broadcast), attaches them and must
then bootstrap exactly like the context that generated the keys."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("params_id", [1, 2, 3], ids=["bsk_q64", "bsk_prime", "bsk_prime_grouped"])
def test_attached_key_images_bootstrap_like_the_originals(params_id):
    import ctypes as C
    from fhe_amd.native import Context, KEY_CLIENT
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    D2D = 3
    src = Context(seed=bytes(range(32)), params_id=params_id)
    dst = Context(seed=bytes(range(32)), params_id=params_id)
    tb, tk = C.c_void_p(), C.c_void_p()
    try:
        src.keygen()
        bp, bsk_bytes, kp, ksk_bytes = src.device_keys()
        assert (bsk_bytes, ksk_bytes) == dst.device_keys_size()
        assert hip.hipMalloc(C.byref(tb), bsk_bytes) == 0 and hip.hipMalloc(C.byref(tk), ksk_bytes) == 0
        assert hip.hipMemcpy(tb, C.c_void_p(bp), bsk_bytes, D2D) == 0      # synchronous
        assert hip.hipMemcpy(tk, C.c_void_p(kp), ksk_bytes, D2D) == 0
        dst.attach_device_keys(tb.value, tk.value)
        dst.import_keys(KEY_CLIENT, src.export_keys(KEY_CLIENT))
        rng = np.random.default_rng(3)
        vals = [int(v) for v in rng.integers(0, 256, size=80)]           # 320 blocks: the matrix-core key switch
        blocks = np.concatenate([np.frombuffer(src.encrypt(v, 8)[16:], dtype=np.uint64).reshape(4, 2049) for v in vals])
        luts = np.stack([src.make_lut(lambda x: (3 * x + 1) % 16), src.make_lut(lambda x: x)])
        idx = (np.arange(len(blocks)) % 2).astype(np.uint32)
        a, b = src.pbs_batch(blocks, luts, idx), dst.pbs_batch(blocks, luts, idx)
        assert np.array_equal(a, b)
        # and the operator level on the attached context
        x, y = dst.encrypt(200, 8), dst.encrypt(77, 8)
        assert dst.decrypt(dst.evaluate("add", 8, x, y)) == (200 + 77) & 0xFF
        assert src.decrypt(dst.evaluate("mul", 8, x, y)) == (200 * 77) & 0xFF
    finally:
        dst.close()
        src.close()
        hip.hipFree(tb)
        hip.hipFree(tk)


def _two_rank_worker(rank, world, port, q):
    """One rank of the bench's key replication, both ranks on the one GPU of the box; gloo stands
    in for RCCL (two ranks cannot share a device under RCCL), everything else is the bench's path:
    library-owned device memory seen as a torch tensor, one broadcast per key image, attach."""
    import os
    import sys
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
    ctx, keep = replicated_context(dist, rank, world, dev, bytes(range(32)), 3)
    try:
        vals = list(range(40, 72))
        lo, hi = shard_bounds(len(vals), world, rank)
        blocks = np.concatenate([np.frombuffer(ctx.encrypt(v, 8)[16:], dtype=np.uint64).reshape(4, 2049) for v in vals[lo:hi]])
        out = ctx.pbs_batch(blocks, np.stack([ctx.make_lut(lambda x: (x + 1) % 16)]), None)
        boots_ok = ctx.decrypt_blocks(out) == [(((v >> (2 * k)) & 3) + 1) % 16 for v in vals[lo:hi] for k in range(4)]
        summ = ctx.decrypt(ctx.evaluate("add", 8, ctx.encrypt(vals[lo], 8), ctx.encrypt(vals[hi - 1], 8)))
        t = max_over_ranks(dist, 1.0 + rank, device=dev)
        q.put((rank, (lo, hi), summ, (vals[lo] + vals[hi - 1]) & 0xFF, boots_ok, t, len(keep)))
        dist.barrier()
    finally:
        ctx.close()
        dist.destroy_process_group()


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
    # the two ranks are the first processes of the run to import torch, which takes minutes on a box
    # that has not paged the image in yet: wait with a sign of life every 20 s, give up after 10 min
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
