"""lfhe_pbs_batch_device_noalloc (the bootstrap batch on the library's static scratch array, for
callers that cannot allocate) against lfhe_pbs_batch_device, through the C ABI, bit for bit: two
calls in a row on one GPU, the second smaller than the first (the array keeps the first call's
contents), across the key-switch paths (matrix cores from 128 ciphertexts, VALU below) and the
blind-rotation kernels (256- and 512-thread), for every bootstrapping-key flavour."""
import numpy as np
import pytest
import torch  # noqa: F401 -- at collection time, before any test loads liblfhe: PyTorch ships a HIP runtime of its
#                own, and a process that loads the system's first (through liblfhe) finds no GPU in torch afterwards

pytestmark = pytest.mark.gpu

N = 2048


def _blocks(ctx, count, salt):
    rng = np.random.default_rng(900 + salt)
    vals = rng.integers(0, 256, size=(count + 3) // 4)
    return np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1)
                           for v in vals])[:count]


@pytest.mark.parametrize("params_id", [5, 3, 1, 2, 4, 6],
                         ids=["bsk_prime_grouped3", "bsk_prime_grouped", "bsk_q64", "bsk_prime", "bsk_q64_grouped", "bsk_q64_grouped3"])
def test_no_alloc_entry_point_matches_the_allocating_one(params_id):
    import torch
    from fhe_amd.native import Context, LfheError
    dev = torch.device("cuda", 0)
    ctx = Context(seed=bytes(range(32)), params_id=params_id)
    try:
        ctx.keygen()
        f = [(lambda x, j=j: ((2 * j + 1) * x + j) % 16) for j in range(16)]
        luts = np.stack([ctx.make_lut(g) for g in f])
        d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
        # the first call on a device is the set-up (it asks for the array's address and makes the ordering
        # event): an empty batch does it, so that every call measured below is an ordinary one
        ctx.pbs_batch_device_noalloc(0, 0, 0, None, 0)
        # first call at the largest size the entry point takes, then smaller ones on the same (now dirty) array
        for call, count in enumerate((1024, 300, 100, 1, 700)):
            blocks = _blocks(ctx, count, call)
            idx = ((np.arange(count) * 7 + call) % 16).astype(np.uint32)
            d_in = torch.from_numpy(blocks.view(np.int64)).to(dev)
            d_idx = torch.from_numpy(idx.view(np.int32)).to(dev)
            want = torch.zeros_like(d_in)
            got = torch.zeros_like(d_in)
            torch.cuda.synchronize()
            ctx.pbs_batch_device(count, d_in.data_ptr(), d_luts.data_ptr(), d_idx.data_ptr(), want.data_ptr())
            ctx.sync()
            free_before = torch.cuda.mem_get_info()[0]
            ctx.pbs_batch_device_noalloc(count, d_in.data_ptr(), d_luts.data_ptr(), d_idx.data_ptr(), got.data_ptr())
            ctx.sync()
            assert torch.cuda.mem_get_info()[0] == free_before, "the call changed the amount of free device memory"
            assert torch.equal(got, want), (params_id, count)
            out = got.cpu().numpy().view(np.uint64)
            sample = list(range(0, count, max(1, count // 16)))
            msgs = ctx.decrypt_blocks(blocks[sample])
            assert ctx.decrypt_blocks(out[sample]) == [f[idx[i]](m) % 16 for i, m in zip(sample, msgs)]
        # no table index: every ciphertext takes table 0
        blocks = _blocks(ctx, 40, 77)
        d_in = torch.from_numpy(blocks.view(np.int64)).to(dev)
        want, got = torch.zeros_like(d_in), torch.zeros_like(d_in)
        ctx.pbs_batch_device(40, d_in.data_ptr(), d_luts.data_ptr(), None, want.data_ptr())
        ctx.pbs_batch_device_noalloc(40, d_in.data_ptr(), d_luts.data_ptr(), None, got.data_ptr())
        ctx.sync()
        assert torch.equal(got, want)
        # beyond the stated maximum: a clear error, nothing launched
        with pytest.raises(LfheError) as e:
            ctx.pbs_batch_device_noalloc(1025, d_in.data_ptr(), d_luts.data_ptr(), None, got.data_ptr())
        assert e.value.code == -1 and "1024" in str(e.value)
        ctx.pbs_batch_device_noalloc(0, 0, 0, None, 0)      # empty batch: nothing to do
    finally:
        ctx.close()


def test_no_alloc_entry_point_needs_keys():
    from fhe_amd.native import Context, LfheError
    ctx = Context(seed=bytes(range(32)), params_id=3)
    try:
        with pytest.raises(LfheError) as e:
            ctx.pbs_batch_device_noalloc(4, 1, 1, None, 1)
        assert e.value.code == -2
    finally:
        ctx.close()


def test_two_contexts_back_to_back_share_the_array_in_order():
    """Two contexts (two streams) call the entry point one right after the other with no wait in
    between: the second call's key switch would overwrite the small ciphertexts the first call's
    blind rotation is still reading, were the calls not ordered on the device by the library."""
    import torch
    from fhe_amd.native import Context
    dev = torch.device("cuda", 0)
    a = Context(seed=bytes(range(32)), params_id=3)
    b = Context(seed=bytes(range(1, 33)), params_id=3)
    try:
        a.keygen()
        b.keygen()
        work = []
        for k, ctx in enumerate((a, b)):
            luts = np.stack([ctx.make_lut(lambda x, j=j: (3 * x + j) % 16) for j in range(4)])
            blocks = _blocks(ctx, 1024, 50 + k)
            idx = ((np.arange(1024) + k) % 4).astype(np.uint32)
            d = {"in": torch.from_numpy(blocks.view(np.int64)).to(dev), "luts": torch.from_numpy(luts.view(np.int64)).to(dev),
                 "idx": torch.from_numpy(idx.view(np.int32)).to(dev)}
            d["want"], d["got"] = torch.zeros_like(d["in"]), torch.zeros_like(d["in"])
            ctx.pbs_batch_device(1024, d["in"].data_ptr(), d["luts"].data_ptr(), d["idx"].data_ptr(), d["want"].data_ptr())
            ctx.sync()
            work.append(d)
        torch.cuda.synchronize()
        for rounds in range(3):          # a, b, a, b, a, b with nothing in between
            for ctx, d in zip((a, b), work):
                ctx.pbs_batch_device_noalloc(1024, d["in"].data_ptr(), d["luts"].data_ptr(), d["idx"].data_ptr(),
                                             d["got"].data_ptr())
        a.sync()
        b.sync()
        for d in work:
            assert torch.equal(d["got"], d["want"])
    finally:
        a.close()
        b.close()
