"""Compression for storage: packing key switch into GLWE ciphertexts modulo 2^12 and decompression
by sample extraction + bootstrap, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[1, 3], ids=["bsk_q64", "bsk_prime_grouped"])
def ctx(request):
    from fhe_amd.native import Context
    c = Context(seed=bytes(range(32)), params_id=request.param)
    c.keygen()
    c.compression_keygen()
    yield c
    c.close()


def test_round_trip_values_sizes_and_reuse(ctx):
    import random
    rnd = random.Random(5)
    widths = [8, 64, 1, 256, 16, 32, 64, 128]                 # 4+32+1+128+8+16+32+64 = 285 blocks: two GLWEs
    vals = [rnd.randrange(1 << w) for w in widths]
    cts = [ctx.encrypt(v, w) for v, w in zip(vals, widths)]
    blob = ctx.compress(cts)
    plain = sum(len(c) for c in cts)
    assert len(blob) < plain // 1000                           # 16 392 bytes per block -> about 6 bytes
    assert len(blob) == 24 + 4 * len(cts) + (1024 + 256) * 12 // 8 + ((1024 + 29) * 12 + 7) // 8
    back = ctx.decompress(blob)
    assert [len(c) for c in back] == [len(c) for c in cts]
    assert [ctx.decrypt(c) for c in back] == vals
    # results of operators compress too, and decompressed ciphertexts are operands like any other
    s = ctx.evaluate("add", 64, cts[1], cts[6])
    m = ctx.evaluate("mul", 16, cts[4], cts[4])
    b2 = ctx.decompress(ctx.compress([s, m]))
    assert ctx.decrypt(b2[0]) == (vals[1] + vals[6]) & ((1 << 64) - 1)
    assert ctx.decrypt(b2[1]) == (vals[4] * vals[4]) & 0xFFFF
    assert ctx.decrypt(ctx.evaluate("sub", 64, b2[0], back[6])) == vals[1]
    # compressing what was decompressed gives the same values again
    assert [ctx.decrypt(c) for c in ctx.decompress(ctx.compress(back))] == vals


def test_many_blocks_and_error_paths(ctx):
    from fhe_amd.native import LfheError
    rng = np.random.default_rng(8)
    vals = [int(v) for v in rng.integers(0, 1 << 62, size=40)]                # 1280 blocks: five full GLWEs
    cts = [ctx.encrypt(v, 64) for v in vals]
    blob = ctx.compress(cts)
    assert [ctx.decrypt(c) for c in ctx.decompress(blob)] == vals
    with pytest.raises(LfheError):
        ctx.decompress(blob[:-1])
    with pytest.raises(LfheError):
        ctx.decompress(b"\\x00" * 64)
    with pytest.raises(LfheError):
        ctx.compress([cts[0][:-8]])
