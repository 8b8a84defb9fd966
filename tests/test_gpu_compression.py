"""Compression for storage: packing key switch into GLWE ciphertexts modulo 2^12 and decompression
by sample extraction + bootstrap, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[1, 3, 4, 5, 6], ids=["bsk_q64", "bsk_prime_grouped", "bsk_q64_grouped", "bsk_prime_grouped3", "bsk_q64_grouped3"])
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


@pytest.fixture(scope="module")
def oracle_keys():
    """Client keys, key-switching key, compression secret and packing key made by the oracle, once."""
    from oracle import compress as OC
    from oracle.oracle import Oracle, Rng, seed_bytes
    orc = Oracle()
    lwe_sk, glwe_sk, bsk, ksk = orc.gen_keys(seed_bytes(40))
    comp_sk = OC.gen_comp_sk(Rng(seed_bytes(41), 1))
    pksk = OC.gen_packing_ksk(glwe_sk, comp_sk, Rng(seed_bytes(41), 2))
    return orc, lwe_sk, glwe_sk, bsk, ksk, comp_sk, pksk


@pytest.mark.parametrize("flavour,params_id", [("q64", 1), ("prime", 2), ("grouped", 3), ("q64_grouped", 4), ("grouped3", 5), ("q64_grouped3", 6)])
def test_compression_bit_exact_against_the_oracle(oracle_keys, flavour, params_id):
    """Same keys on both sides (made by the oracle, imported raw): the compressed list is the
    oracle's byte for byte, and every decompressed block is the oracle's bootstrap of the extracted
    block word for word."""
    import struct
    from fhe_amd.native import Context
    from oracle import compress as OC
    from oracle.oracle import seed_bytes
    orc, lwe_sk, glwe_sk, bsk, ksk, comp_sk, pksk = oracle_keys
    dec = OC.Decompressor(flavour, comp_sk, glwe_sk, seed_bytes(42))
    if flavour == "q64":
        main_bsk = bsk
    elif flavour in ("q64_grouped", "q64_grouped3"):
        main_bsk = orc.gen_bsk_grouped(lwe_sk, glwe_sk, seed_bytes(40), 3 if flavour == "q64_grouped3" else 2)
    elif flavour == "prime":
        main_bsk = orc.gen_bsk_qp(lwe_sk, glwe_sk, seed_bytes(40))
    else:
        main_bsk = orc.gen_bsk_qp_grouped(lwe_sk, glwe_sk, seed_bytes(40), 3 if flavour == "grouped3" else 2)
    ctx = Context(seed=seed_bytes(43), params_id=params_id)
    try:
        ctx.import_server_keys_raw(main_bsk, ksk, None)
        ctx.import_client_keys_raw(lwe_sk, glwe_sk)
        ctx.import_compression_keys_raw(pksk, dec.bsk)
        widths, vals = [8, 64, 1, 16], [0xA7, 0x0123456789ABCDEF, 1, 0xBEEF]          # 4 + 32 + 1 + 8 = 45 blocks
        cts = [ctx.encrypt(v, w) for v, w in zip(vals, widths)]
        blocks = np.concatenate([np.frombuffer(c[16:], dtype=np.uint64).reshape(-1, 2049) for c in cts])
        blob = ctx.compress(cts)
        glwe = OC.pack(blocks, pksk, 4)
        want = struct.pack("<IHHIIII", 0x5A43464C, 1, 12, len(cts), len(blocks), 4, 256)
        for w in widths:
            want += struct.pack("<HH", w, 1 if w == 1 else w // 2)
        want += OC.pack_bits(glwe[0][:1024 + len(blocks)], 12)
        assert blob == want
        back = ctx.decompress(blob)
        assert [ctx.decrypt(c) for c in back] == vals
        small = OC.extract(glwe, len(blocks))
        lut = dec.table()
        got = np.concatenate([np.frombuffer(c[16:], dtype=np.uint64).reshape(-1, 2049) for c in back])
        for j in (0, 1, 17, 36, 44):
            assert np.array_equal(got[j], dec.bootstrap(small[j], lut)), j
    finally:
        ctx.close()
