"""HIP path vs CPU oracle through the C ABI, bit for bit (integer work: exact equality).

Ciphertext-level parity is GPU-vs-own-oracle: the reference tree holds no engine and no stored
ciphertexts for this path, so these bits are unpinned against the reference; what the reference's
fixtures do pin (the negacyclic product, decrypted operator results) is covered in
test_oracle_golden.py and the operator tests."""
import numpy as np
import pytest

from oracle.oracle import Oracle, Params as OParams, Rng, seed_bytes, lib as olib

pytestmark = pytest.mark.gpu

N = 2048
M64 = (1 << 64) - 1


def _mk(n, ks_base_log=4, ks_level=4, seed=1, ms_zeros=0, grouped=False, factored=True, bits=2, shape="split", env=None):
    """Oracle and library context over the same keys.  ms_zeros > 0 adds that many encryptions of
    zero and switches the modulus-switch noise reduction on in both.  grouped: the bootstrapping key
    with 2^bits GGSWs over 2^64 per group of `bits` key bits.  shape: which kernel small launches of a grouped
    key take -- "split" (the default: up to one ciphertext per two CUs on two CUs each, up to one per CU on 512
    threads), "wide" (512 threads, never two CUs) or "narrow" (the 256-thread kernel at every size).  env: further
    environment variables for the time the context is created."""
    from fhe_amd.native import Context, Params, default_params
    op = OParams(lwe_dim=n, ks_base_log=ks_base_log, ks_level=ks_level)
    orc = Oracle(op)
    keys = orc.gen_keys(seed_bytes(seed))
    p = Params(**{k: getattr(op, k) for k in op.__dataclass_fields__})
    if grouped:
        keys = (keys[0], keys[1], orc.gen_bsk_grouped(keys[0], keys[1], seed_bytes(seed), bits), keys[3])
        p.bsk_grouping = bits
    zeros = None
    if ms_zeros:
        d = default_params()
        p.ms_zeros_count, p.ms_bound_log2 = ms_zeros, d.ms_bound_log2
        p.ms_r_sigma_factor, p.ms_input_variance = d.ms_r_sigma_factor, d.ms_input_variance
        zeros = orc.gen_ms_zeros(keys[0], seed_bytes(seed), ms_zeros)
        orc.set_ms(zeros)
        orc.zeros = zeros
    import os
    setup = dict(env or {})
    if shape in ("wide", "narrow"):
        setup["LFHE_G64_SPLIT_BELOW"] = "0"
    if shape == "narrow":
        setup["LFHE_WIDE_BELOW"] = "0"
    for k in ("LFHE_WIDE_BELOW", "LFHE_G64_SPLIT_BELOW", "LFHE_G64_SPLIT_POLL"):
        os.environ.pop(k, None)
    os.environ.update(setup)
    try:
        ctx = Context(seed=seed_bytes(100 + seed), params=p)
    finally:
        for k in setup:
            os.environ.pop(k, None)
    if not factored:                      # keep the grouped key's image as generated (the fall-back path)
        os.environ["LFHE_BSK_FACTORED"] = "0"
    try:
        ctx.import_server_keys_raw(keys[2], keys[3], zeros)
    finally:
        os.environ.pop("LFHE_BSK_FACTORED", None)
    ctx.import_client_keys_raw(keys[0], keys[1])
    return orc, keys, ctx


@pytest.fixture(scope="module")
def small():
    orc, keys, ctx = _mk(12)
    yield orc, keys, ctx
    ctx.close()


@pytest.fixture(scope="module")
def full():
    orc, keys, ctx = _mk(918, seed=2, ms_zeros=1449)   # the reference's parameter set, noise reduction on
    yield orc, keys, ctx, orc.prepare_bsk(keys[2])
    ctx.close()


def test_ntt_forward_matches_definition(small):
    orc, _, ctx = small
    primes, psis = ctx.ntt_info()
    r = Rng(seed_bytes(20))
    polys = np.stack([r.u64s(N), np.full(N, 1 << 63, dtype=np.uint64), np.arange(N, dtype=np.uint64)])
    polys[2][::3] = M64
    got = ctx.ntt_forward(polys)
    for c in range(polys.shape[0]):
        cent = polys[c].astype(np.int64)
        for k in range(2):
            p = primes[k]
            res = np.array([int(v) % p for v in cent], dtype=np.uint64)
            want = Oracle.ntt_definition(p, psis[k], res)
            assert np.array_equal(got[c, k], want), (c, k)


def test_ntt_inverse_round_trip_and_product(small):
    orc, _, ctx = small
    primes, _ = ctx.ntt_info()
    r = Rng(seed_bytes(21))
    a = r.u64s(N)
    small_poly = np.array([(r.u64() % (1 << 23)) - (1 << 22) for _ in range(N)], dtype=np.int64)
    # round trip of a polynomial whose centred coefficients are below P/2
    assert np.array_equal(ctx.ntt_inverse(ctx.ntt_forward(a[None]))[0], a)
    # exact negacyclic product through the device transforms equals the oracle's product
    A = ctx.ntt_forward(a[None])[0]
    S = ctx.ntt_forward(small_poly.astype(np.uint64)[None])[0]
    prod = np.stack([np.array([int(x) * int(y) % primes[k] for x, y in zip(A[k], S[k])], dtype=np.uint64)
                     for k in range(2)])
    assert np.array_equal(ctx.ntt_inverse(prod[None])[0], orc.mul_schoolbook(small_poly, a))


def test_external_product_bit_exact(small):
    orc, keys, ctx = small
    bsk = keys[2]
    W = orc.ggsw_words()
    r = Rng(seed_bytes(22))
    glwe = np.stack([r.u64s(2 * N) for _ in range(6)])
    glwe[4][:] = 1 << 63                     # every digit +2^22
    glwe[5][::2] = (1 << 63) + (1 << 41)     # digits alternate -2^22+1 / random
    idx = np.array([0, 3, 7, 11, 5, 2], dtype=np.uint32)
    got = ctx.external_product(idx, glwe)
    for c in range(len(idx)):
        want = orc.external_product(bsk[idx[c] * W:(idx[c] + 1) * W], glwe[c])
        assert np.array_equal(got[c].reshape(-1), want), c


@pytest.mark.parametrize("base_log,level", [(4, 4), (3, 3), (7, 2)])
def test_keyswitch_bit_exact(base_log, level):
    orc, keys, ctx = _mk(40, base_log, level, seed=3)
    try:
        r = Rng(seed_bytes(23))
        cts = np.stack([r.u64s(N + 1) for _ in range(19)])   # ragged: not a multiple of the tile
        cts[1][:] = 0
        cts[2][:] = M64
        cts[3][:] = (1 << 63) - (1 << 47)                    # decomposition carries ripple to the top
        got = ctx.keyswitch(cts)
        for c in range(cts.shape[0]):
            assert np.array_equal(got[c], orc.keyswitch(keys[3], cts[c])), c
    finally:
        ctx.close()


@pytest.mark.parametrize("n,base_log,level", [(40, 4, 4), (33, 3, 3), (918, 4, 4), (100, 7, 2)])
def test_keyswitch_on_the_matrix_cores_bit_exact(n, base_log, level):
    """Large batches key-switch as an int8 matrix product (v_mfma_i32_32x32x32_i8) over byte limbs of
    the key; forced here for a small ragged batch: every word equals the oracle's, including inputs
    whose digits are all extreme (largest partial sums) and the VALU kernel's output."""
    import os
    os.environ["LFHE_KS_MFMA_MIN"] = "1"
    try:
        orc, keys, ctx = _mk(n, base_log, level, seed=8)
    finally:
        del os.environ["LFHE_KS_MFMA_MIN"]
    orc2, keys2, ctx_valu = None, None, None
    try:
        r = Rng(seed_bytes(31))
        cts = np.stack([r.u64s(N + 1) for _ in range(131)])  # ragged: not a multiple of the 128-row tile
        cts[1][:] = 0
        cts[2][:] = M64
        cts[3][:] = (1 << 63) - (1 << 47)                    # decomposition carries ripple to the top
        half = 1 << (base_log - 1)
        word = 0
        for lv in range(level):                              # every digit -B/2: the most negative sums
            word |= half << (64 - base_log * (lv + 1))
        cts[4][:] = word
        cts[5][:] = (word - (1 << (64 - base_log * level))) & M64   # every digit B/2 - 1 (level > 1: carries aside)
        got = ctx.keyswitch(cts)
        for c in list(range(8)) + [64, 127, 128, 130]:
            assert np.array_equal(got[c], orc.keyswitch(keys[3], cts[c])), c
        os.environ["LFHE_KS_MFMA_MIN"] = "0"
        try:
            orc2, keys2, ctx_valu = _mk(n, base_log, level, seed=8)
        finally:
            del os.environ["LFHE_KS_MFMA_MIN"]
        assert np.array_equal(got, ctx_valu.keyswitch(cts))
    finally:
        ctx.close()
        if ctx_valu is not None:
            ctx_valu.close()


def test_blind_rotate_and_pbs_bit_exact_small_n(small):
    orc, keys, ctx = small
    lwe_sk, glwe_sk, bsk, ksk = keys
    pb = orc.prepare_bsk(bsk)
    r = Rng(seed_bytes(24))
    f = [lambda x: x, lambda x: (3 * x + 1) % 16, lambda x: 15 - x]
    luts_o = [orc.make_lut(g) for g in f]
    luts_g = np.stack([ctx.make_lut(g) for g in f])
    for lo, lg in zip(luts_o, luts_g):
        assert np.array_equal(lo[N:], lg) and not lo[:N].any()
    smalls = np.stack([r.u64s(orc.n + 1) for _ in range(5)])
    smalls[4][:] = 0                                          # every rotation zero: accumulator untouched
    idx = np.array([0, 1, 2, 1, 0], dtype=np.uint32)
    got = ctx.blind_rotate(smalls, luts_g, idx)
    for c in range(5):
        want = orc.blind_rotate(pb, smalls[c], luts_o[idx[c]])
        assert np.array_equal(got[c].reshape(-1), want), c
    cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in range(16)])
    idx = np.array([m % 3 for m in range(16)], dtype=np.uint32)
    out = ctx.pbs_batch(cts, luts_g, idx)
    for m in range(16):
        assert np.array_equal(out[m], orc.pbs(ksk, pb, cts[m], luts_o[idx[m]])), m


def test_pbs_full_parameters(full):
    """The benchmark's parameter set (n = 918): every output decrypts to f(m); a subsample is
    compared bit for bit with the oracle."""
    orc, keys, ctx, pb = full
    lwe_sk, glwe_sk, bsk, ksk = keys
    r = Rng(seed_bytes(25))
    f = [lambda x: x, lambda x: (x * x + 3) % 16]
    luts_o = [orc.make_lut(g) for g in f]
    luts_g = np.stack([ctx.make_lut(g) for g in f])
    B = 96
    msgs = [int(r.u64() % 16) for _ in range(B)]
    cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in msgs])
    idx = np.array([i % 2 for i in range(B)], dtype=np.uint32)
    out = ctx.pbs_batch(cts, luts_g, idx)
    for i in range(B):
        assert orc.decrypt_block(glwe_sk, out[i]) == f[idx[i]](msgs[i]) % 16, i
    assert np.array_equal(out, orc.pbs_batch(ksk, pb, cts, np.stack(luts_o), idx, nthreads=16))   # all 96, bit for bit
    # idempotence of a second bootstrap by the identity: value preserved
    out2 = ctx.pbs_batch(out, luts_g[:1])
    for i in range(0, B, 7):
        assert orc.decrypt_block(glwe_sk, out2[i]) == f[idx[i]](msgs[i]) % 16


def test_ms_noise_reduction_bit_exact(full):
    """The candidate search between key switch and modulus switch: same choice and same words as
    the oracle; the estimate ends within the bound; the plaintext is untouched."""
    orc, keys, ctx, _ = full
    lwe_sk, glwe_sk, bsk, ksk = keys
    r = Rng(seed_bytes(26))
    cts = np.stack([orc.encrypt_block(glwe_sk, i % 16, r) for i in range(70)])
    smalls = ctx.keyswitch(cts)
    smalls[3] = 0                       # rounds exactly: estimate is the input variance alone, nothing added
    got, chosen = ctx.ms_noise_reduce(smalls)
    picks = []
    for i in range(len(smalls)):
        want, pick = orc.ms_noise_reduce(smalls[i], orc.zeros)
        picks.append(pick)
        assert pick == chosen[i] and np.array_equal(got[i], want), i
        assert orc.ms_estimate(got[i]) <= 1 << 32
        if i != 3:
            assert orc.decode(orc.lwe_phase(lwe_sk, got[i])) == i % 16
    assert chosen[3] == -1 and max(picks) > 3 and len(set(picks)) > 10   # the search really runs
    # batches of more than 8192 search eight candidates per round instead of sixteen: same choices, same words
    reps = 120
    got_l, chosen_l = ctx.ms_noise_reduce(np.tile(smalls, (reps, 1)))
    assert np.array_equal(chosen_l, np.tile(chosen, reps)) and np.array_equal(got_l, np.tile(got, (reps, 1)))
    # without encryptions of zero in the key set the stage refuses to run
    o2, k2, c2 = _mk(12)
    try:
        from fhe_amd.native import LfheError
        with pytest.raises(LfheError):
            c2.ms_noise_reduce(np.zeros((1, 13), dtype=np.uint64))
    finally:
        c2.close()


# ---- bootstrap over the prime ring modulus (bootstrapping key and accumulator in Z_q') ------------
def _mk_prime(n, seed, ms_zeros=0, wide=True, grouped=False, bits=2):
    """wide=False forces the 256-thread blind rotation for small launches too (by default launches
    of up to two ciphertexts per CU take the 512-thread kernel).  grouped: the key with four GGSWs
    per pair of key bits, blind rotation in (n + 1) / 2 steps."""
    import os
    from fhe_amd.native import Context, Params, default_params, BSK_MODULUS_PRIME
    os.environ.pop("LFHE_WIDE_BELOW", None)
    if not wide:
        os.environ["LFHE_WIDE_BELOW"] = "0"
    op = OParams(lwe_dim=n)
    orc = Oracle(op)
    lwe_sk, glwe_sk, _, ksk = orc.gen_keys(seed_bytes(seed))
    bskq = (orc.gen_bsk_qp_grouped(lwe_sk, glwe_sk, seed_bytes(seed), bits) if grouped
            else orc.gen_bsk_qp(lwe_sk, glwe_sk, seed_bytes(seed)))
    p = Params(**{k: getattr(op, k) for k in op.__dataclass_fields__})
    p.bsk_modulus = BSK_MODULUS_PRIME
    p.bsk_grouping = bits if grouped else 1
    zeros = None
    if ms_zeros:
        d = default_params()
        p.ms_zeros_count, p.ms_bound_log2 = ms_zeros, d.ms_bound_log2
        p.ms_r_sigma_factor, p.ms_input_variance = d.ms_r_sigma_factor, d.ms_input_variance
        zeros = orc.gen_ms_zeros(lwe_sk, seed_bytes(seed), ms_zeros)
        orc.set_ms(zeros)
    orc.test_zeros = zeros      # for tests that load the same keys into a second context
    ctx = Context(seed=seed_bytes(100 + seed), params=p)
    os.environ.pop("LFHE_WIDE_BELOW", None)
    ctx.import_server_keys_raw(bskq, ksk, zeros)
    ctx.import_client_keys_raw(lwe_sk, glwe_sk)
    return orc, (lwe_sk, glwe_sk, bskq, ksk), ctx


@pytest.mark.parametrize("wide", [True, False], ids=["512-thread", "256-thread"])
def test_prime_modulus_external_product_and_blind_rotation_bit_exact(wide):
    orc, keys, ctx = _mk_prime(12, seed=4, wide=wide)
    try:
        lwe_sk, glwe_sk, bskq, ksk = keys
        q = orc.qp
        assert orc.qp_gadget() == 83886080 and orc.qp_noise_bound() == 5
        W = orc.ggsw_words()
        r = Rng(seed_bytes(27))
        glwe = np.stack([r.u64s(2 * N) % np.uint64(q) for _ in range(5)])
        glwe[3][:] = q // 2                        # largest positive centred value: digit +2^22
        glwe[4][:] = q // 2 + 1                    # most negative
        glwe[2][::2] = 83886080 // 2               # ties of the digit rounding
        idx = np.array([0, 3, 7, 11, 5], dtype=np.uint32)
        got = ctx.external_product(idx, glwe)
        for c in range(len(idx)):
            want = orc.external_product_qp(bskq[idx[c] * W:(idx[c] + 1) * W], glwe[c])
            assert np.array_equal(got[c].reshape(-1), want), c
        pb = orc.prepare_bsk_qp(bskq)
        f = [lambda x: x, lambda x: (3 * x + 1) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(4)])
        smalls[3][:] = 0
        li = np.array([0, 1, 1, 0], dtype=np.uint32)
        got = ctx.blind_rotate(smalls, luts_g, li)
        for c in range(4):
            assert np.array_equal(got[c].reshape(-1), orc.blind_rotate_qp(pb, smalls[c], luts_o[li[c]])), c
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in range(16)])
        li = np.array([m % 2 for m in range(16)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, li)
        for m in range(16):
            assert np.array_equal(out[m], orc.pbs_qp(ksk, pb, cts[m], luts_o[li[m]])), m
            assert orc.decrypt_block(glwe_sk, out[m]) == f[li[m]](m) % 16
    finally:
        ctx.close()


@pytest.mark.parametrize("wide", [True, False], ids=["512-thread", "256-thread"])
def test_prime_modulus_pbs_full_parameters(wide):
    """n = 918 with the noise reduction on: outputs decrypt to f(m), a subsample is bit-identical
    to the oracle, the output noise stays far below half a message step."""
    orc, keys, ctx = _mk_prime(918, seed=5, ms_zeros=1449, wide=wide)
    try:
        lwe_sk, glwe_sk, bskq, ksk = keys
        pb = orc.prepare_bsk_qp(bskq)
        r = Rng(seed_bytes(28))
        f = [lambda x: x, lambda x: (x * x + 3) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        B = 64
        msgs = [int(r.u64() % 16) for _ in range(B)]
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in msgs])
        idx = np.array([i % 2 for i in range(B)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, idx)
        worst = 0
        for i in range(B):
            want = f[idx[i]](msgs[i]) % 16
            assert orc.decrypt_block(glwe_sk, out[i]) == want, i
            e = (orc.lwe_phase(glwe_sk, out[i]) - orc.encode(want)) % (1 << 64)
            worst = max(worst, min(e, (1 << 64) - e))
        assert worst < 1 << 53            # half a message step is 2^58
        assert np.array_equal(out, orc.pbs_batch_qp(ksk, pb, cts, np.stack(luts_o), idx, nthreads=16))   # all 64, bit for bit
    finally:
        ctx.close()


@pytest.mark.parametrize("wide", [True, False], ids=["512-thread", "256-thread"])
@pytest.mark.parametrize("n", [12, 13])
def test_grouped_key_blind_rotation_bit_exact(n, wide):
    """Two key bits per step: accumulators and bootstrapped ciphertexts equal the oracle's bit for
    bit, for an even and an odd key length, including all-zero masks (every step skipped) and masks
    whose pairs cancel (a + a' = 2N)."""
    orc, keys, ctx = _mk_prime(n, seed=6, grouped=True, wide=wide)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_qp_grouped(bskg)
        r = Rng(seed_bytes(29))
        f = [lambda x: x, lambda x: (3 * x + 1) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(6)])
        smalls[3][:] = 0
        smalls[4][0::2] = np.uint64(1 << 63)          # a = N for every first element of a pair
        smalls[4][1:n:2] = np.uint64(1 << 63)         # ... and a' = N: a + a' = 2N
        smalls[5][0:n:2] = 0                          # a = 0, a' random
        li = np.array([0, 1, 1, 0, 1, 0], dtype=np.uint32)
        got = ctx.blind_rotate(smalls, luts_g, li)
        for c in range(6):
            assert np.array_equal(got[c].reshape(-1), orc.blind_rotate_qp(pb, smalls[c], luts_o[li[c]])), c
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in range(16)])
        li = np.array([m % 2 for m in range(16)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, li)
        for m in range(16):
            assert np.array_equal(out[m], orc.pbs_qp(ksk, pb, cts[m], luts_o[li[m]])), m
            assert orc.decrypt_block(glwe_sk, out[m]) == f[li[m]](m) % 16
    finally:
        ctx.close()


@pytest.mark.parametrize("wide", [True, False], ids=["512-thread", "256-thread"])
@pytest.mark.parametrize("n", [12, 13, 14])
def test_three_key_bits_per_step_bit_exact(n, wide):
    """Eight GGSWs per triple of key bits, a third fewer steps: accumulators and bootstrapped
    ciphertexts equal the oracle's bit for bit, for key lengths that leave a full, a one-bit and a
    two-bit last group, for the 512-thread kernel and for one and two ciphertexts per 256-thread
    workgroup (odd batches), all-zero masks and masks whose exponents cancel."""
    orc, keys, ctx = _mk_prime(n, seed=6, grouped=True, bits=3, wide=wide)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        assert len(bskg) == 8 * ((n + 2) // 3) * orc.ggsw_words()
        pb = orc.prepare_bsk_qp_grouped(bskg, 3)
        r = Rng(seed_bytes(29))
        f = [lambda x: x, lambda x: (3 * x + 1) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(7)])
        smalls[3][:] = 0
        smalls[4][:n] = np.uint64(1 << 63)            # every exponent N: pairs of them cancel, triples give X^N
        smalls[5][0:n:3] = 0                          # a = 0, a', a'' random
        li = np.array([0, 1, 1, 0, 1, 0, 1], dtype=np.uint32)
        got = ctx.blind_rotate(smalls, luts_g, li)    # seven: 512 threads each, or two per workgroup with an odd one left
        for c in range(7):
            assert np.array_equal(got[c].reshape(-1), orc.blind_rotate_qp(pb, smalls[c], luts_o[li[c]])), c
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in range(16)] * 37)[:577]     # one per workgroup (2 .. 3.5 per CU)
        li = np.array([m % 2 for m in range(577)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, li)
        for m in (0, 1, 7, 15, 16, 300, 575, 576):
            assert np.array_equal(out[m], orc.pbs_qp(ksk, pb, cts[m], luts_o[li[m]])), m
        assert ctx.decrypt_blocks(out) == [f[li[i]](i % 16) % 16 for i in range(577)]
        assert np.array_equal(ctx.pbs_batch(cts[:9], luts_g, li[:9]), out[:9])
        assert np.array_equal(ctx.pbs_batch(cts[:301], luts_g, li[:301]), out[:301])       # two per workgroup, odd
    finally:
        ctx.close()


def test_three_key_bits_per_step_full_parameters():
    """n = 918 = 306 triples, noise reduction on: every output decrypts to f(m) with noise far below
    half a message step, every one of the 600 bit-identical to the oracle."""
    orc, keys, ctx = _mk_prime(918, seed=11, ms_zeros=1449, grouped=True, bits=3)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_qp_grouped(bskg, 3)
        r = Rng(seed_bytes(36))
        f = [lambda x: x, lambda x: (x * x + 3) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        B = 600
        msgs = [int(r.u64() % 16) for _ in range(B)]
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in msgs])
        idx = np.array([i % 2 for i in range(B)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, idx)
        worst = 0
        for i in range(B):
            want = f[idx[i]](msgs[i]) % 16
            assert orc.decrypt_block(glwe_sk, out[i]) == want, i
            e = (orc.lwe_phase(glwe_sk, out[i]) - orc.encode(want)) % (1 << 64)
            worst = max(worst, min(e, (1 << 64) - e))
        assert worst < 1 << 53
        assert np.array_equal(out, orc.pbs_batch_qp(ksk, pb, cts, np.stack(luts_o), idx, nthreads=16))
        assert np.array_equal(ctx.pbs_batch(cts[:8], luts_g, idx[:8]), out[:8])
    finally:
        ctx.close()


def test_grouped_key_with_a_non_invertible_word_keeps_the_plain_image():
    """The factored image divides by the words of every fourth GGSW; a key in which one of them is zero
    (here: a whole GGSW of zeros, which no key generation produces) is detected at load, held as
    generated and bootstrapped by the plain instantiation of the kernels: same bits as the oracle."""
    orc, keys, ctx = _mk_prime(12, seed=11, grouped=True)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        assert ctx.bsk_factored()
        W = orc.ggsw_words()
        broken = bskg.copy()
        broken[7 * W:8 * W] = 0                      # K3 of the second pair of key bits
        ctx.import_server_keys_raw(broken, ksk, None)
        assert not ctx.bsk_factored()
        pb = orc.prepare_bsk_qp_grouped(broken)
        r = Rng(seed_bytes(35))
        lut_o, lut_g = orc.make_lut(lambda x: (5 * x + 2) % 16), ctx.make_lut(lambda x: (5 * x + 2) % 16)
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(5)])
        got = ctx.blind_rotate(smalls, lut_g[None], None)
        for c in range(5):
            assert np.array_equal(got[c].reshape(-1), orc.blind_rotate_qp(pb, smalls[c], lut_o)), c
        ctx.import_server_keys_raw(bskg, ksk, None)
        assert ctx.bsk_factored()
    finally:
        ctx.close()


@pytest.mark.parametrize("modulus", ["prime", "q64"])
def test_three_bit_key_with_a_non_invertible_word_is_refused(modulus):
    """Three key bits per step exist only with the factored image: a key in which a word of the fourth or
    eighth GGSW of a group has no inverse (here: a whole GGSW of zeros) is refused at load with
    LFHE_ERR_UNSUPPORTED, and the context goes on working with a good key afterwards."""
    from fhe_amd.native import LfheError
    if modulus == "prime":
        orc, keys, ctx = _mk_prime(12, seed=11, grouped=True, bits=3)
    else:
        orc, keys, ctx = _mk(12, seed=11, grouped=True, bits=3)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        W = orc.ggsw_words()
        for which in (3, 15):                        # K3 of the first group, K7 of the second
            broken = bskg.copy()
            broken[which * W:(which + 1) * W] = 0
            with pytest.raises(LfheError) as e:
                ctx.import_server_keys_raw(broken, ksk, None)
            assert "LFHE_ERR_UNSUPPORTED" in str(e.value) and "another key" in str(e.value)
            with pytest.raises(LfheError):           # the refused key left the context without server keys
                ctx.blind_rotate(np.zeros((1, orc.n + 1), dtype=np.uint64), ctx.make_lut(lambda x: x)[None], None)
        ctx.import_server_keys_raw(bskg, ksk, None)
        pb = orc.prepare_bsk_qp_grouped(bskg, 3) if modulus == "prime" else orc.prepare_bsk_grouped(bskg, 3)
        rotate = orc.blind_rotate_qp if modulus == "prime" else orc.blind_rotate
        r = Rng(seed_bytes(36))
        lut_o, lut_g = orc.make_lut(lambda x: (5 * x + 2) % 16), ctx.make_lut(lambda x: (5 * x + 2) % 16)
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(3)])
        got = ctx.blind_rotate(smalls, lut_g[None], None)
        for c in range(3):
            assert np.array_equal(got[c].reshape(-1), rotate(pb, smalls[c], lut_o)), c
    finally:
        ctx.close()


def test_grouped_key_pbs_full_parameters():
    """n = 918, noise reduction on, grouped key: outputs decrypt to f(m), a subsample is
    bit-identical to the oracle, the output noise stays far below half a message step."""
    orc, keys, ctx = _mk_prime(918, seed=7, ms_zeros=1449, grouped=True)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_qp_grouped(bskg)
        r = Rng(seed_bytes(30))
        f = [lambda x: x, lambda x: (x * x + 3) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        B = 600                                       # more than two per CU: a full-occupancy launch
        msgs = [int(r.u64() % 16) for _ in range(B)]
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in msgs])
        idx = np.array([i % 2 for i in range(B)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, idx)
        worst = 0
        for i in range(B):
            want = f[idx[i]](msgs[i]) % 16
            assert orc.decrypt_block(glwe_sk, out[i]) == want, i
            e = (orc.lwe_phase(glwe_sk, out[i]) - orc.encode(want)) % (1 << 64)
            worst = max(worst, min(e, (1 << 64) - e))
        assert worst < 1 << 53            # half a message step is 2^58
        for i in (0, 1, 333, 599):
            assert np.array_equal(out[i], orc.pbs_qp(ksk, pb, cts[i], luts_o[idx[i]])), i
        # and every one of the 600, bit for bit (the oracle on the host's cores)
        assert np.array_equal(out, orc.pbs_batch_qp(ksk, pb, cts, np.stack(luts_o), idx, nthreads=16))
        # a small launch takes the 512-thread kernel: same bits
        assert np.array_equal(ctx.pbs_batch(cts[:8], luts_g, idx[:8]), out[:8])
        # launches of more than two ciphertexts per CU put two ciphertexts in a workgroup; an odd batch
        # leaves the last workgroup with one; one ciphertext per workgroup (forced) gives the same bits
        assert np.array_equal(ctx.pbs_batch(cts[:577], luts_g, idx[:577]), out[:577])
        import os
        os.environ["LFHE_PAIR_ABOVE"] = "-1"
        try:
            from fhe_amd.native import Context
            single = Context(seed=seed_bytes(107), params=ctx.params)
        finally:
            del os.environ["LFHE_PAIR_ABOVE"]
        try:
            single.import_server_keys_raw(bskg, ksk, orc.test_zeros)
            assert np.array_equal(single.pbs_batch(cts, luts_g, idx), out)
        finally:
            single.close()
        # more than four workgroups per CU: the instantiation with a barrier after every phase, same bits
        big = ctx.pbs_batch(np.concatenate([cts, cts[:500]]), luts_g, np.concatenate([idx, idx[:500]]))
        assert np.array_equal(big[:B], out) and np.array_equal(big[B:], out[:500])
        # the key image is held in factored form (two products per GGSW entry); the plain image, which a
        # key with a non-invertible word falls back to, gives the same bits in every workgroup shape
        os.environ["LFHE_BSK_FACTORED"] = "0"
        try:
            plain = Context(seed=seed_bytes(108), params=ctx.params)
            plain.import_server_keys_raw(bskg, ksk, orc.test_zeros)
        finally:
            del os.environ["LFHE_BSK_FACTORED"]
        try:
            assert np.array_equal(plain.pbs_batch(cts, luts_g, idx), out)
            assert np.array_equal(plain.pbs_batch(cts[:8], luts_g, idx[:8]), out[:8])
            assert np.array_equal(plain.pbs_batch(cts[:300], luts_g, idx[:300]), out[:300])
        finally:
            plain.close()
    finally:
        ctx.close()


def test_padding_bit_tables_full_parameters():
    """Tables that use the padding bit (the operator schedules compose four carry states in one
    bootstrap this way): inputs 0 .. 30, entries modulo 32; an input x >= 16 comes back as minus the
    entry of x - 16.  Decrypted values as that rule says, every output bit-identical to the oracle."""
    orc, keys, ctx = _mk_prime(918, seed=9, ms_zeros=1449, grouped=True)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_qp_grouped(bskg)
        r = Rng(seed_bytes(31))
        # state - 1 of four composed states at position 3, the same resolved, and a table of mixed signs
        f = [lambda x: 0 if x == 15 else -8, lambda x: -1, lambda x: (x % 7) - 3]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        assert all(np.array_equal(a[-ctx.N:], b) and not a[:-ctx.N].any() for a, b in zip(luts_o, luts_g))
        msgs = [x for x in range(31)] * 3
        idx = np.array([i // 31 for i in range(len(msgs))], dtype=np.uint32)
        delta = orc.encode(1)
        cts = np.stack([orc.lwe_encrypt(glwe_sk, m * delta, orc.p.glwe_noise_log2, r) for m in msgs])
        out = ctx.pbs_batch(cts, luts_g, idx)
        for i, m in enumerate(msgs):
            want = f[idx[i]](m) if m < 16 else -f[idx[i]](m - 16)
            e = (orc.lwe_phase(glwe_sk, out[i]) - want * delta) % (1 << 64)
            assert min(e, (1 << 64) - e) < 1 << 53, (i, m, want)
        assert np.array_equal(out, orc.pbs_batch_qp(ksk, pb, cts, np.stack(luts_o), idx, nthreads=16))
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", ["split", "wide", "narrow"], ids=["two-CUs", "512-thread", "256-thread"])
@pytest.mark.parametrize("n,factored,bits", [(12, True, 2), (13, True, 2), (12, False, 2), (13, False, 2),
                                              (12, True, 3), (13, True, 3), (14, True, 3)],
                         ids=["12-factored", "13-factored", "12-plain", "13-plain", "12-triples", "13-triples", "14-triples"])
def test_grouped_key_over_the_torus_bit_exact(n, factored, bits, shape):
    """Two or three key bits per step, every modulus 2^64 (two-prime transform and lift per step, two
    ciphertexts per workgroup): accumulators and bootstrapped ciphertexts equal the oracle's bit for
    bit, for key lengths with and without a ragged last group, odd and even batches, all-zero masks,
    masks whose pairs cancel (a + a' = 2N) and accumulators whose digits are all extreme."""
    from fhe_amd.native import LfheError
    orc, keys, ctx = _mk(n, seed=9, grouped=True, factored=factored, bits=bits, shape=shape)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_grouped(bskg, bits)
        r = Rng(seed_bytes(33))
        W = orc.ggsw_words()
        glwe = np.stack([r.u64s(2 * N) for _ in range(3)])
        glwe[2][:] = 1 << 63
        gi = np.array([0, 5, (1 << bits) * ((n + bits - 1) // bits) - 1], dtype=np.uint32)
        if factored:    # the image holds (K2/K3, K1/K3, K3, K0 - K1 K2/K3) per group: its rows are no GGSWs
            with pytest.raises(LfheError):
                ctx.external_product(gi, glwe)
        else:           # the key rows as the external-product kernel sees them
            got = ctx.external_product(gi, glwe)
            for c in range(3):
                assert np.array_equal(got[c].reshape(-1), orc.external_product(bskg[gi[c] * W:(gi[c] + 1) * W], glwe[c])), c
        f = [lambda x: x, lambda x: (3 * x + 1) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(7)])
        smalls[3][:] = 0
        smalls[4][0::2] = np.uint64(1 << 63)          # a = N for every first element of a pair
        smalls[4][1:n:2] = np.uint64(1 << 63)         # ... and a' = N: a + a' = 2N
        smalls[5][0:n:2] = 0                          # a = 0, a' random
        li = np.array([0, 1, 1, 0, 1, 0, 1], dtype=np.uint32)
        got = ctx.blind_rotate(smalls, luts_g, li)    # seven: the last workgroup holds one ciphertext
        for c in range(7):
            assert np.array_equal(got[c].reshape(-1), orc.blind_rotate(pb, smalls[c], luts_o[li[c]])), c
        assert np.array_equal(ctx.blind_rotate(smalls[:6], luts_g, li[:6]), got[:6])
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in range(16)])
        li = np.array([m % 2 for m in range(16)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, li)
        for m in range(16):
            assert np.array_equal(out[m], orc.pbs(ksk, pb, cts[m], luts_o[li[m]])), m
            assert orc.decrypt_block(glwe_sk, out[m]) == f[li[m]](m) % 16
    finally:
        ctx.close()


@pytest.mark.parametrize("bits", [2, 3], ids=["pairs", "triples"])
def test_grouped_key_over_the_torus_full_parameters(bits):
    """n = 918, noise reduction on, grouped key over 2^64: outputs decrypt to f(m), a subsample is
    bit-identical to the oracle (the two 49-bit primes of the kernel against the two 62-bit primes of
    the oracle: the lift of every step stayed inside its window), noise far below half a message step."""
    orc, keys, ctx = _mk(918, seed=10, ms_zeros=1449, grouped=True, bits=bits)
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_grouped(bskg, bits)
        r = Rng(seed_bytes(34))
        f = [lambda x: x, lambda x: (x * x + 3) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        B = 601
        msgs = [int(r.u64() % 16) for _ in range(B)]
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in msgs])
        idx = np.array([i % 2 for i in range(B)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, idx)
        worst = 0
        for i in range(B):
            want = f[idx[i]](msgs[i]) % 16
            assert orc.decrypt_block(glwe_sk, out[i]) == want, i
            e = (orc.lwe_phase(glwe_sk, out[i]) - orc.encode(want)) % (1 << 64)
            worst = max(worst, min(e, (1 << 64) - e))
        assert worst < 1 << 53            # half a message step is 2^58
        for i in (0, 1, 333, 600):
            assert np.array_equal(out[i], orc.pbs(ksk, pb, cts[i], luts_o[idx[i]])), i
        # and every one of the 601, bit for bit (the oracle on the host's cores): every lift of every
        # step stayed inside its window
        assert np.array_equal(out, orc.pbs_batch(ksk, pb, cts, np.stack(luts_o), idx, nthreads=16))
        assert np.array_equal(ctx.pbs_batch(cts[:8], luts_g, idx[:8]), out[:8])
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", ["split", "wide", "narrow"], ids=["two-CUs", "512-thread", "256-thread"])
@pytest.mark.parametrize("factored,bits", [(True, 2), (False, 2), (True, 3)], ids=["factored", "plain", "triples"])
def test_lift_guard_of_the_grouped_key_over_the_torus(factored, bits, shape):
    """Every lift of the grouped key over 2^64 is checked against a guard band at the end of its window;
    a ciphertext that reaches the band is bootstrapped again, on the device, with the step taken apart
    into its four (eight with three key bits per step) unconditionally exact products.  Never taken at the shipped band (130 standard
    deviations out); with the band forced down to the middle of the window every ciphertext takes the
    slow path -- and must come back with exactly the oracle's bits, like the fast path."""
    import os
    orc, keys, ctx = _mk(13, seed=9, grouped=True, factored=factored, bits=bits, shape=shape)
    _, _, forced = _mk(13, seed=9, grouped=True, factored=factored, bits=bits, shape=shape, env={"LFHE_G64_GUARD_FRAC": "1e-4"})
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_grouped(bskg, bits)
        r = Rng(seed_bytes(35))
        f = [lambda x: x, lambda x: (5 * x + 2) % 16]
        luts_o = [orc.make_lut(g) for g in f]
        luts_g = np.stack([ctx.make_lut(g) for g in f])
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(9)])
        smalls[3][:] = 0
        li = np.array([i % 2 for i in range(9)], dtype=np.uint32)
        want = np.stack([orc.blind_rotate(pb, smalls[c], luts_o[li[c]]).reshape(2, N) for c in range(9)])
        assert np.array_equal(ctx.blind_rotate(smalls, luts_g, li), want)
        assert ctx.lift_fallbacks() == 0
        assert np.array_equal(forced.blind_rotate(smalls, luts_g, li), want)
        # ciphertext 3 has an all-zero mask and body: every step multiplies by the GGSW of 1, whose lift is tiny
        assert 7 <= forced.lift_fallbacks() <= 9
        cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in range(16)] * 20)     # 320: two per workgroup
        li = np.array([m % 2 for m in range(320)], dtype=np.uint32)
        out = ctx.pbs_batch(cts, luts_g, li)
        before = forced.lift_fallbacks()
        assert np.array_equal(forced.pbs_batch(cts, luts_g, li), out)
        assert forced.lift_fallbacks() - before >= 300 and ctx.lift_fallbacks() == 0
        for m in range(16):
            assert np.array_equal(out[m], orc.pbs(ksk, pb, cts[m], luts_o[li[m]])), m
    finally:
        ctx.close()
        forced.close()


@pytest.mark.parametrize("bits", [2, 3], ids=["pairs", "triples"])
def test_two_cu_kernel_gives_up_and_the_slow_path_takes_over(bits):
    """The two workgroups of a ciphertext wait for each other a bounded number of polls.  With the bound
    forced to zero a workgroup gives up at the first poll that does not find its partner's counter raised,
    marks its ciphertext and leaves (its partner follows); the slow path behind the launch bootstraps every
    marked ciphertext: the oracle's bits either way, fallbacks counted, and the context goes on working."""
    orc, keys, ctx = _mk(13, seed=9, grouped=True, bits=bits, env={"LFHE_G64_SPLIT_POLL": "0"})
    try:
        lwe_sk, glwe_sk, bskg, ksk = keys
        pb = orc.prepare_bsk_grouped(bskg, bits)
        r = Rng(seed_bytes(37))
        lut_o, lut_g = orc.make_lut(lambda x: (3 * x + 1) % 16), ctx.make_lut(lambda x: (3 * x + 1) % 16)
        smalls = np.stack([r.u64s(orc.n + 1) for _ in range(11)])
        want = np.stack([orc.blind_rotate(pb, smalls[c], lut_o).reshape(2, N) for c in range(11)])
        assert np.array_equal(ctx.blind_rotate(smalls, lut_g[None], None), want)
        first = ctx.lift_fallbacks()
        assert 1 <= first <= 11          # a pair that happens to run in step never finds its partner missing
        assert np.array_equal(ctx.blind_rotate(smalls[:5], lut_g[None], None), want[:5])
        assert first <= ctx.lift_fallbacks() <= first + 5
    finally:
        ctx.close()


@pytest.mark.parametrize("params_id", [1, 2, 3, 4, 5, 6])
def test_library_keygen_round_trip(params_id):
    """Keys made by the library itself (both bootstrapping-key moduli): encrypt -> bootstrap ->
    decrypt, and the oracle decrypts library ciphertexts with the exported secret key."""
    from fhe_amd.native import Context, KEY_CLIENT
    ctx = Context(seed=seed_bytes(77), params_id=params_id)
    try:
        ctx.keygen()
        blob = ctx.export_keys(KEY_CLIENT)
        words = np.frombuffer(blob[-(918 + 2048) * 8:], dtype=np.uint64)
        glwe_sk = words[918:].copy()
        orc = Oracle()
        lut = ctx.make_lut(lambda x: (x + 5) % 16)
        for v, w in ((0xA7, 8), (1, 1), (0xBEEF, 16)):
            ct = ctx.encrypt(v, w)
            assert ctx.decrypt(ct) == v
            blocks = np.frombuffer(ct[16:], dtype=np.uint64).reshape(-1, N + 1)
            digits = [orc.decrypt_block(glwe_sk, b) for b in blocks]
            assert sum(d << (2 * i) for i, d in enumerate(digits)) == v
            out = ctx.pbs_batch(blocks, lut[None])
            assert [orc.decrypt_block(glwe_sk, b) for b in out] == [(d + 5) % 16 for d in digits]
    finally:
        ctx.close()


@pytest.mark.parametrize("params_id", [1, 2, 3, 4, 5, 6],
                         ids=["bsk_q64", "bsk_prime", "bsk_prime_grouped", "bsk_q64_grouped", "bsk_prime_grouped3", "bsk_q64_grouped3"])
def test_full_size_batch_properties(params_id):
    """The benchmark's configuration (4096 blocks, 16 distinct tables) through size-independent
    properties: every output decrypts to f(m); a second pass gives the same bits (no dependence on
    scheduling); bootstrapping the identity's outputs with f equals bootstrapping the inputs with
    f as decrypted values; the output noise stays far inside the decision margin."""
    from fhe_amd.native import Context, KEY_CLIENT
    ctx = Context(seed=seed_bytes(90), params_id=params_id)
    try:
        ctx.keygen()
        blob = ctx.export_keys(KEY_CLIENT)
        sk = np.frombuffer(blob[-(918 + 2048) * 8:], dtype=np.uint64)[918:].copy()
        B = 4096
        rng = np.random.default_rng(17)
        vals = rng.integers(0, 256, size=B // 4)
        blocks = np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1)
                                 for v in vals])
        msgs = np.array([(int(vals[i // 4]) >> (2 * (i % 4))) & 3 for i in range(B)])
        f = [(lambda x, j=j: ((2 * j + 1) * x + j) % 16) for j in range(16)]
        luts = np.stack([ctx.make_lut(g) for g in f])
        idx = (np.arange(B) % 16).astype(np.uint32)
        want = np.array([f[idx[i]](int(msgs[i])) for i in range(B)], dtype=np.uint64)

        def decode(cts):
            with np.errstate(over="ignore"):
                phase = cts[:, N] - (cts[:, :N] * sk[None, :]).sum(axis=1, dtype=np.uint64)
            m = ((phase + np.uint64(1 << 58)) >> np.uint64(59)) & np.uint64(15)
            err = (phase - (m << np.uint64(59))).astype(np.int64)
            return m, np.abs(err).max()
        out = ctx.pbs_batch(blocks, luts, idx)
        got, worst = decode(out)
        assert np.array_equal(got, want)
        assert worst < 1 << 53
        assert np.array_equal(ctx.pbs_batch(blocks, luts, idx), out)
        # a subsample spread over the launch is the oracle's output bit for bit, under the keys the LIBRARY
        # made (exported, handed to the oracle as raw words)
        from fhe_amd.native import KEY_SERVER
        words = np.frombuffer(ctx.export_keys(KEY_SERVER), dtype=np.uint8)
        n_ggsw = {1: 918, 2: 918, 3: 1836, 4: 1836, 5: 2448, 6: 2448}[params_id]
        nb, nk, nz = n_ggsw * 4 * N, N * 4 * 919, 1449 * 919
        raw = np.frombuffer(words[-(nb + nk + nz) * 8:].tobytes(), dtype=np.uint64)
        bsk_raw, ksk_raw, zeros_raw = raw[:nb].copy(), raw[nb:nb + nk].copy(), raw[nb + nk:].reshape(1449, 919).copy()
        orc = Oracle()
        orc.set_ms(zeros_raw)
        prep = {1: orc.prepare_bsk, 2: orc.prepare_bsk_qp, 3: orc.prepare_bsk_qp_grouped, 4: orc.prepare_bsk_grouped,
                5: lambda k: orc.prepare_bsk_qp_grouped(k, 3), 6: lambda k: orc.prepare_bsk_grouped(k, 3)}[params_id]
        batch = orc.pbs_batch if params_id in (1, 4, 6) else orc.pbs_batch_qp
        pick = np.arange(0, B, 31)[:128]
        luts_o = np.stack([np.concatenate([np.zeros(N, dtype=np.uint64), l]) for l in luts])
        assert np.array_equal(out[pick], batch(ksk_raw, prep(bsk_raw), blocks[pick], luts_o, idx[pick], nthreads=16))
        ident = ctx.pbs_batch(blocks, luts[:1])
        got2, _ = decode(ctx.pbs_batch(ident, luts, idx))
        assert np.array_equal(got2, want)
    finally:
        ctx.close()
