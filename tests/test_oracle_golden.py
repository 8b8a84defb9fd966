"""The CPU oracle against the reference's own fixtures and against its own definitions."""
import numpy as np
import pytest

from oracle.oracle import Oracle, Params, Rng, load_reference_cpk_fixture, seed_bytes, lib

M64 = (1 << 64) - 1
SMALL = Params(lwe_dim=24, poly_size=256, lwe_noise_log2=30, glwe_noise_log2=10)


def centered(x):
    return x.astype(np.int64)


def test_reference_fixture_parameters(golden_dir):
    """privateKey.bin carries the numeric parameter set the oracle defaults to
    (reference: sdk/relayer/src/tfhe.ts:14-19 names it; the blob holds the values)."""
    blob = open(golden_dir + "/privateKey.bin", "rb").read()
    u64 = lambda off: int.from_bytes(blob[off:off + 8], "little")
    p = Params()
    assert u64(23867) == p.lwe_dim
    assert u64(23879) == p.glwe_dim
    assert u64(23891) == p.poly_size
    assert int.from_bytes(blob[23911:23915], "little") == p.lwe_noise_log2
    assert int.from_bytes(blob[23927:23931], "little") == p.glwe_noise_log2
    assert u64(23935) == p.pbs_base_log and u64(23947) == p.pbs_level
    assert u64(23959) == p.ks_base_log and u64(23971) == p.ks_level
    assert u64(23983) == p.message_modulus and u64(23995) == p.carry_modulus


def test_reference_fixture_noise_reduction_parameters(golden_dir):
    """The modulus_switch_noise_reduction fields of the same blob are the oracle's defaults."""
    import struct
    blob = open(golden_dir + "/privateKey.bin", "rb").read()
    m = Oracle().ms_params()
    assert int.from_bytes(blob[24068:24072], "little") == m.zeros_count == 1449
    assert struct.unpack("<d", blob[24080:24088])[0] == 2.0 ** m.bound_log2
    assert struct.unpack("<d", blob[24092:24100])[0] == m.r_sigma_factor
    assert struct.unpack("<d", blob[24104:24112])[0] == m.input_variance


def test_ms_noise_reduction_properties():
    """First candidate within the bound wins (every earlier one exceeds it), the estimate agrees
    with its floating-point definition, the plaintext is untouched, and the actual rounding error
    under the real key shrinks."""
    orc = Oracle()
    n = orc.n
    lwe_sk = np.zeros(n, dtype=np.uint64)
    lib().orc_gen_binary_key(Rng(seed_bytes(3), 1).ref, lwe_sk, n)
    zeros = orc.gen_ms_zeros(lwe_sk, seed_bytes(3), 1449)
    for z in zeros[:50]:
        ph = orc.lwe_phase(lwe_sk, z)
        assert min(ph, (1 << 64) - ph) <= 1 << 45                            # encryptions of zero
    m = orc.ms_params()
    bound = 1 << (m.bound_log2 - 26)
    r = Rng(seed_bytes(7))

    def errors(ct):
        t = (ct >> np.uint64(51)) + np.uint64(1) >> np.uint64(1)
        return ((t << np.uint64(52)) - ct).astype(np.int64)

    def float_estimate(ct):
        e = errors(ct).astype(np.float64) / 2.0 ** 64
        mean = e[-1] - e[:-1].sum() / 2
        return (abs(mean) + m.r_sigma_factor * np.sqrt(m.input_variance + (e[:-1] ** 2).sum() / 4)) * 2.0 ** 38

    def true_error(ct):   # rounding error of the phase under the real key, in units of the 2N grid
        e = errors(ct)
        return (int(e[-1]) - int((e[:-1] * lwe_sk.astype(np.int64)).sum())) / 2.0 ** 52

    before, after, tries = [], [], []
    for i in range(120):
        ct = orc.lwe_encrypt(lwe_sk, orc.encode(i % 16), 45, r)
        assert abs(orc.ms_estimate(ct) / float_estimate(ct) - 1) < 1e-5
        out, pick = orc.ms_noise_reduce(ct, zeros)
        est = orc.ms_estimate(out)
        assert est <= bound
        for k in range(-1, pick):   # the candidates before the chosen one all exceed the bound
            cand = ct if k < 0 else ct + zeros[k]
            assert orc.ms_estimate(cand) > bound
        if pick >= 0:
            assert np.array_equal(out, ct + zeros[pick])
        else:
            assert np.array_equal(out, ct)
        assert orc.decode(orc.lwe_phase(lwe_sk, out)) == i % 16
        before.append(true_error(ct)); after.append(true_error(out)); tries.append(pick + 2)
    assert 5 < np.mean(tries) < 60
    # the estimate bounds the mean part of the error: its spread under the real key drops
    assert np.std(after) < 0.9 * np.std(before)
    # no satisfying candidate: the best one is taken
    tight = orc.ms_params(40)
    tight.bound_log2 = 57
    ct = orc.lwe_encrypt(lwe_sk, 0, 45, r)
    out, pick = orc.ms_noise_reduce(ct, zeros[:40], tight)
    ests = [orc.ms_estimate(ct)] + [orc.ms_estimate(ct + z) for z in zeros[:40]]
    assert pick == int(np.argmin(ests)) - 1 and orc.ms_estimate(out) == min(ests)


def test_negacyclic_product_pinned_by_reference_public_key(golden_dir):
    """Known answer from the reference's key fixtures: body = mask (*) rev(s) + e, |e| <= 2^17.
    Any error in the sign convention, the wrap or the CRT lift gives a full-range residual."""
    s, mask, body = load_reference_cpk_fixture(golden_dir)
    assert set(np.unique(s)) <= {0, 1} and int(s.sum()) == 975
    orc = Oracle()
    rev = np.ascontiguousarray(s[::-1])
    for mul in (orc.mul_ntt, orc.mul_schoolbook):
        e = centered(body - mul(rev, mask))
        assert np.abs(e).max() <= 1 << 17
        assert 1000 < np.abs(e).mean() < (1 << 17)
    # and the wrong conventions really are detected
    bad = centered(body - orc.mul_ntt(s, mask))
    assert np.abs(bad).max() > 1 << 60


@pytest.mark.parametrize("N", [256, 2048])
def test_ntt_route_equals_schoolbook(N):
    orc = Oracle(Params(poly_size=N))
    r = Rng(seed_bytes(3))
    b = r.u64s(N)
    small = np.array([(r.u64() % (1 << 23)) - (1 << 22) for _ in range(N)], dtype=np.int64)
    assert np.array_equal(orc.mul_ntt(small, b), orc.mul_schoolbook(small, b))
    # extreme magnitudes: the integer result reaches the 2^96 bound the CRT modulus must cover
    small[:] = 1 << 22
    b[:] = 1 << 63
    assert np.array_equal(orc.mul_ntt(small, b), orc.mul_schoolbook(small, b))
    small[::2] = -(1 << 22)
    b[:] = (1 << 63) - 1
    assert np.array_equal(orc.mul_ntt(small, b), orc.mul_schoolbook(small, b))


def test_decompose_matches_definition():
    r = Rng(seed_bytes(5))
    for base_log, levels in ((23, 1), (4, 4), (8, 3)):
        for _ in range(2000):
            x = r.u64()
            d = Oracle.decompose(x, base_log, levels)
            B = 1 << base_log
            assert all(-B // 2 <= int(v) <= B // 2 for v in d)
            rec = sum(int(v) << (64 - base_log * (j + 1)) for j, v in enumerate(d)) & M64
            err = (x - rec) & M64
            err = err - (1 << 64) if err >= 1 << 63 else err
            assert abs(err) <= 1 << (64 - base_log * levels - 1)
    # the tie case keeps +B/2 (tfhe-rs balanced rule), it does not wrap to -B/2
    assert int(Oracle.decompose(1 << 63, 23, 1)[0]) == 1 << 22
    assert int(Oracle.decompose((1 << 63) + (1 << 41), 23, 1)[0]) == -(1 << 22) + 1


def test_lwe_and_glwe_round_trip():
    orc = Oracle(SMALL)
    lwe_sk, glwe_sk, _, _ = orc.gen_keys(seed_bytes(1))
    r = Rng(seed_bytes(2))
    for m in range(16):
        ct = orc.encrypt_block(glwe_sk, m, r)
        assert orc.decrypt_block(glwe_sk, ct) == m
    pt = np.array([orc.encode(i % 16) for i in range(orc.N)], dtype=np.uint64)
    ct = orc.glwe_encrypt(glwe_sk, pt, r)
    ph = orc.glwe_phase(glwe_sk, ct)
    assert np.abs(centered(ph - pt)).max() <= 1 << SMALL.glwe_noise_log2


def test_external_product_and_cmux():
    orc = Oracle(SMALL)
    lwe_sk, glwe_sk, bsk, _ = orc.gen_keys(seed_bytes(7))
    r = Rng(seed_bytes(8))
    pt = np.array([orc.encode(i % 16) for i in range(orc.N)], dtype=np.uint64)
    glwe = orc.glwe_encrypt(glwe_sk, pt, r)
    W = orc.ggsw_words()
    for i in range(6):
        ggsw = bsk[i * W:(i + 1) * W]
        out = orc.external_product(ggsw, glwe)
        ph = orc.glwe_phase(glwe_sk, out)
        want = pt * lwe_sk[i]
        assert np.abs(centered(ph - want)).max() < 1 << 52
        rot = 37 + i
        acc = orc.cmux_rotate(ggsw, rot, glwe)
        ph = orc.glwe_phase(glwe_sk, acc)
        want = orc.monomial_mul(pt, rot) if lwe_sk[i] else pt
        assert np.abs(centered(ph - want)).max() < 1 << 53


def test_keyswitch_and_pbs_small_params():
    orc = Oracle(SMALL)
    lwe_sk, glwe_sk, bsk, ksk = orc.gen_keys(seed_bytes(9))
    pb = orc.prepare_bsk(bsk)
    r = Rng(seed_bytes(10))
    ident = orc.make_lut(lambda x: x)
    sq = orc.make_lut(lambda x: (x * x + 1) % 16)
    for m in range(16):
        ct = orc.encrypt_block(glwe_sk, m, r)
        small = orc.keyswitch(ksk, ct)
        assert orc.decode(orc.lwe_phase(lwe_sk, small)) == m
        assert orc.decrypt_block(glwe_sk, orc.pbs(ksk, pb, ct, ident)) == m
        assert orc.decrypt_block(glwe_sk, orc.pbs(ksk, pb, ct, sq)) == (m * m + 1) % 16
    cts = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in range(8)])
    luts = np.stack([ident, sq])
    idx = np.array([0, 1] * 4, dtype=np.uint32)
    out = orc.pbs_batch(ksk, pb, cts, luts, idx, nthreads=2)
    for m in range(8):
        want = m if m % 2 == 0 else (m * m + 1) % 16
        assert orc.decrypt_block(glwe_sk, out[m]) == want
        assert np.array_equal(out[m], orc.pbs(ksk, pb, cts[m], luts[idx[m]]))


def test_ntt_definition_is_the_transform_the_product_uses():
    """out = INTT(NTT(a) . NTT(b)) with the O(N^2) definition equals the negacyclic product mod p."""
    N, p = 64, 7681  # 7681 = 60*128 + 1
    L = lib()
    psi = next(c for c in (L.orc_powmod(g, (p - 1) // (2 * N), p) for g in range(2, 50))
               if L.orc_powmod(c, N, p) == p - 1)
    r = Rng(seed_bytes(11))
    a = r.u64s(N) % p
    b = r.u64s(N) % p
    A, B = Oracle.ntt_definition(p, psi, a), Oracle.ntt_definition(p, psi, b)
    Cc = (A.astype(object) * B.astype(object)) % p
    want = np.zeros(N, dtype=object)
    for i in range(N):
        for j in range(N):
            k = i + j
            t = int(a[i]) * int(b[j])
            if k < N:
                want[k] += t
            else:
                want[k - N] -= t
    want %= p
    # inverse by definition: a_n = N^-1 sum_j C_j psi^-(2j+1)n
    ninv = pow(N, p - 2, p)
    psi_inv = pow(int(psi), p - 2, p)
    got = [(ninv * sum(int(Cc[j]) * pow(psi_inv, (2 * j + 1) * n, p) for j in range(N))) % p for n in range(N)]
    assert got == list(want)


@pytest.mark.parametrize("n", [12, 13])
def test_grouped_key_blind_rotation_over_prime_modulus(n):
    """Two key bits per step (four GGSWs per pair, one per bit pattern): the bootstrap
    evaluates the table, and the accumulator is the rotated table up to noise -- for an even and an
    odd key length (the odd one pads the last pair with a zero bit)."""
    orc = Oracle(Params(lwe_dim=n))
    lwe_sk, glwe_sk, _, ksk = orc.gen_keys(seed_bytes(21))
    bskg = orc.gen_bsk_qp_grouped(lwe_sk, glwe_sk, seed_bytes(21))
    assert bskg.size == 4 * ((n + 1) // 2) * orc.ggsw_words()
    assert int(bskg.max()) < orc.qp
    pb = orc.prepare_bsk_qp_grouped(bskg)
    pb1 = orc.prepare_bsk_qp(orc.gen_bsk_qp(lwe_sk, glwe_sk, seed_bytes(21)))
    r = Rng(seed_bytes(22))
    f = lambda x: (7 * x + 2) % 16
    lut = orc.make_lut(f)
    q = orc.qp
    for m in range(16):
        ct = orc.encrypt_block(glwe_sk, m, r)
        out = orc.pbs_qp(ksk, pb, ct, lut)
        assert orc.decrypt_block(glwe_sk, out) == f(m)
        # the grouped and the one-bit rotation end on the same plaintext polynomial: their
        # accumulators differ by noise only
        small = orc.keyswitch(ksk, ct)
        a2 = orc.blind_rotate_qp(pb, small, lut).astype(object)
        a1 = orc.blind_rotate_qp(pb1, small, lut).astype(object)
        def phase(acc):   # b - a (*) s over Z_q'
            a, b = acc[:orc.N], acc[orc.N:]
            out = list(b)
            s = [int(v) for v in glwe_sk]
            for i, si in enumerate(s):
                if si:
                    for j in range(orc.N):
                        k = i + j
                        if k < orc.N:
                            out[k] -= a[j]
                        else:
                            out[k - orc.N] += a[j]
            return [int(v) % q for v in out]
        if m < 2:
            d = [(x - y) % q for x, y in zip(phase(a2), phase(a1))]
            assert max(min(v, q - v) for v in d) < q >> 8
    # all-zero mask: every step still runs (the sequence of operations does not depend on the
    # ciphertext) and multiplies by an encryption of 1: the accumulator is the table itself, rotated
    # by the body, up to noise
    small = np.zeros(n + 1, dtype=np.uint64)
    z2 = orc.blind_rotate_qp(pb, small, lut)
    z1 = orc.blind_rotate_qp(pb1, small, lut)
    d = [(x - y) % q for x, y in zip(phase(z2.astype(object)), phase(z1.astype(object)))]
    assert max(min(v, q - v) for v in d) < q >> 8
    assert not np.array_equal(z2, z1)


def test_compression_oracle_round_trip():
    """Packing key switch + storage modulus + extraction keep the (shifted) message: the phase of
    every extracted block under the compression secret is 4 m delta up to a fraction of a box, and
    the decompression bootstrap returns m under the compute key."""
    from oracle import compress as OC
    orc = Oracle()
    lwe_sk, glwe_sk, _, _ = orc.gen_keys(seed_bytes(50))
    comp_sk = OC.gen_comp_sk(Rng(seed_bytes(51), 1))
    pksk = OC.gen_packing_ksk(glwe_sk, comp_sk, Rng(seed_bytes(51), 2))
    r = Rng(seed_bytes(52))
    msgs = [i % 4 for i in range(9)]
    blocks = np.stack([orc.encrypt_block(glwe_sk, m, r) for m in msgs])
    glwe = OC.pack(blocks, pksk, 4)
    assert glwe.shape == (1, 5 * 256) and int(glwe.max()) < 1 << 12
    small = OC.extract(glwe, len(msgs))
    for m, s in zip(msgs, small):
        phase = (int(s[-1]) - int((s[:-1] * comp_sk).sum(dtype=np.uint64))) % (1 << 64)
        err = (phase - 4 * m * orc.p.delta) % (1 << 64)
        assert min(err, (1 << 64) - err) < 1 << 57              # a quarter of a box; the table tolerates two boxes
    packed = OC.pack_bits(glwe[0][:1024 + len(msgs)], 12)
    assert len(packed) == ((1024 + len(msgs)) * 12 + 7) // 8
    dec = OC.Decompressor("prime", comp_sk, glwe_sk, seed_bytes(53))
    out = dec.bootstrap(small[3], dec.table())
    assert orc.decrypt_block(glwe_sk, out) == msgs[3]


@pytest.mark.parametrize("n,bits", [(10, 2), (11, 2), (10, 3), (11, 3), (12, 3)])
def test_grouped_key_blind_rotation_over_the_torus(n, bits):
    """Two or three key bits per step with every modulus 2^64: the bootstrap evaluates the table; the
    accumulator carries the same plaintext polynomial as the one-bit rotation's (glwe_phase differs by
    noise only); one step equals the one-bit steps it replaces up to noise."""
    orc = Oracle(Params(lwe_dim=n))
    lwe_sk, glwe_sk, bsk, ksk = orc.gen_keys(seed_bytes(31))
    bskg = orc.gen_bsk_grouped(lwe_sk, glwe_sk, seed_bytes(31), bits)
    W = orc.ggsw_words()
    assert bskg.size == (1 << bits) * ((n + bits - 1) // bits) * W
    pb, pb1 = orc.prepare_bsk_grouped(bskg, bits), orc.prepare_bsk(bsk)
    r = Rng(seed_bytes(32))
    f = lambda x: (7 * x + 2) % 16
    lut = orc.make_lut(f)

    def close(x, y, bound):
        d = (x.astype(np.uint64) - y.astype(np.uint64)).astype(np.int64)
        return int(np.abs(d).max()) < bound
    for m in range(16):
        ct = orc.encrypt_block(glwe_sk, m, r)
        assert orc.decrypt_block(glwe_sk, orc.pbs(ksk, pb, ct, lut)) == f(m)
        if m < 2:
            small = orc.keyswitch(ksk, ct)
            a2, a1 = orc.blind_rotate(pb, small, lut), orc.blind_rotate(pb1, small, lut)
            assert close(orc.glwe_phase(glwe_sk, a2), orc.glwe_phase(glwe_sk, a1), 1 << 52)
            assert not np.array_equal(a2, a1)
    # one grouped step against the CMux steps of the same key bits
    acc = orc.glwe_encrypt(glwe_sk, r.u64s(orc.N) & np.uint64(~((1 << 50) - 1) & (2**64 - 1)), r)
    exps = [1234, 3071, 4095][:bits]
    two = acc
    for i, e in enumerate(exps):
        two = orc.cmux_rotate(bsk[i * W:(i + 1) * W], e, two)
    one = orc.group_step(bskg[:(1 << bits) * W], exps, acc)
    assert close(orc.glwe_phase(glwe_sk, one), orc.glwe_phase(glwe_sk, two), 1 << 48)

