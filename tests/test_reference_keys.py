"""The key blobs of the reference's SDK (sdk/relayer/src/test/keys/{privateKey,publicKey}.bin, copies
under tests/golden/; made by sdk/relayer/generateKeys.js with node-tfhe 1.4.0-alpha.3) drive the
backend: the parameter set and the secret keys are read from the ClientKey, server keys are
generated for them on the GPU box, and the reference's compact public key encrypts values that its
own secret key decrypts."""
import os
import struct

import numpy as np
import pytest

from fhe_amd import tfhers_wire as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _blobs():
    return (open(os.path.join(GOLDEN, "privateKey.bin"), "rb").read(),
            open(os.path.join(GOLDEN, "publicKey.bin"), "rb").read())


def test_client_key_blob_is_read():
    priv, pub = _blobs()
    k = W.parse_client_key(priv)
    p = k["params"]
    assert (p["lwe_dim"], p["glwe_dim"], p["poly_size"]) == (918, 1, 2048)
    assert (p["pbs_base_log"], p["pbs_level"], p["ks_base_log"], p["ks_level"]) == (23, 1, 4, 4)
    assert (p["message_modulus"], p["carry_modulus"], p["max_noise_level"]) == (4, 4, 5)
    assert (p["lwe_noise_log2"], p["glwe_noise_log2"]) == (45, 17)
    assert p["ms_zeros_count"] == 1449 and p["ms_bound"] == 2.0 ** 58
    # Hamming weights of the three binary keys of the fixture
    assert (int(k["glwe_sk"].sum()), int(k["lwe_sk"].sum()), int(k["pke_sk"].sum())) == (996, 460, 975)
    mask, body = W.parse_compact_public_key(pub)
    assert len(mask) == len(body) == 2048
    # the library's default parameters are exactly the blob's
    from fhe_amd.native import default_params
    d = default_params()
    for name in ("lwe_dim", "glwe_dim", "poly_size", "pbs_base_log", "pbs_level", "ks_base_log", "ks_level",
                 "message_modulus", "carry_modulus", "lwe_noise_log2", "glwe_noise_log2", "ms_zeros_count"):
        assert getattr(d, name) == p[name], name
    assert 2.0 ** d.ms_bound_log2 == p["ms_bound"] and d.ms_r_sigma_factor == p["ms_r_sigma_factor"]
    assert d.ms_input_variance == p["ms_input_variance"]
    with pytest.raises(ValueError):
        W.parse_client_key(bytes(5000))
    lfpk = W.compact_public_key_to_lfpk(pub)
    assert struct.unpack("<IIII", lfpk[:16]) == (0x4B50464C, 2048, 17, 1) and len(lfpk) == 16 + 2 * 2048 * 8


def test_key_blobs_are_written_in_the_reference_format():
    """Emission is pinned by the fixtures themselves: their vectors, re-emitted, give the fixtures back
    byte for byte; other keys round-trip through the reader; a public key made here for a secret key
    has the residual body - mask (*) rev(s) the reference's own public key has (noise of 2^17)."""
    priv, pub = _blobs()
    k = W.parse_client_key(priv)
    assert W.emit_client_key(k["glwe_sk"], k["lwe_sk"], k["pke_sk"]) == priv
    mask, body = W.parse_compact_public_key(pub)
    assert W.emit_compact_public_key(mask, body) == pub
    rng = np.random.default_rng(4)
    mine = [rng.integers(0, 2, size=n, dtype=np.uint64) for n in W.CLIENT_KEY_DIMS]
    blob = W.emit_client_key(*mine)
    assert len(blob) == len(priv)
    back = W.parse_client_key(blob)
    assert all(np.array_equal(a, b) for a, b in zip(mine, (back["glwe_sk"], back["lwe_sk"], back["pke_sk"])))
    assert back["params"] == k["params"]
    with pytest.raises(ValueError):
        W.emit_client_key(mine[0][:-1], mine[1], mine[2])
    with pytest.raises(ValueError):
        W.emit_client_key(mine[0] + np.uint64(1), mine[1], mine[2])

    def residual(mask, body, s):
        n = len(s)
        acc = body.copy()
        with np.errstate(over="ignore"):
            for i in np.nonzero(s[::-1])[0]:
                i = int(i)
                acc[i:] -= mask[:n - i]
                acc[:i] += mask[n - i:]
        return np.abs(acc.astype(np.int64)).max()
    assert residual(mask, body, k["pke_sk"]) <= 1 << 17                       # the reference's own key
    m2, b2 = W.make_compact_public_key(mine[2], bytes(range(16)))
    assert residual(m2, b2, mine[2]) <= 1 << 17
    m3, b3 = W.parse_compact_public_key(W.emit_compact_public_key(m2, b2))
    assert np.array_equal(m3, m2) and np.array_equal(b3, b2)


@pytest.mark.gpu
def test_keys_exported_in_the_reference_format_come_back_and_work():
    """Library keys -> reference-format ClientKey / CompactPublicKey blobs -> read back like the
    reference's own fixtures: encryption under the emitted public key, cast, computation, decryption."""
    from fhe_amd.native import Context, KEY_CLIENT
    src = Context(seed=bytes(range(32)), params_id=3)
    ctx = Context(seed=bytes(range(1, 33)), params_id=3)
    try:
        src.keygen()
        words = np.frombuffer(src.export_keys(KEY_CLIENT)[-(918 + 2048) * 8:], dtype=np.uint64)
        lwe_sk, glwe_sk = words[:918].copy(), words[918:].copy()
        pke_sk = np.random.default_rng(9).integers(0, 2, size=2048, dtype=np.uint64)
        priv = W.emit_client_key(glwe_sk, lwe_sk, pke_sk)
        pub = W.emit_compact_public_key(*W.make_compact_public_key(pke_sk, bytes(range(16))))
        k = W.parse_client_key(priv)
        ctx.import_client_keys_raw(k["lwe_sk"], k["glwe_sk"])
        ctx.keygen_server()
        ctx.set_pke_key(k["pke_sk"])
        lfpk = W.compact_public_key_to_lfpk(pub)
        a = ctx.cast_from_pke(ctx.encrypt_public(lfpk, 0xC0DE, 16))
        assert ctx.decrypt(a) == 0xC0DE
        assert src.decrypt(a) == 0xC0DE                      # the same compute key as the exporting context's
        assert ctx.decrypt(ctx.evaluate("add", 16, a, a)) == (2 * 0xC0DE) & 0xFFFF
    finally:
        src.close()
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("params_id", [1, 2])
def test_reference_client_key_drives_the_backend(params_id):
    from fhe_amd.native import Context
    from oracle.oracle import Oracle
    priv, _ = _blobs()
    k = W.parse_client_key(priv)
    ctx = Context(seed=bytes(range(32)), params_id=params_id)
    try:
        ctx.import_client_keys_raw(k["lwe_sk"], k["glwe_sk"])
        ctx.keygen_server()
        a, b = ctx.encrypt(0xC0FFEE, 32), ctx.encrypt(0x123456, 32)
        out = ctx.evaluate("mul", 32, a, b)
        assert ctx.decrypt(out) == (0xC0FFEE * 0x123456) & 0xFFFFFFFF
        assert ctx.decrypt(ctx.evaluate("gt", 32, a, b)) == 1
        # an independent decryption of the result with the blob's key
        orc = Oracle()
        blocks = np.frombuffer(out[16:], dtype=np.uint64).reshape(-1, 2049)
        digits = [orc.decrypt_block(k["glwe_sk"], blk) for blk in blocks]
        assert sum((d & 3) << (2 * i) for i, d in enumerate(digits)) == (0xC0FFEE * 0x123456) & 0xFFFFFFFF
    finally:
        ctx.close()


@pytest.mark.gpu
def test_reference_public_key_encrypts_for_its_secret_key():
    """publicKey.bin (tfhe-rs) -> lfhe_encrypt_public -> decryption with the secret key stored in
    privateKey.bin (tfhe-rs): the public-key encryption is checked against keys this code did not make."""
    from fhe_amd.native import Context, LfheError
    priv, pub = _blobs()
    k = W.parse_client_key(priv)
    lfpk = W.compact_public_key_to_lfpk(pub)
    ctx = Context(seed=bytes(range(32)))
    try:
        ctx.import_client_keys_raw(k["lwe_sk"], k["pke_sk"])       # the PKE secret as the decryption key
        for v, w in ((0xA7, 8), (0xDEADBEEFCAFEF00D, 64), (1, 1), ((1 << 256) - 12345, 256)):
            ct = ctx.encrypt_public(lfpk, v, w)
            assert ctx.decrypt(ct) == v, (v, w)
        # without the reversed-key flag the same words decrypt to noise
        plain = lfpk[:12] + struct.pack("<I", 0) + lfpk[16:]
        assert ctx.decrypt(ctx.encrypt_public(plain, 0xDEADBEEFCAFEF00D, 64)) != 0xDEADBEEFCAFEF00D
        with pytest.raises(LfheError):
            ctx.encrypt_public(lfpk[:100], 1, 8)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_inputs_under_the_reference_public_key_are_cast_and_computed_on():
    """The reference's flow for user inputs: encrypt with the compact public key, cast to the compute
    key on the server, operate, decrypt with the compute secret key -- all three keys from the blobs."""
    from fhe_amd.native import Context
    priv, pub = _blobs()
    k = W.parse_client_key(priv)
    lfpk = W.compact_public_key_to_lfpk(pub)
    ctx = Context(seed=bytes(range(32)), params_id=2)
    try:
        ctx.import_client_keys_raw(k["lwe_sk"], k["glwe_sk"])
        ctx.keygen_server()
        ctx.set_pke_key(k["pke_sk"])
        a = ctx.cast_from_pke(ctx.encrypt_public(lfpk, 0xBEEF, 16))
        b = ctx.cast_from_pke(ctx.encrypt_public(lfpk, 0x1234, 16))
        assert ctx.decrypt(a) == 0xBEEF and ctx.decrypt(b) == 0x1234            # now under the compute key
        assert ctx.decrypt(ctx.evaluate("add", 16, a, b)) == (0xBEEF + 0x1234) & 0xFFFF
        assert ctx.decrypt(ctx.evaluate("lt", 16, a, b)) == 0
        raw = ctx.encrypt_public(lfpk, 0xBEEF, 16)
        assert ctx.decrypt(raw) != 0xBEEF                                       # not decryptable before the cast
    finally:
        ctx.close()


def _extract_under_pke_secret(c0, bodies, pke_sk):
    """Phases of the LWE ciphertexts of a compact list under the public-key secret (numpy, no GPU):
    block i = coefficient i of c1 - c0 (*) rev(s)."""
    n = len(c0)
    out = []
    with np.errstate(over="ignore"):
        for i in range(len(bodies)):
            m = np.empty(n, dtype=np.uint64)
            m[:i + 1] = c0[i::-1]
            m[i + 1:] = np.uint64(0) - c0[n - 1:i:-1]
            out.append(int(np.uint64(bodies[i]) - (m[::-1] * pke_sk).sum(dtype=np.uint64)))
    return out


def test_compact_lists_decrypt_under_the_reference_keys_and_are_found_in_a_container():
    """A compact ciphertext list built under the reference's publicKey.bin decrypts, coefficient by
    coefficient, under the public-key secret stored in privateKey.bin (two radix blocks per
    ciphertext when packed); the list and radix ciphertexts are found inside a bincode-style
    container whatever the framing around their length-prefixed word vectors."""
    priv, pub = _blobs()
    k = W.parse_client_key(priv)
    mask, body = W.parse_compact_public_key(pub)
    vals, widths = [0xDEADBEEFCAFEF00D, 1, 200, 0xABCD], [64, 1, 8, 16]
    blocks = []
    for v, w in zip(vals, widths):
        blocks += [v & 1] if w == 1 else [(v >> (2 * i)) & 3 for i in range(w // 2)]
    for packed in (True, False):
        c0, bodies = W.build_compact_list(mask, body, vals, widths, bytes(range(32)), packed=packed)
        want = [blocks[i] + 4 * (blocks[i + 1] if i + 1 < len(blocks) else 0) for i in range(0, len(blocks), 2)] if packed else blocks
        assert len(bodies) == len(want)
        got = [((p + (1 << 58)) >> 59) & 15 for p in _extract_under_pke_secret(c0, bodies, k["pke_sk"])]
        assert got == want
        noise = [((p - (m << 59) + (1 << 63)) % (1 << 64)) - (1 << 63) for p, m in zip(_extract_under_pke_secret(c0, bodies, k["pke_sk"]), want)]
        assert max(abs(e) for e in noise) < 1 << 40
        # a container in the style of the key fixtures: header strings, version tags, the vector, trailing fields, proof bytes
        words = np.concatenate([c0, bodies]).astype("<u8")
        blob = (b"\x03" + bytes(7) + b"0.5" + bytes(4) + b"\x19" + bytes(7) + b"CompactCiphertextList" + bytes(9)
                + struct.pack("<Q", len(words)) + words.tobytes() + struct.pack("<IQQ", 0, 2049, len(bodies)) + bytes(range(200)))
        m2, b2 = W.parse_compact_list(blob)
        assert np.array_equal(m2, c0) and np.array_equal(b2, bodies)
    with pytest.raises(ValueError):
        W.parse_compact_list(bytes(5000))
    # radix ciphertext: blocks as length-prefixed vectors of 2049 words, metadata behind each
    rng = np.random.default_rng(3)
    blks = rng.integers(0, 1 << 64, size=(4, 2049), dtype=np.uint64)
    meta = struct.pack("<I", 0) + bytes(16) + struct.pack("<Q", 64) + struct.pack("<IQ", 0, 3) + struct.pack("<IQ", 0, 2) + bytes(24)
    blob = b"hdr" + bytes(20) + struct.pack("<Q", 4) + b"".join(struct.pack("<IQ", 1, 2049) + b.tobytes() + meta for b in blks)
    ct = W.parse_radix_ciphertext(blob, 8)
    assert ct[:16] == struct.pack("<IHHIBBH", 0x5443464C, 1, 8, 4, 2, 3, 0)
    assert np.array_equal(np.frombuffer(ct[16:], dtype=np.uint64).reshape(4, 2049), blks)
    with pytest.raises(ValueError):
        W.parse_radix_ciphertext(blob, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [True, False], ids=["packed", "one-block-per-ciphertext"])
def test_compact_input_list_is_expanded_cast_and_computed_on(packed):
    """The reference's input flow end to end (sdk/relayer/src/sdk/encrypt.ts:54-196 on the client;
    expand + cast on the coprocessor): values of mixed types in ONE compact list under the
    reference's public key -> ciphertexts under the compute key -> operators -> decryption with the
    reference's client key."""
    from fhe_amd.native import Context, LfheError
    priv, pub = _blobs()
    k = W.parse_client_key(priv)
    mask, body = W.parse_compact_public_key(pub)
    vals, widths = [0xDEADBEEFCAFEF00D, 1, 200, 0xABCD, 0x0123456789ABCDEF], [64, 1, 8, 16, 64]
    c0, bodies = W.build_compact_list(mask, body, vals, widths, bytes(range(1, 33)), packed=packed)
    ctx = Context(seed=bytes(range(32)), params_id=3)
    try:
        ctx.import_client_keys_raw(k["lwe_sk"], k["glwe_sk"])
        ctx.keygen_server()
        ctx.set_pke_key(k["pke_sk"])
        cts = ctx.expand_compact_list(c0, bodies, widths, packed=packed)
        assert [ctx.decrypt(c) for c in cts] == vals
        assert [len(c) for c in cts] == [ctx.ciphertext_size(w) for w in widths]
        assert ctx.decrypt(ctx.evaluate("add", 64, cts[0], cts[4])) == (vals[0] + vals[4]) & ((1 << 64) - 1)
        assert ctx.decrypt(ctx.evaluate("select", 16, cts[3], ctx.encrypt(7, 16), cts[1])) == 0xABCD
        with pytest.raises(LfheError):
            ctx.expand_compact_list(c0, bodies[:-1], widths, packed=packed)       # length does not match the types
    finally:
        ctx.close()


def test_ciphertext_blob_as_raw_parts_for_tfhers_constructors():
    """The blocks and metadata tfhe-rs' raw-parts constructors take, from a blob of this library;
    the same blocks wrapped as length-prefixed word vectors (how bincode writes them inside a
    tfhe-rs container) are found again by the container reader."""
    import struct
    import numpy as np
    from fhe_amd import tfhers_wire as tw
    rng = np.random.default_rng(5)
    words = rng.integers(0, 1 << 64, size=(4, 2049), dtype=np.uint64)
    blob = struct.pack("<IHHIBBH", 0x5443464C, 1, 8, 4, 2, 3, 0) + words.astype("<u8").tobytes()
    parts = tw.radix_ciphertext_raw_parts(blob)
    assert parts["bit_width"] == 8 and parts["degree"] == 3 and parts["noise_level"] == 2
    assert parts["message_modulus"] == parts["carry_modulus"] == 4 and parts["lwe_dimension"] == 2048
    assert all(np.array_equal(b, w) for b, w in zip(parts["blocks"], words))
    container = b"\x07" * 40 + b"".join(struct.pack("<Q", 2049) + b.astype("<u8").tobytes() + b"\x00" * 60
                                         for b in parts["blocks"])
    back = tw.parse_radix_ciphertext(container, 8)
    assert back[16:] == blob[16:]
    import pytest
    with pytest.raises(ValueError):
        tw.radix_ciphertext_raw_parts(blob[:-8])
