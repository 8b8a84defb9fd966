"""Reading and writing the key blobs of the reference's SDK (tfhe-rs `safe_serialize`, node-tfhe
1.4.0-alpha.3: sdk/relayer/generateKeys.js:20-66 writes privateKey.bin / publicKey.bin).

Not a general bincode reader: the blobs are walked for their length-prefixed u64 vectors, which is
enough to lift out the secret keys and the compact public key; the numeric parameters sit at fixed
positions behind the LWE secret key (layout observed in the reference's own fixtures,
sdk/relayer/src/test/keys/).

Writing works for that one parameter set (PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 with its
dedicated public-key parameters): the framing around the key vectors -- serialization header,
version tags, parameter structs -- is the fixtures' own (extracted as data by
tools/extract_wire_framing.py), the vectors are the caller's.  Re-emitting the fixtures' vectors
reproduces the fixtures byte for byte (tests/test_reference_keys.py).
"""
from __future__ import annotations

import base64
import os
import struct

import numpy as np

# framing of high_level_api::ClientKey: before the GLWE key, between GLWE and LWE key, between the
# LWE key and the public-key-encryption secret key (the compute parameters), after it (its parameters)
CLIENT_KEY_FRAMING = (
    "AwAAAAAAAAAwLjUAAAAAAwAAAAAAAAAwLjEZAAAAAAAAAGhpZ2hfbGV2ZWxfYXBpOjpDbGllbnRLZXkBAAAAAwAAAAAAAAAAAAAAAAAAAAAIAAAAAAAA",
    "AAAAAAAIAAAAAAAAAAAAAJYDAAAAAAAA",
    "AAAAAAAAAAAAAAAAAAAAAAAAAAABAAAAAAAAAJYDAAAAAAAAAAAAAAEAAAAAAAAAAAAAAAAIAAAAAAAAAAAAAAEAAAAAAAAALQAAAAAAAAABAAAAAAAAABEAAAAAAAAAFwAAAAAAAAAAAAAAAQAAAAAAAAAAAAAABAAAAAAAAAAAAAAABAAAAAAAAAAAAAAABAAAAAAAAAAAAAAABAAAAAAAAAAAAAAABQAAAAAAAADYppQh5CRgwAAAAAAAAAAAAAAAAAAAAAAAAAAAQAAAAAAAAAAAAAAAAAAAAAEAAAAAAAAAAKkFAAAAAAAAAAAAAAAAAAAAAJBDAAAAAGEQK5kVXCpAAAAAAM823pD6ppE+AQAAAAAAAAAAAAAAAAAIAAAAAAAA",
    "AQAAAAAAAAAACAAAAAAAAAAAAAABAAAAAAAAABEAAAAAAAAABAAAAAAAAAAAAAAABAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAEAAAAAAAAAAAAAAAAAAAAAAAAAAAgAAAAAAAAAAAAAABAAAAAAAAAAAAAAABAAAAAAAAAAAAAAAAQAAAAAAAAAAAAAAAAAAAAA=",
)
# framing of high_level_api::CompactPublicKey: before and after the mask||body vector
COMPACT_PUBLIC_KEY_FRAMING = (
    "AwAAAAAAAAAwLjUAAAAAAwAAAAAAAAAwLjEgAAAAAAAAAGhpZ2hfbGV2ZWxfYXBpOjpDb21wYWN0UHVibGljS2V5AQAAAAAAAAAAAAAAAQAAAAAAAAAAAAAAABAAAAAAAAA=",
    "AAAAAAAIAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAABAAAAAAAAAAAEAAAAAAAAAAAgAAAAAAAAAAAAAAQAAAAAAAAARAAAAAAAAAAQAAAAAAAAAAAAAAAQAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAABAAAAAAAAAAAAAAAAAAAAAAAAAAAIAAAAAAAAAAAAAAAAAAAA=",
)
CLIENT_KEY_DIMS = (2048, 918, 2048)      # GLWE key, LWE key, public-key-encryption secret key


def _u64_vectors(blob: bytes, accept, min_len: int = 512):
    """(offset of data, array) of every `u64 length | length * u64` run whose content passes
    `accept` (a byte string may look like a length prefix by accident, so content decides)."""
    out = []
    off = 0
    n = len(blob)
    while off + 8 <= n:
        ln = int.from_bytes(blob[off:off + 8], "little")
        if min_len <= ln <= 1 << 20 and off + 8 + 8 * ln <= n:
            v = np.frombuffer(blob, dtype="<u8", count=ln, offset=off + 8)
            if accept(v):
                out.append((off + 8, v.copy()))
                off += 8 + 8 * ln
                continue
        off += 1
    return out


def parse_client_key(blob: bytes) -> dict:
    """{'glwe_sk', 'lwe_sk', 'pke_sk': uint64 arrays of 0/1, 'params': dict} of a serialized
    high_level_api::ClientKey of the KS_PBS parameter family."""
    vecs = _u64_vectors(blob, lambda v: int(v.max(initial=0)) <= 1 and 0 < int(v.sum()) < len(v))
    if len(vecs) < 2:
        raise ValueError("no binary secret keys found in the blob")
    (o_glwe, glwe), (o_lwe, lwe) = vecs[0], vecs[1]
    base = o_lwe + 8 * len(lwe)            # the parameter struct follows the LWE secret key
    u64 = lambda rel: int.from_bytes(blob[base + rel:base + rel + 8], "little")
    u32 = lambda rel: int.from_bytes(blob[base + rel:base + rel + 4], "little")
    f64 = lambda rel: struct.unpack("<d", blob[base + rel:base + rel + 8])[0]
    params = {
        "lwe_dim": u64(28), "glwe_dim": u64(40), "poly_size": u64(52),
        "lwe_noise_log2": u32(72), "glwe_noise_log2": u32(88),
        "pbs_base_log": u64(96), "pbs_level": u64(108), "ks_base_log": u64(120), "ks_level": u64(132),
        "message_modulus": u64(144), "carry_modulus": u64(156), "max_noise_level": u64(168),
        "log2_p_fail": f64(176),
        "ms_zeros_count": u32(229), "ms_bound": f64(241), "ms_r_sigma_factor": f64(253),
        "ms_input_variance": f64(265),
    }
    if params["lwe_dim"] != len(lwe) or params["glwe_dim"] * params["poly_size"] != len(glwe):
        raise ValueError("parameter block does not match the key sizes: unknown layout")
    out = {"glwe_sk": glwe, "lwe_sk": lwe, "params": params, "pke_sk": None}
    if len(vecs) > 2:
        out["pke_sk"] = vecs[2][1]
    return out


def parse_compact_public_key(blob: bytes):
    """(mask[N], body[N]) of a serialized high_level_api::CompactPublicKey."""
    # mask and body are uniform 64-bit words: a real key vector has both halves of the word populated
    vecs = _u64_vectors(blob, lambda v: len(v) % 2 == 0 and int((v >> np.uint64(48)).astype(bool).mean() > 0.9),
                        min_len=1024)
    if not vecs:
        raise ValueError("no key vector found in the blob")
    v = max(vecs, key=lambda ov: len(ov[1]))[1]
    return v[: len(v) // 2].copy(), v[len(v) // 2:].copy()


def compact_public_key_to_lfpk(blob: bytes, noise_log2: int = 17) -> bytes:
    """The library's public-key blob for a tfhe-rs compact public key.  tfhe-rs keys satisfy
    body = mask (*) rev(s) + e (the reversed secret), flagged in the header so that
    lfhe_encrypt_public emits blocks that decrypt under s itself."""
    mask, body = parse_compact_public_key(blob)
    if len(mask) != 2048:
        raise ValueError("only polynomial size 2048 is supported")
    hdr = struct.pack("<IIII", 0x4B50464C, 2048, noise_log2, 1)
    return hdr + mask.astype("<u8").tobytes() + body.astype("<u8").tobytes()


def emit_client_key(glwe_sk, lwe_sk, pke_sk) -> bytes:
    """Serialized high_level_api::ClientKey of the reference's parameter set holding these binary keys."""
    keys = [np.ascontiguousarray(k, dtype="<u8") for k in (glwe_sk, lwe_sk, pke_sk)]
    for k, n in zip(keys, CLIENT_KEY_DIMS):
        if k.shape != (n,) or int(k.max(initial=0)) > 1:
            raise ValueError("keys must be binary vectors of 2048, 918 and 2048 words")
    out = bytearray()
    for frame, k in zip(CLIENT_KEY_FRAMING, keys + [None]):
        out += base64.b64decode(frame)
        if k is not None:
            out += k.tobytes()
    return bytes(out)


def emit_compact_public_key(mask, body) -> bytes:
    """Serialized high_level_api::CompactPublicKey (tfhe-rs convention: body = mask (*) rev(s) + e)."""
    v = np.concatenate([np.ascontiguousarray(mask, dtype="<u8"), np.ascontiguousarray(body, dtype="<u8")])
    if v.shape != (4096,):
        raise ValueError("mask and body must have 2048 words each")
    return base64.b64decode(COMPACT_PUBLIC_KEY_FRAMING[0]) + v.tobytes() + base64.b64decode(COMPACT_PUBLIC_KEY_FRAMING[1])


class _Draws:
    """Where the builders below take their randomness from.  seed = None: the operating system's
    generator (os.urandom), which is what anything that protects real data must use.  A seed: numpy's
    PCG64, reproducible and NOT cryptographic -- for known-answer vectors and tests only."""

    def __init__(self, seed: bytes | None):
        self._rng = None if seed is None else np.random.default_rng(list(seed))

    def words(self, n: int) -> np.ndarray:
        if self._rng is None:
            return np.frombuffer(os.urandom(8 * n), dtype="<u8").astype(np.uint64)
        return self._rng.integers(0, 1 << 64, size=n, dtype=np.uint64)

    def bits(self, n: int) -> np.ndarray:
        return self.words(n) & np.uint64(1)

    def tuniform(self, n: int, log2_bound: int) -> np.ndarray:
        """TUniform(b) as tfhe-rs defines it: [-2^b, 2^b], the end points at half weight."""
        v = self.words(n) & np.uint64((1 << (log2_bound + 2)) - 1)
        with np.errstate(over="ignore"):
            return ((v + np.uint64(1)) >> np.uint64(1)) - np.uint64(1 << log2_bound)


def make_compact_public_key(pke_sk, seed: bytes | None = None, noise_log2: int = 17):
    """(mask, body) of a fresh compact public key under the binary secret `pke_sk` in the convention of
    tfhe-rs: body = mask (*) rev(s) + e over Z_2^64[X]/(X^N+1), rev(s)_i = s_{N-1-i}, e from
    TUniform(noise_log2).  seed: None draws from the operating system; a seed gives reproducible,
    non-cryptographic randomness (known-answer vectors only)."""
    s = np.ascontiguousarray(pke_sk, dtype=np.uint64)
    n = len(s)
    rng = _Draws(seed)
    mask = rng.words(n)
    e = rng.tuniform(n, noise_log2)
    body = e.copy()
    rev = s[::-1]
    with np.errstate(over="ignore"):
        for i in np.nonzero(rev)[0]:          # body += X^i * mask (negacyclic)
            i = int(i)
            body[i:] += mask[:n - i]
            body[:i] -= mask[n - i:]
    return mask, body


# ---- ciphertexts ------------------------------------------------------------------------------------
# No ciphertext fixture exists in the reference tree (sdk/relayer/src/test holds keys only), so the
# version tags and field order of tfhe-rs' serialized ciphertext types cannot be pinned here.  What
# IS certain is how bincode writes the payloads: a `Vec<u64>` is its length as a u64 followed by the
# words.  The readers below therefore find the ciphertext words the same way the key readers find
# the keys -- by walking the blob for length-prefixed runs of the right length whose content looks
# like ciphertext words -- and ignore the framing around them.  Nothing here writes tfhe-rs
# ciphertext blobs: a writer with guessed version tags would produce files tfhe-rs refuses.  The way
# out of this library into tfhe-rs objects is radix_ciphertext_raw_parts: the words and metadata
# that tfhe-rs' own raw-parts constructors take.
_LWE_WORDS = 2049


def _looks_uniform(v) -> bool:
    return bool((v >> np.uint64(48)).astype(bool).mean() > 0.9)


def parse_compact_list(blob: bytes):
    """(mask[2048], bodies[n]) of a serialized compact ciphertext list holding n <= 2048 LWE
    ciphertexts under a 2048-dimensional public-key secret (tfhe-rs LweCompactCiphertextList: one
    vector of 2048 + n words).  Works on the proven and the plain list alike: the proof bytes are
    not 64-bit ciphertext words and are skipped."""
    vecs = [(o, v) for o, v in _u64_vectors(blob, _looks_uniform, min_len=2049) if 2048 < len(v) <= 4096]
    if not vecs:
        raise ValueError("no compact ciphertext list found in the blob")
    v = max(vecs, key=lambda ov: len(ov[1]))[1]
    return v[:2048].copy(), v[2048:].copy()


def parse_radix_ciphertext(blob: bytes, bit_width: int | None = None) -> bytes:
    """A serialized tfhe-rs radix ciphertext (FheUintN / RadixCiphertext: a vector of shortint
    blocks, each holding its LWE words as a length-prefixed vector of 2049) as this library's
    ciphertext blob.  bit_width: checked against the number of blocks found (two bits per block;
    one block: ebool) when given.  Degree and noise level are read from the two versioned u64
    fields that follow a block's modulus when they look plausible, else taken as 3 and 1."""
    vecs = _u64_vectors(blob, lambda v: len(v) == _LWE_WORDS and _looks_uniform(v[:2048]), min_len=_LWE_WORDS)
    vecs = [(o, v) for o, v in vecs if len(v) == _LWE_WORDS]
    if not vecs:
        raise ValueError("no LWE blocks of 2049 words found in the blob")
    nb = len(vecs)
    width = 1 if nb == 1 and bit_width in (None, 1) else 2 * nb
    if bit_width is not None and bit_width != width:
        raise ValueError(f"{nb} blocks do not make a {bit_width}-bit ciphertext")
    degree, noise = (1 if width == 1 else 3), 1
    end = vecs[0][0] + 8 * _LWE_WORDS        # behind the words: modulus (u32 tag, u128, u64), then degree and noise level
    if end + 52 <= len(blob):
        d = int.from_bytes(blob[end + 32:end + 40], "little")
        z = int.from_bytes(blob[end + 44:end + 52], "little")
        if 0 < d <= 3 and 0 < z <= 2:
            degree, noise = d, z
    out = bytearray(struct.pack("<IHHIBBH", 0x5443464C, 1, width, nb, noise, degree, 0))
    for _, v in vecs:
        out += v.astype("<u8").tobytes()
    return bytes(out)


def radix_ciphertext_raw_parts(ct: bytes) -> dict:
    """This library's ciphertext blob taken apart into what tfhe-rs' raw-parts constructors take
    (`LweCiphertext::from_container(words, CiphertextModulus::new_native())`,
    `shortint::Ciphertext::new(lwe, Degree::new(degree), NoiseLevel, MessageModulus(4), CarryModulus(4), ..)`,
    `RadixCiphertext::from(blocks)`): the route from a result of this library into tfhe-rs objects
    that does not depend on the version tags of tfhe-rs' serialized ciphertext types, which no
    fixture in the reference pins.  Blocks are little-endian in significance, 2049 words each (mask
    of 2048 under the GLWE secret key, body last), native modulus 2^64, KeyswitchBootstrap order."""
    if len(ct) < 16:
        raise ValueError("short ciphertext blob")
    magic, version, width, nb, noise, degree, _ = struct.unpack("<IHHIBBH", ct[:16])
    if magic != 0x5443464C or version != 1 or len(ct) != 16 + nb * _LWE_WORDS * 8:
        raise ValueError("not a ciphertext blob of this library")
    words = np.frombuffer(ct, dtype="<u8", offset=16).reshape(nb, _LWE_WORDS)
    return {"bit_width": width, "blocks": [w.copy() for w in words], "degree": degree, "noise_level": noise,
            "message_modulus": 4, "carry_modulus": 4,     # an ebool is a block of the same space holding 0 or 1
            "ciphertext_modulus_log2": 64, "lwe_dimension": _LWE_WORDS - 1, "pbs_order": "KeyswitchBootstrap"}


def build_compact_list(pk_mask, pk_body, values: list, widths: list, seed: bytes | None = None, packed: bool = True,
                       noise_log2: int = 17):
    """What a client's input builder does (sdk/relayer/src/sdk/encrypt.ts:54-196, without the proof):
    all values' radix blocks in order, two per ciphertext when packed (low + 4 * high), encrypted
    as ONE compact list under the compact public key (tfhe-rs convention: key body = mask (*) rev(s)
    + e): u binary, c0 = a (*) u + e1, c1 = b (*) u + e2 + M.  Returns (mask c0[2048], bodies[n]).
    seed: None draws u and the noise from the operating system; a seed gives reproducible,
    non-cryptographic randomness (known-answer vectors only).  Parity unpinned: the reference holds
    no ciphertext fixture; the convention follows from its key fixtures."""
    n = 2048
    blocks = []
    for v, w in zip(values, widths):
        if w == 1:
            blocks.append(int(v) & 1)
        else:
            blocks += [(int(v) >> (2 * i)) & 3 for i in range(w // 2)]
    if packed:
        blocks = [blocks[i] + 4 * (blocks[i + 1] if i + 1 < len(blocks) else 0) for i in range(0, len(blocks), 2)]
    if not 0 < len(blocks) <= n:
        raise ValueError("a compact list holds 1 .. 2048 ciphertexts")
    rng = _Draws(seed)
    u = rng.bits(n)
    tu = lambda: rng.tuniform(n, noise_log2)
    a, b = np.ascontiguousarray(pk_mask, dtype=np.uint64), np.ascontiguousarray(pk_body, dtype=np.uint64)
    c0, c1 = tu(), tu()
    with np.errstate(over="ignore"):
        for i in np.nonzero(u)[0]:            # += X^i * polynomial (negacyclic)
            i = int(i)
            c0[i:] += a[:n - i]
            c0[:i] -= a[n - i:]
            c1[i:] += b[:n - i]
            c1[:i] -= b[n - i:]
        msg = np.array(blocks, dtype=np.uint64) << np.uint64(59)
        c1[:len(blocks)] += msg
    return c0, c1[:len(blocks)].copy()
