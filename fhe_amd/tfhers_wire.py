"""Reading the key blobs the reference's SDK produces (tfhe-rs `safe_serialize`, node-tfhe
1.4.0-alpha.3: sdk/relayer/generateKeys.js:20-66 writes privateKey.bin / publicKey.bin).

Not a general bincode reader: the blobs are walked for their length-prefixed u64 vectors, which is
enough to lift out the secret keys and the compact public key; the numeric parameters sit at fixed
positions behind the LWE secret key (layout observed in the reference's own fixtures,
sdk/relayer/src/test/keys/).  Emitting these formats is not implemented.
"""
from __future__ import annotations

import struct

import numpy as np


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
