"""CPU restatement of the compression path (test infrastructure, like the rest of oracle/): packing
key switch into GLWE ciphertexts of dimension 4 x 256, modulus switch to 2^12, and decompression
by sample extraction + bootstrap under a key for the compression secret.

Follows the construction the reference's extension uses for its own ciphertexts
(ml/extensions/rust/src/compression.rs:150-188 key generation, :244-290 packing key switch +
CompressedModulusSwitchedGlweCiphertext::compress, :110-128 extract) in the form tfhe-rs gives it
for radix blocks: the message is first moved up by the message modulus, the decompression table moves
it back.  Shape: tfhe-rs COMP_PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 (not in the reference
tree).  PARITY UNPINNED against the reference: it holds no compressed ciphertexts."""
from __future__ import annotations

import numpy as np

from .oracle import Oracle, Params, Rng, lib

KC, NC, BASE_LOG, LEVELS, STORAGE_LOG, NOISE_LOG2 = 4, 256, 4, 3, 12, 43
N = 2048
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def gen_comp_sk(rng: Rng) -> np.ndarray:
    return (rng.u64s(KC * NC) & np.uint64(1)).astype(np.uint64)


def _negacyclic_add(body: np.ndarray, a: np.ndarray, s: np.ndarray) -> None:
    """body[r] += a[r] (*) s for every row r; s binary of length NC."""
    with np.errstate(over="ignore"):
        for i in np.nonzero(s)[0]:
            i = int(i)
            body[:, i:] += a[:, :NC - i]
            if i:
                body[:, :i] -= a[:, NC - i:]


def gen_packing_ksk(glwe_sk: np.ndarray, comp_sk: np.ndarray, rng: Rng) -> np.ndarray:
    """[N * LEVELS][(KC + 1) * NC]: row (i, lv) encrypts the constant s_i 2^(64 - BASE_LOG (lv + 1))."""
    rows = N * LEVELS
    out = np.zeros((rows, (KC + 1) * NC), dtype=np.uint64)
    out[:, :KC * NC] = rng.u64s(rows * KC * NC).reshape(rows, KC * NC)
    bound = 1 << NOISE_LOG2
    e = (rng.u64s(rows * NC) % np.uint64(2 * bound + 1)).astype(np.int64) - bound
    body = e.astype(np.uint64).reshape(rows, NC)
    for p in range(KC):
        _negacyclic_add(body, out[:, p * NC:(p + 1) * NC], comp_sk[p * NC:(p + 1) * NC])
    with np.errstate(over="ignore"):
        for lv in range(LEVELS):
            body[lv::LEVELS, 0] += glwe_sk.astype(np.uint64) << np.uint64(64 - BASE_LOG * (lv + 1))
    out[:, KC * NC:] = body
    return out


def digits(x: np.ndarray) -> np.ndarray:
    """Signed base-2^BASE_LOG digits of the top BASE_LOG * LEVELS bits (rounded): [..., LEVELS], level 0 first."""
    rep = BASE_LOG * LEVELS
    state = (((x >> np.uint64(64 - rep - 1)) + np.uint64(1)) >> np.uint64(1)) & np.uint64((1 << rep) - 1)
    state = state.astype(np.int64)
    d = np.zeros(x.shape + (LEVELS,), dtype=np.int64)
    for lv in range(LEVELS - 1, -1, -1):
        o = state & ((1 << BASE_LOG) - 1)
        state >>= BASE_LOG
        carry = (((o - 1) | state) & o) >> (BASE_LOG - 1)
        state += carry
        d[..., lv] = o - (carry << BASE_LOG)
    return d


def pack(blocks: np.ndarray, pksk: np.ndarray, scale: int) -> np.ndarray:
    """blocks [count][N + 1] -> stored GLWEs [groups][(KC + 1) * NC] of STORAGE_LOG-bit words."""
    count = blocks.shape[0]
    groups = (count + NC - 1) // NC
    out = np.zeros((groups, (KC + 1) * NC), dtype=np.uint16)
    with np.errstate(over="ignore"):
        x = blocks * np.uint64(scale)
        D = digits(x[:, :N]).reshape(count, N * LEVELS).astype(np.uint64)       # two's complement
        prod = D @ pksk                                                            # wraps modulo 2^64
        for g in range(groups):
            cnt = min(NC, count - g * NC)
            acc = np.zeros((KC + 1, NC), dtype=np.uint64)
            for j in range(cnt):
                row = (np.uint64(0) - prod[g * NC + j]).reshape(KC + 1, NC)       # the key switch subtracts
                row[KC, 0] += x[g * NC + j, N]
                # X^j * row
                acc[:, j:] += row[:, :NC - j]
                if j:
                    acc[:, :j] -= row[:, NC - j:]
            r = ((acc >> np.uint64(64 - STORAGE_LOG - 1)) + np.uint64(1)) >> np.uint64(1)
            out[g] = (r & np.uint64((1 << STORAGE_LOG) - 1)).astype(np.uint16).reshape(-1)
    return out


def pack_bits(v: np.ndarray, bits: int) -> bytes:
    acc, have, out = 0, 0, bytearray()
    for w in v.tolist():
        acc |= w << have
        have += bits
        while have >= 8:
            out.append(acc & 0xFF)
            acc >>= 8
            have -= 8
    if have:
        out.append(acc & 0xFF)
    return bytes(out)


def extract(glwe: np.ndarray, count: int) -> np.ndarray:
    """Sample extraction of coefficient j of GLWE j // NC for every block j, scaled back to 2^64."""
    small = np.zeros((count, KC * NC + 1), dtype=np.uint64)
    sh = np.uint64(64 - STORAGE_LOG)
    with np.errstate(over="ignore"):
        for ct in range(count):
            G = glwe[ct // NC].astype(np.uint64).reshape(KC + 1, NC) << sh
            j = ct % NC
            for p in range(KC):
                a = G[p]
                small[ct, p * NC:p * NC + j + 1] = a[j::-1]
                if j + 1 < NC:
                    small[ct, p * NC + j + 1:(p + 1) * NC] = np.uint64(0) - a[:j:-1]
            small[ct, KC * NC] = G[KC, j]
    return small


class Decompressor:
    """Bootstrap of extracted blocks under a key for the compression secret (an Oracle of LWE
    dimension KC * NC, key of the given flavour: 'q64', 'q64_grouped', 'q64_grouped3', 'prime', 'grouped' or 'grouped3')."""

    def __init__(self, flavour: str, comp_sk: np.ndarray, glwe_sk: np.ndarray, seed: bytes):
        self.flavour = flavour
        self.orc = Oracle(Params(lwe_dim=KC * NC))
        if flavour == "q64":
            rng = Rng(seed, 3)
            L = lib()
            import ctypes as C
            self.bsk = np.zeros(L.orc_bsk_len(C.byref(self.orc._cp)), dtype=np.uint64)
            L.orc_gen_bsk(self.orc._ctx, comp_sk, glwe_sk, rng.ref, self.bsk)
            self.pb = self.orc.prepare_bsk(self.bsk)
        elif flavour in ("q64_grouped", "q64_grouped3"):
            bits = 3 if flavour == "q64_grouped3" else 2
            self.bsk = self.orc.gen_bsk_grouped(comp_sk, glwe_sk, seed, bits)
            self.pb = self.orc.prepare_bsk_grouped(self.bsk, bits)
        elif flavour == "prime":
            self.bsk = self.orc.gen_bsk_qp(comp_sk, glwe_sk, seed)
            self.pb = self.orc.prepare_bsk_qp(self.bsk)
        else:
            bits = 3 if flavour == "grouped3" else 2
            self.bsk = self.orc.gen_bsk_qp_grouped(comp_sk, glwe_sk, seed, bits)
            self.pb = self.orc.prepare_bsk_qp_grouped(self.bsk, bits)

    def table(self, message_modulus: int = 4) -> np.ndarray:
        return self.orc.make_lut(lambda x: ((x + message_modulus // 2) // message_modulus) % message_modulus)

    def bootstrap(self, small: np.ndarray, lut: np.ndarray) -> np.ndarray:
        o = self.orc
        if self.flavour.startswith("q64"):
            return o.sample_extract(o.blind_rotate(self.pb, small, lut))
        acc = o.blind_rotate_qp(self.pb, small, lut)
        q = o.qp
        a, b = acc[:N], acc[N:]
        out = np.zeros(N + 1, dtype=np.uint64)
        L = lib()
        out[0] = L.orc_from_qp(int(a[0]))
        for j in range(1, N):
            v = int(a[N - j])
            out[j] = L.orc_from_qp(q - v if v else 0)
        out[N] = L.orc_from_qp(int(b[0]))
        return out
