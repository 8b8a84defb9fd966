"""ctypes bindings for the CPU oracle (oracle/tfhe_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg.  The product package (fhe_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with the Makefile next to this file (gcc, seconds)."""
    src = os.path.join(_HERE, "tfhe_oracle.c")
    hdr = os.path.join(_HERE, "tfhe_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "lwe_dim", "glwe_dim", "poly_size", "pbs_base_log", "pbs_level", "ks_base_log", "ks_level",
        "message_modulus", "carry_modulus", "lwe_noise_log2", "glwe_noise_log2")]


class _MsParams(C.Structure):
    _fields_ = [("zeros_count", C.c_uint32), ("bound_log2", C.c_uint32), ("r_sigma_factor", C.c_double),
                ("input_variance", C.c_double)]


class _Rng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 8), ("counter", C.c_uint64), ("stream", C.c_uint64),
                ("buf", C.c_uint32 * 16), ("pos", C.c_uint32)]


@dataclass(frozen=True)
class Params:
    lwe_dim: int = 918
    glwe_dim: int = 1
    poly_size: int = 2048
    pbs_base_log: int = 23
    pbs_level: int = 1
    ks_base_log: int = 4
    ks_level: int = 4
    message_modulus: int = 4
    carry_modulus: int = 4
    lwe_noise_log2: int = 45
    glwe_noise_log2: int = 17

    @property
    def big_dim(self) -> int:
        return self.glwe_dim * self.poly_size

    @property
    def delta(self) -> int:
        return (1 << 63) // (self.message_modulus * self.carry_modulus)


_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, u64, u32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
    sig = {
        "orc_rng_init": (None, [C.POINTER(_Rng), C.c_char_p, u64]),
        "orc_rng_u64": (u64, [C.POINTER(_Rng)]),
        "orc_rng_tuniform": (u64, [C.POINTER(_Rng), u32]),
        "orc_ctx_new": (vp, [C.POINTER(_Params)]),
        "orc_ctx_free": (None, [vp]),
        "orc_negacyclic_mul_schoolbook": (None, [_u64p, _i64p, _u64p, u32]),
        "orc_negacyclic_mul_ntt": (None, [vp, _u64p, _i64p, _u64p]),
        "orc_monomial_mul": (None, [_u64p, _u64p, u32, u32]),
        "orc_ntt_definition": (None, [u64, u64, _u64p, _u64p, u32]),
        "orc_mulmod": (u64, [u64, u64, u64]),
        "orc_powmod": (u64, [u64, u64, u64]),
        "orc_decompose": (None, [u64, u32, u32, _i64p]),
        "orc_gen_binary_key": (None, [C.POINTER(_Rng), _u64p, sz]),
        "orc_lwe_encrypt": (None, [_u64p, sz, u64, u32, C.POINTER(_Rng), _u64p]),
        "orc_lwe_phase": (u64, [_u64p, sz, _u64p]),
        "orc_glwe_encrypt": (None, [vp, _u64p, vp, C.POINTER(_Rng), _u64p]),
        "orc_glwe_phase": (None, [vp, _u64p, _u64p, _u64p]),
        "orc_bsk_len": (sz, [C.POINTER(_Params)]),
        "orc_gen_bsk": (None, [vp, _u64p, _u64p, C.POINTER(_Rng), _u64p]),
        "orc_ksk_len": (sz, [C.POINTER(_Params)]),
        "orc_gen_ksk": (None, [vp, _u64p, _u64p, C.POINTER(_Rng), _u64p]),
        "orc_external_product": (None, [vp, _u64p, _u64p, _u64p]),
        "orc_cmux_rotate": (None, [vp, _u64p, u32, _u64p]),
        "orc_modswitch": (u32, [u64, u32]),
        "orc_keyswitch": (None, [vp, _u64p, _u64p, _u64p]),
        "orc_default_ms_params": (None, [C.POINTER(_MsParams)]),
        "orc_gen_ms_zeros": (None, [vp, _u64p, C.POINTER(_Rng), u32, _u64p]),
        "orc_ms_estimate": (u64, [vp, C.POINTER(_MsParams), _u64p]),
        "orc_ms_noise_reduce": (C.c_int32, [vp, C.POINTER(_MsParams), _u64p, _u64p]),
        "orc_ctx_set_ms": (None, [vp, C.POINTER(_MsParams), vp]),
        "orc_blind_rotate": (None, [vp, _u64p, _u64p, _u64p, _u64p]),
        "orc_sample_extract": (None, [vp, _u64p, _u64p]),
        "orc_pbs": (None, [vp, _u64p, _u64p, _u64p, _u64p, _u64p]),
        "orc_bsk_prepare": (vp, [vp, _u64p]),
        "orc_bsk_prepared_free": (None, [vp]),
        "orc_blind_rotate_prepared": (None, [vp, vp, _u64p, _u64p, _u64p]),
        "orc_pbs_prepared": (None, [vp, _u64p, vp, _u64p, _u64p, _u64p]),
        "orc_pbs_batch_prepared": (None, [vp, _u64p, vp, _u64p, _u64p, vp, _u64p, sz, C.c_int]),
        "orc_make_lut": (None, [vp, _u64p, _u64p]),
        "orc_qp": (u64, []),
        "orc_qp_gadget": (u64, [vp]),
        "orc_qp_noise_bound": (u32, [vp]),
        "orc_to_qp": (u64, [u64]),
        "orc_from_qp": (u64, [u64]),
        "orc_qp_digit": (C.c_int64, [vp, u64]),
        "orc_gen_bsk_qp": (None, [vp, _u64p, _u64p, C.POINTER(_Rng), _u64p]),
        "orc_external_product_qp": (None, [vp, _u64p, _u64p, _u64p]),
        "orc_bsk_qp_prepare": (vp, [vp, _u64p]),
        "orc_bsk_qp_grouped_count": (sz, [C.POINTER(_Params)]),
        "orc_gen_bsk_qp_grouped": (None, [vp, _u64p, _u64p, C.POINTER(_Rng), _u64p]),
        "orc_bsk_qp_grouped_prepare": (vp, [vp, _u64p]),
        "orc_bsk_qp_grouped_count_n": (sz, [C.POINTER(_Params), u32]),
        "orc_gen_bsk_qp_grouped_n": (None, [vp, _u64p, _u64p, C.POINTER(_Rng), u32, _u64p]),
        "orc_bsk_qp_grouped_prepare_n": (vp, [vp, _u64p, u32]),
        "orc_bsk_grouped_count": (sz, [C.POINTER(_Params)]),
        "orc_gen_bsk_grouped": (None, [vp, _u64p, _u64p, C.POINTER(_Rng), _u64p]),
        "orc_bsk_grouped_prepare": (vp, [vp, _u64p]),
        "orc_bsk_grouped_count_n": (sz, [C.POINTER(_Params), u32]),
        "orc_gen_bsk_grouped_n": (None, [vp, _u64p, _u64p, C.POINTER(_Rng), u32, _u64p]),
        "orc_bsk_grouped_prepare_n": (vp, [vp, _u64p, u32]),
        "orc_group_step_n": (None, [vp, _u64p, u32, np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS"), _u64p]),
        "orc_group_step": (None, [vp, _u64p, u32, u32, _u64p]),
        "orc_blind_rotate_qp": (None, [vp, vp, _u64p, _u64p, _u64p]),
        "orc_pbs_qp": (None, [vp, _u64p, vp, _u64p, _u64p, _u64p]),
        "orc_pbs_batch_qp": (None, [vp, _u64p, vp, _u64p, _u64p, vp, _u64p, sz, C.c_int]),
        "orc_encode": (u64, [vp, u64]),
        "orc_decode": (u64, [vp, u64]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


class Rng:
    def __init__(self, seed: bytes, stream: int = 0):
        assert len(seed) == 32
        self._r = _Rng()
        lib().orc_rng_init(C.byref(self._r), seed, stream)

    def u64(self) -> int:
        return lib().orc_rng_u64(C.byref(self._r))

    def u64s(self, n: int) -> np.ndarray:
        return np.array([self.u64() for _ in range(n)], dtype=np.uint64)

    def tuniform(self, b: int) -> int:
        return lib().orc_rng_tuniform(C.byref(self._r), b)

    @property
    def ref(self):
        return C.byref(self._r)


def seed_bytes(tag: int) -> bytes:
    return bytes((tag * 131 + i * 7 + 1) & 0xFF for i in range(32))


class Oracle:
    """One parameter set + the oracle's NTT tables."""

    def __init__(self, params: Params = Params()):
        self.p = params
        self._cp = _Params(**{k: getattr(params, k) for k in params.__dataclass_fields__})
        self._ctx = lib().orc_ctx_new(C.byref(self._cp))
        self.N = params.poly_size
        self.k = params.glwe_dim
        self.n = params.lwe_dim
        self.big = params.big_dim

    def __del__(self):
        try:
            lib().orc_ctx_free(self._ctx)
        except Exception:
            pass

    # -- polynomial arithmetic
    def mul_schoolbook(self, small: np.ndarray, b: np.ndarray) -> np.ndarray:
        out = np.zeros(self.N, dtype=np.uint64)
        lib().orc_negacyclic_mul_schoolbook(out, np.ascontiguousarray(small, dtype=np.int64),
                                            np.ascontiguousarray(b, dtype=np.uint64), self.N)
        return out

    def mul_ntt(self, small: np.ndarray, b: np.ndarray) -> np.ndarray:
        out = np.zeros(self.N, dtype=np.uint64)
        lib().orc_negacyclic_mul_ntt(self._ctx, out, np.ascontiguousarray(small, dtype=np.int64),
                                     np.ascontiguousarray(b, dtype=np.uint64))
        return out

    def monomial_mul(self, poly: np.ndarray, rot: int) -> np.ndarray:
        out = np.zeros(self.N, dtype=np.uint64)
        lib().orc_monomial_mul(out, np.ascontiguousarray(poly, dtype=np.uint64), rot % (2 * self.N), self.N)
        return out

    @staticmethod
    def ntt_definition(p: int, psi: int, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros_like(a)
        lib().orc_ntt_definition(p, psi, a, out, len(a))
        return out

    @staticmethod
    def decompose(x: int, base_log: int, levels: int) -> np.ndarray:
        d = np.zeros(levels, dtype=np.int64)
        lib().orc_decompose(x & (2**64 - 1), base_log, levels, d)
        return d

    # -- keys
    def gen_keys(self, seed: bytes):
        """(lwe_sk[n], glwe_sk[kN], bsk, ksk), all from one seed with fixed stream ids."""
        L = lib()
        r = Rng(seed, 1)
        lwe_sk = np.zeros(self.n, dtype=np.uint64)
        L.orc_gen_binary_key(r.ref, lwe_sk, self.n)
        r = Rng(seed, 2)
        glwe_sk = np.zeros(self.big, dtype=np.uint64)
        L.orc_gen_binary_key(r.ref, glwe_sk, self.big)
        bsk = np.zeros(L.orc_bsk_len(C.byref(self._cp)), dtype=np.uint64)
        L.orc_gen_bsk(self._ctx, lwe_sk, glwe_sk, Rng(seed, 3).ref, bsk)
        ksk = np.zeros(L.orc_ksk_len(C.byref(self._cp)), dtype=np.uint64)
        L.orc_gen_ksk(self._ctx, glwe_sk, lwe_sk, Rng(seed, 4).ref, ksk)
        return lwe_sk, glwe_sk, bsk, ksk

    # -- LWE
    def encode(self, m: int) -> int:
        return lib().orc_encode(self._ctx, m)

    def decode(self, phase: int) -> int:
        return lib().orc_decode(self._ctx, phase)

    def lwe_encrypt(self, sk: np.ndarray, plaintext: int, noise_log2: int, rng: Rng) -> np.ndarray:
        ct = np.zeros(len(sk) + 1, dtype=np.uint64)
        lib().orc_lwe_encrypt(sk, len(sk), plaintext & (2**64 - 1), noise_log2, rng.ref, ct)
        return ct

    def encrypt_block(self, glwe_sk: np.ndarray, m: int, rng: Rng) -> np.ndarray:
        return self.lwe_encrypt(glwe_sk, self.encode(m), self.p.glwe_noise_log2, rng)

    def lwe_phase(self, sk: np.ndarray, ct: np.ndarray) -> int:
        return lib().orc_lwe_phase(sk, len(sk), np.ascontiguousarray(ct, dtype=np.uint64))

    def decrypt_block(self, glwe_sk: np.ndarray, ct: np.ndarray) -> int:
        return self.decode(self.lwe_phase(glwe_sk, ct))

    # -- GLWE / GGSW
    def glwe_encrypt(self, glwe_sk: np.ndarray, pt, rng: Rng) -> np.ndarray:
        ct = np.zeros((self.k + 1) * self.N, dtype=np.uint64)
        ptr = None if pt is None else np.ascontiguousarray(pt, dtype=np.uint64).ctypes.data_as(C.c_void_p)
        lib().orc_glwe_encrypt(self._ctx, glwe_sk, ptr, rng.ref, ct)
        return ct

    def glwe_phase(self, glwe_sk: np.ndarray, ct: np.ndarray) -> np.ndarray:
        ph = np.zeros(self.N, dtype=np.uint64)
        lib().orc_glwe_phase(self._ctx, glwe_sk, np.ascontiguousarray(ct, dtype=np.uint64), ph)
        return ph

    def ggsw_words(self) -> int:
        return self.p.pbs_level * (self.k + 1) * (self.k + 1) * self.N

    def external_product(self, ggsw: np.ndarray, glwe: np.ndarray) -> np.ndarray:
        out = np.zeros((self.k + 1) * self.N, dtype=np.uint64)
        lib().orc_external_product(self._ctx, np.ascontiguousarray(ggsw), np.ascontiguousarray(glwe), out)
        return out

    def cmux_rotate(self, ggsw: np.ndarray, rot: int, acc: np.ndarray) -> np.ndarray:
        acc = np.array(acc, dtype=np.uint64, copy=True)
        lib().orc_cmux_rotate(self._ctx, np.ascontiguousarray(ggsw), rot, acc)
        return acc

    # -- bootstrap path
    def modswitch(self, x: int) -> int:
        return lib().orc_modswitch(x, self.N.bit_length())

    def keyswitch(self, ksk: np.ndarray, ct_big: np.ndarray) -> np.ndarray:
        out = np.zeros(self.n + 1, dtype=np.uint64)
        lib().orc_keyswitch(self._ctx, ksk, np.ascontiguousarray(ct_big, dtype=np.uint64), out)
        return out

    # -- modulus-switch noise reduction
    def ms_params(self, zeros_count: int | None = None) -> _MsParams:
        m = _MsParams()
        lib().orc_default_ms_params(C.byref(m))
        if zeros_count is not None:
            m.zeros_count = zeros_count
        return m

    def gen_ms_zeros(self, lwe_sk: np.ndarray, seed: bytes, count: int) -> np.ndarray:
        z = np.zeros((count, self.n + 1), dtype=np.uint64)
        lib().orc_gen_ms_zeros(self._ctx, lwe_sk, Rng(seed, 5).ref, count, z)
        return z

    def ms_estimate(self, small: np.ndarray, m: _MsParams | None = None) -> int:
        m = m or self.ms_params()
        return lib().orc_ms_estimate(self._ctx, C.byref(m), np.ascontiguousarray(small, dtype=np.uint64))

    def ms_noise_reduce(self, small: np.ndarray, zeros: np.ndarray, m: _MsParams | None = None):
        """(reduced ciphertext, index of the zero that was added or -1)."""
        m = m or self.ms_params(len(zeros))
        out = np.array(small, dtype=np.uint64, copy=True)
        pick = lib().orc_ms_noise_reduce(self._ctx, C.byref(m), np.ascontiguousarray(zeros, dtype=np.uint64), out)
        return out, pick

    def set_ms(self, zeros: np.ndarray | None, m: _MsParams | None = None):
        """Make pbs()/pbs_batch() include the noise-reduction step (None switches it off)."""
        if zeros is None:
            self._ms_keep = None
            lib().orc_ctx_set_ms(self._ctx, None, None)
            return
        z = np.ascontiguousarray(zeros, dtype=np.uint64)
        m = m or self.ms_params(len(z))
        self._ms_keep = (z, m)
        lib().orc_ctx_set_ms(self._ctx, C.byref(m), z.ctypes.data_as(C.c_void_p))

    def prepare_bsk(self, bsk: np.ndarray):
        return _PreparedBsk(self, bsk)

    # -- grouped key over 2^64: 2^bits GGSWs per group of `bits` (2 or 3) key bits; blind_rotate / pbs /
    # pbs_batch take the prepared key like the one-bit key's
    def gen_bsk_grouped(self, lwe_sk: np.ndarray, glwe_sk: np.ndarray, seed: bytes, bits: int = 2) -> np.ndarray:
        assert bits in (2, 3)
        count = lib().orc_bsk_grouped_count_n(C.byref(self._cp), bits)
        bsk = np.zeros(count * self.ggsw_words(), dtype=np.uint64)
        lib().orc_gen_bsk_grouped_n(self._ctx, lwe_sk, glwe_sk, Rng(seed, 8 if bits == 2 else 10).ref, bits, bsk)
        return bsk

    def prepare_bsk_grouped(self, bsk: np.ndarray, bits: int = 2):
        return _PreparedBsk(self, bsk, grouped64=True, bits=bits)

    def group_step(self, ggsws: np.ndarray, exponents, acc: np.ndarray) -> np.ndarray:
        """acc <- (sum over m of X^(sum of the exponents whose bit is set in m) GGSW_m) [.] acc over 2^64."""
        acc = np.array(acc, dtype=np.uint64, copy=True)
        a = np.array(exponents, dtype=np.uint32)
        lib().orc_group_step_n(self._ctx, np.ascontiguousarray(ggsws, dtype=np.uint64), len(a), a, acc)
        return acc

    def blind_rotate(self, pbsk, lwe_small: np.ndarray, lut: np.ndarray) -> np.ndarray:
        acc = np.zeros((self.k + 1) * self.N, dtype=np.uint64)
        lib().orc_blind_rotate_prepared(self._ctx, pbsk.h, np.ascontiguousarray(lwe_small),
                                        np.ascontiguousarray(lut), acc)
        return acc

    def sample_extract(self, glwe: np.ndarray) -> np.ndarray:
        out = np.zeros(self.big + 1, dtype=np.uint64)
        lib().orc_sample_extract(self._ctx, np.ascontiguousarray(glwe), out)
        return out

    def pbs(self, ksk: np.ndarray, pbsk, ct_big: np.ndarray, lut: np.ndarray) -> np.ndarray:
        out = np.zeros(self.big + 1, dtype=np.uint64)
        lib().orc_pbs_prepared(self._ctx, ksk, pbsk.h, np.ascontiguousarray(ct_big),
                               np.ascontiguousarray(lut), out)
        return out

    def pbs_batch(self, ksk, pbsk, cts: np.ndarray, luts: np.ndarray, lut_index=None,
                  nthreads: int = 1) -> np.ndarray:
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.big + 1)
        luts = np.ascontiguousarray(luts, dtype=np.uint64)
        out = np.zeros_like(cts)
        idx = None
        if lut_index is not None:
            idx_arr = np.ascontiguousarray(lut_index, dtype=np.uint32)
            idx = idx_arr.ctypes.data_as(C.c_void_p)
        lib().orc_pbs_batch_prepared(self._ctx, ksk, pbsk.h, cts, luts, idx, out, cts.shape[0], nthreads)
        return out

    # -- bootstrap over the prime ring modulus q' (bootstrapping key and accumulator in Z_q')
    @property
    def qp(self) -> int:
        return lib().orc_qp()

    def qp_gadget(self) -> int:
        return lib().orc_qp_gadget(self._ctx)

    def qp_noise_bound(self) -> int:
        return lib().orc_qp_noise_bound(self._ctx)

    def qp_digit(self, res: int) -> int:
        return lib().orc_qp_digit(self._ctx, res)

    def gen_bsk_qp(self, lwe_sk: np.ndarray, glwe_sk: np.ndarray, seed: bytes) -> np.ndarray:
        bsk = np.zeros(lib().orc_bsk_len(C.byref(self._cp)), dtype=np.uint64)
        lib().orc_gen_bsk_qp(self._ctx, lwe_sk, glwe_sk, Rng(seed, 6).ref, bsk)
        return bsk

    def gen_bsk_qp_grouped(self, lwe_sk: np.ndarray, glwe_sk: np.ndarray, seed: bytes, bits: int = 2) -> np.ndarray:
        """Grouped key over q': 2^bits GGSWs per group of `bits` (2 or 3) key bits (tfhe_oracle.h)."""
        count = lib().orc_bsk_qp_grouped_count_n(C.byref(self._cp), bits)
        bsk = np.zeros(count * self.ggsw_words(), dtype=np.uint64)
        lib().orc_gen_bsk_qp_grouped_n(self._ctx, lwe_sk, glwe_sk, Rng(seed, 7 if bits == 2 else 9).ref, bits, bsk)
        return bsk

    def prepare_bsk_qp_grouped(self, bsk: np.ndarray, bits: int = 2):
        return _PreparedBsk(self, bsk, prime=True, grouped=True, bits=bits)

    def external_product_qp(self, ggsw: np.ndarray, glwe: np.ndarray) -> np.ndarray:
        out = np.zeros((self.k + 1) * self.N, dtype=np.uint64)
        lib().orc_external_product_qp(self._ctx, np.ascontiguousarray(ggsw), np.ascontiguousarray(glwe), out)
        return out

    def prepare_bsk_qp(self, bsk: np.ndarray):
        return _PreparedBsk(self, bsk, prime=True)

    def blind_rotate_qp(self, pbsk, lwe_small: np.ndarray, lut: np.ndarray) -> np.ndarray:
        acc = np.zeros((self.k + 1) * self.N, dtype=np.uint64)
        lib().orc_blind_rotate_qp(self._ctx, pbsk.h, np.ascontiguousarray(lwe_small), np.ascontiguousarray(lut), acc)
        return acc

    def pbs_qp(self, ksk: np.ndarray, pbsk, ct_big: np.ndarray, lut: np.ndarray) -> np.ndarray:
        out = np.zeros(self.big + 1, dtype=np.uint64)
        lib().orc_pbs_qp(self._ctx, ksk, pbsk.h, np.ascontiguousarray(ct_big), np.ascontiguousarray(lut), out)
        return out

    def pbs_batch_qp(self, ksk, pbsk, cts: np.ndarray, luts: np.ndarray, lut_index=None, nthreads: int = 1):
        cts = np.ascontiguousarray(cts, dtype=np.uint64).reshape(-1, self.big + 1)
        luts = np.ascontiguousarray(luts, dtype=np.uint64)
        out = np.zeros_like(cts)
        idx = None
        if lut_index is not None:
            idx_arr = np.ascontiguousarray(lut_index, dtype=np.uint32)
            idx = idx_arr.ctypes.data_as(C.c_void_p)
        lib().orc_pbs_batch_qp(self._ctx, ksk, pbsk.h, cts, luts, idx, out, cts.shape[0], nthreads)
        return out

    def make_lut(self, f) -> np.ndarray:
        """Accumulator for f: [0, msg*carry) -> [0, msg*carry)."""
        space = self.p.message_modulus * self.p.carry_modulus
        table = np.array([f(i) % (2 * space) for i in range(space)], dtype=np.uint64)
        lut = np.zeros((self.k + 1) * self.N, dtype=np.uint64)
        lib().orc_make_lut(self._ctx, table, lut)
        return lut


class _PreparedBsk:
    def __init__(self, orc: Oracle, bsk: np.ndarray, prime: bool = False, grouped: bool = False,
                 grouped64: bool = False, bits: int = 2):
        self._orc = orc
        prep = lib().orc_bsk_qp_prepare if prime else lib().orc_bsk_prepare
        if grouped64:
            self.h = lib().orc_bsk_grouped_prepare_n(orc._ctx, np.ascontiguousarray(bsk, dtype=np.uint64), bits)
        elif grouped:
            self.h = lib().orc_bsk_qp_grouped_prepare_n(orc._ctx, np.ascontiguousarray(bsk, dtype=np.uint64), bits)
        else:
            self.h = prep(orc._ctx, np.ascontiguousarray(bsk, dtype=np.uint64))

    def __del__(self):
        try:
            lib().orc_bsk_prepared_free(self.h)
        except Exception:
            pass


# ---- reference key fixtures (tfhe-rs safe_serialize blobs) ---------------------------------------
def load_reference_cpk_fixture(golden_dir: str):
    """(secret[2048] as int64 0/1, mask[2048], body[2048]) from the reference's
    sdk/relayer/src/test/keys/{privateKey,publicKey}.bin (copies under tests/golden/)."""
    priv = open(os.path.join(golden_dir, "privateKey.bin"), "rb").read()
    pub = open(os.path.join(golden_dir, "publicKey.bin"), "rb").read()
    s = np.frombuffer(priv, dtype="<u8", count=2048, offset=24133).astype(np.int64)
    ab = np.frombuffer(pub, dtype="<u8", count=4096, offset=98)
    return s, ab[:2048].copy(), ab[2048:].copy()
