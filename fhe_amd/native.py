"""ctypes binding of liblfhe.so (include/lfhe.h).  The HIP library is the only compute path:
loading fails loudly when it has not been built, and creating a context fails without a GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFHE_LIB", os.path.join(_HERE, "liblfhe.so"))  # override: A/B builds

LFHE_OK = 0
ERR_NAMES = {-1: "LFHE_ERR_ARG", -2: "LFHE_ERR_STATE", -3: "LFHE_ERR_DEVICE", -4: "LFHE_ERR_SIZE",
             -5: "LFHE_ERR_UNSUPPORTED", -6: "LFHE_ERR_FORMAT"}
KEY_CLIENT, KEY_SERVER = 1, 2
PARAMS_DEFAULT = 1
PARAMS_PRIME_BSK = 2          # same set, bootstrapping key over the NTT prime
PARAMS_PRIME_BSK_GROUPED = 3  # prime modulus, two key bits per blind-rotation step
PARAMS_Q64_GROUPED = 4        # every modulus 2^64 as published, two key bits per blind-rotation step
PARAMS_PRIME_BSK_GROUPED3 = 5  # prime modulus, three key bits per blind-rotation step
PARAMS_Q64_GROUPED3 = 6        # every modulus 2^64, three key bits per blind-rotation step
BSK_MODULUS_PRIME = 703687441727489

OPS = {"add": 0, "sub": 1, "mul": 2, "eq": 3, "ne": 4, "ge": 5, "gt": 6, "le": 7, "lt": 8, "min": 9, "max": 10,
       "and": 11, "or": 12, "xor": 13, "not": 14, "neg": 15, "select": 16, "shl": 17, "shr": 18, "rotl": 19,
       "rotr": 20, "div": 21, "rem": 22, "cast": 23}
COMPARISONS = {"eq", "ne", "ge", "gt", "le", "lt"}


class LfheError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "lwe_dim", "glwe_dim", "poly_size", "pbs_base_log", "pbs_level", "ks_base_log", "ks_level",
        "message_modulus", "carry_modulus", "lwe_noise_log2", "glwe_noise_log2", "ms_zeros_count",
        "ms_bound_log2")] + [("ms_r_sigma_factor", C.c_double), ("ms_input_variance", C.c_double),
                             ("bsk_modulus", C.c_uint64), ("bsk_grouping", C.c_uint32), ("reserved", C.c_uint32)]


def default_params() -> Params:
    p = Params()
    load().lfhe_default_params(C.byref(p))
    return p


class Step(C.Structure):
    """One step of a fused schedule (lfhe_step)."""
    _fields_ = [("op", C.c_uint32), ("bit_width", C.c_uint32), ("lhs", C.c_int32), ("rhs", C.c_int32),
                ("cond", C.c_int32), ("rhs_is_scalar", C.c_uint32), ("scalar", C.c_uint8 * 32)]


def make_steps(steps: list):
    """steps: (op name, width, lhs ref, rhs, cond ref) with rhs a value reference, or ("clear", int)
    for a clear right operand, or None for unary operators."""
    arr = (Step * len(steps))()
    for k, (op, width, lhs, rhs, cond) in enumerate(steps):
        st = arr[k]
        st.op, st.bit_width, st.lhs = OPS[op], width, lhs
        st.rhs, st.cond, st.rhs_is_scalar = -1, -1 if cond is None else cond, 0
        if isinstance(rhs, tuple):
            st.rhs_is_scalar = 1
            st.scalar[:] = list(int(rhs[1]).to_bytes(32, "little"))
        elif rhs is not None:
            st.rhs = rhs
    return arr


_lib = None


def load():
    """Load liblfhe.so.  Raises if the library was not built (python fhe_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python fhe_amd/build.py` "
                          "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32
    pu64, pu32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    sig = {
        "lfhe_version": (C.c_char_p, []),
        "lfhe_default_params": (None, [C.POINTER(Params)]),
        "lfhe_ctx_create": (i32, [C.POINTER(vp), u32, i32, C.c_char_p]),
        "lfhe_ctx_create_custom": (i32, [C.POINTER(vp), C.POINTER(Params), i32, C.c_char_p]),
        "lfhe_ctx_destroy": (None, [vp]),
        "lfhe_encryption_reseed": (i32, [vp, C.c_char_p]),
        "lfhe_arena_chunk": (u64, [u64, u64]),
        "lfhe_last_error": (C.c_char_p, [vp]),
        "lfhe_get_params": (i32, [vp, C.POINTER(Params)]),
        "lfhe_health": (i32, [vp, C.c_char_p, pu64]),
        "lfhe_keygen": (i32, [vp]),
        "lfhe_keygen_server": (i32, [vp]),
        "lfhe_keys_export": (i32, [vp, u32, vp, pu64]),
        "lfhe_keys_import": (i32, [vp, u32, vp, u64]),
        "lfhe_server_keys_import_raw": (i32, [vp, vp, u64, vp, u64, vp, u64]),
        "lfhe_client_keys_import_raw": (i32, [vp, vp, vp]),
        "lfhe_public_key": (i32, [vp, vp, pu64]),
        "lfhe_compression_keygen": (i32, [vp]),
        "lfhe_compression_keys_import_raw": (i32, [vp, vp, u64, vp, u64]),
        "lfhe_compress": (i32, [vp, vp, u64, vp, pu64]),
        "lfhe_decompress": (i32, [vp, vp, u64, vp, pu64]),
        "lfhe_device_keys_size": (i32, [vp, pu64, pu64]),
        "lfhe_device_keys_ptr": (i32, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "lfhe_device_keys_attach": (i32, [vp, vp, vp]),
        "lfhe_encrypt": (i32, [vp, C.c_char_p, u32, vp, pu64]),
        "lfhe_trivial_encrypt": (i32, [vp, C.c_char_p, u32, vp, pu64]),
        "lfhe_encrypt_public": (i32, [vp, vp, u64, C.c_char_p, u32, vp, pu64]),
        "lfhe_decrypt": (i32, [vp, vp, u64, vp, pu32]),
        "lfhe_ciphertext_size": (i32, [vp, u32, pu64]),
        "lfhe_pke_key_set": (i32, [vp, vp]),
        "lfhe_cast_from_pke": (i32, [vp, vp, u64, vp, pu64]),
        "lfhe_expand_compact_list": (i32, [vp, vp, vp, u64, i32, pu32, u32, vp, pu64]),
        "lfhe_op_from_name": (i32, [C.c_char_p]),
        "lfhe_evaluate": (i32, [vp, u32, u32, vp, u64, vp, u64, i32, vp, u64, vp, pu64]),
        "lfhe_evaluate_batch": (i32, [vp, u32, u32, u64, vp, vp, i32, vp, vp]),
        "lfhe_program_dump": (i32, [u32, u32, i32, i32, i32, C.c_char_p, C.c_char_p, pu64]),
        "lfhe_evaluate_fused": (i32, [vp, vp, u32, u32, pu32, u64, vp, vp, u32, vp]),
        "lfhe_fused_dump": (i32, [vp, u32, u32, pu32, vp, vp, u32, C.c_char_p, pu64]),
        "lfhe_store_put": (i32, [vp, C.c_char_p, vp, u64]),
        "lfhe_store_get": (i32, [vp, C.c_char_p, vp, pu64]),
        "lfhe_store_drop": (i32, [vp, C.c_char_p]),
        "lfhe_store_stat": (i32, [vp, C.c_char_p, pu64, pu64, pu32]),
        "lfhe_evaluate_batch_handles": (i32, [vp, u32, u32, u64, vp, vp, i32, vp, vp]),
        "lfhe_evaluate_fused_handles": (i32, [vp, vp, u32, u32, pu32, u64, vp, vp, u32, vp]),
        "lfhe_rand": (i32, [vp, u32, C.c_char_p, u32, vp, pu64]),
        "lfhe_pbs_count": (i32, [vp, pu64]),
        "lfhe_bsk_image_form": (i32, [vp, C.POINTER(C.c_uint32)]),
        "lfhe_lift_fallback_count": (i32, [vp, pu64]),
        "lfhe_make_lut": (i32, [vp, vp, vp]),
        "lfhe_pbs_batch": (i32, [vp, u64, vp, vp, u32, vp, vp]),
        "lfhe_pbs_batch_device": (i32, [vp, u64, vp, vp, vp, vp]),
        "lfhe_sync": (i32, [vp]),
        "lfhe_pbs_batch_device_noalloc": (i32, [vp, u64, vp, vp, vp, vp]),
        "lfhe_pbs_batch_device_timed": (i32, [vp, u64, vp, vp, vp, vp, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "lfhe_ntt_info": (i32, [vp, pu64, pu64]),
        "lfhe_ntt_forward": (i32, [vp, u64, vp, vp]),
        "lfhe_ntt_inverse": (i32, [vp, u64, vp, vp]),
        "lfhe_external_product": (i32, [vp, u64, vp, vp, vp]),
        "lfhe_keyswitch": (i32, [vp, u64, vp, vp]),
        "lfhe_blind_rotate": (i32, [vp, u64, vp, vp, u32, vp, vp]),
        "lfhe_ms_noise_reduce": (i32, [vp, u64, vp, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L._declared = sorted(sig)
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def le32(value: int) -> bytes:
    return int(value).to_bytes(32, "little", signed=False)


def program(op: str, bit_width: int, lhs_noise: int = 1, rhs_noise: int = 1, cond_noise: int = 1,
            scalar: int | None = None) -> dict:
    """The operator's bootstrap schedule (host-side plan; needs no GPU).  `scalar`: the clear shift
    amount or divisor a specialised schedule is built for."""
    import json
    L = load()
    n = C.c_uint64(0)
    sc = None if scalar is None else int(scalar).to_bytes(32, "little")
    rc = L.lfhe_program_dump(OPS[op], bit_width, lhs_noise, rhs_noise, cond_noise, sc, None, C.byref(n))
    if rc != LFHE_OK:
        raise LfheError(rc, f"no program for {op}/{bit_width}")
    buf = C.create_string_buffer(n.value)
    rc = L.lfhe_program_dump(OPS[op], bit_width, lhs_noise, rhs_noise, cond_noise, sc, buf, C.byref(n))
    if rc != LFHE_OK:
        raise LfheError(rc, f"no program for {op}/{bit_width}")
    return json.loads(buf.value.decode())


def fused_program(steps: list, in_widths: list, outputs: list, in_noise: list | None = None) -> dict:
    """The schedule of a fused request (host-side plan; needs no GPU)."""
    import json
    L = load()
    arr = make_steps(steps)
    w = (C.c_uint32 * len(in_widths))(*in_widths)
    o = (C.c_int32 * len(outputs))(*outputs)
    nz = None if in_noise is None else (C.c_int32 * len(in_noise))(*in_noise)
    n = C.c_uint64(0)
    rc = L.lfhe_fused_dump(arr, len(steps), len(in_widths), w, nz, o, len(outputs), None, C.byref(n))
    if rc != LFHE_OK:
        raise LfheError(rc, "no schedule for this request")
    buf = C.create_string_buffer(n.value)
    rc = L.lfhe_fused_dump(arr, len(steps), len(in_widths), w, nz, o, len(outputs), buf, C.byref(n))
    if rc != LFHE_OK:
        raise LfheError(rc, "no schedule for this request")
    return json.loads(buf.value.decode())


class Context:
    """One GPU, one key set.  Mirrors the engine surface the reference's clients call."""

    N = 2048

    def __init__(self, seed: bytes | None = None, device: int = 0, params: Params | None = None,
                 params_id: int = PARAMS_DEFAULT):
        """seed: 32 bytes the keys are derived from (as secret as the keys; fixed seeds are for tests),
        or None for operating-system entropy.  Encryption randomness is fresh in every context
        whatever the seed; `reseed_encryption` makes it reproducible."""
        self._L = load()
        self._h = C.c_void_p()
        if seed is None:
            seed = os.urandom(32)
        if len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        if params is None:
            rc = self._L.lfhe_ctx_create(C.byref(self._h), params_id, device, seed)
        else:
            rc = self._L.lfhe_ctx_create_custom(C.byref(self._h), C.byref(params), device, seed)
        if rc != LFHE_OK:
            raise LfheError(rc, "context creation failed (see stderr)")
        p = Params()
        self._L.lfhe_get_params(self._h, C.byref(p))
        self.params = p
        self.n = p.lwe_dim

    def close(self):
        if self._h:
            self._L.lfhe_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _ck(self, rc: int):
        if rc != LFHE_OK:
            raise LfheError(rc, self._L.lfhe_last_error(self._h).decode())

    def reseed_encryption(self, seed: bytes | None = None):
        """Restart the encryption generator from `seed` (reproducible ciphertexts: tests only) or,
        with None, from operating-system entropy."""
        if seed is not None and len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        self._ck(self._L.lfhe_encryption_reseed(self._h, seed))

    # ---- keys
    def keygen(self):
        self._ck(self._L.lfhe_keygen(self._h))

    def keygen_server(self):
        """Server keys for the client keys this context holds (after an import)."""
        self._ck(self._L.lfhe_keygen_server(self._h))

    def import_server_keys_raw(self, bsk: np.ndarray, ksk: np.ndarray, ms_zeros: np.ndarray | None = None):
        """ms_zeros: the encryptions of zero of the modulus-switch noise reduction, or None for a
        key set without them (the step is then skipped)."""
        bsk, ksk = _u64(bsk), _u64(ksk)
        z = None if ms_zeros is None else _u64(ms_zeros)
        self._ck(self._L.lfhe_server_keys_import_raw(self._h, _ptr(bsk), bsk.size, _ptr(ksk), ksk.size, _ptr(z),
                                                     0 if z is None else z.size))

    def import_client_keys_raw(self, lwe_sk: np.ndarray, glwe_sk: np.ndarray):
        a, b = _u64(lwe_sk), _u64(glwe_sk)
        assert a.size == self.n and b.size == self.N
        self._ck(self._L.lfhe_client_keys_import_raw(self._h, _ptr(a), _ptr(b)))

    def export_keys(self, which: int) -> bytes:
        n = C.c_uint64(0)
        self._ck(self._L.lfhe_keys_export(self._h, which, None, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_keys_export(self._h, which, _ptr(buf), C.byref(n)))
        return buf.tobytes()

    def import_keys(self, which: int, blob: bytes):
        buf = np.frombuffer(blob, dtype=np.uint8)
        self._ck(self._L.lfhe_keys_import(self._h, which, _ptr(buf), buf.size))

    def device_keys(self):
        """(bsk_ptr, bsk_bytes, ksk_ptr, ksk_bytes) of the device key images."""
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.lfhe_device_keys_size(self._h, C.byref(a), C.byref(b)))
        pa, pb = C.c_void_p(), C.c_void_p()
        self._ck(self._L.lfhe_device_keys_ptr(self._h, C.byref(pa), C.byref(pb)))
        return pa.value, a.value, pb.value, b.value

    def device_keys_size(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.lfhe_device_keys_size(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def attach_device_keys(self, bsk_ptr: int, ksk_ptr: int):
        self._ck(self._L.lfhe_device_keys_attach(self._h, C.c_void_p(bsk_ptr), C.c_void_p(ksk_ptr)))

    def health(self) -> str:
        n = C.c_uint64(0)
        self._ck(self._L.lfhe_health(self._h, None, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self._ck(self._L.lfhe_health(self._h, buf, C.byref(n)))
        return buf.value.decode()

    # ---- integers
    def ciphertext_size(self, bit_width: int) -> int:
        n = C.c_uint64()
        self._ck(self._L.lfhe_ciphertext_size(self._h, bit_width, C.byref(n)))
        return n.value

    def encrypt(self, value: int, bit_width: int, trivial: bool = False) -> bytes:
        n = C.c_uint64(self.ciphertext_size(bit_width))
        buf = np.zeros(n.value, dtype=np.uint8)
        fn = self._L.lfhe_trivial_encrypt if trivial else self._L.lfhe_encrypt
        self._ck(fn(self._h, le32(value % (1 << max(bit_width, 1))), bit_width, _ptr(buf), C.byref(n)))
        return buf.tobytes()

    def public_key(self) -> bytes:
        n = C.c_uint64(0)
        self._ck(self._L.lfhe_public_key(self._h, None, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_public_key(self._h, _ptr(buf), C.byref(n)))
        return buf.tobytes()

    def encrypt_public(self, pk: bytes, value: int, bit_width: int) -> bytes:
        """Encrypt with the public key only (works on a context that holds no secret key)."""
        key = np.frombuffer(pk, dtype=np.uint8)
        n = C.c_uint64(self.ciphertext_size(bit_width))
        buf = np.zeros(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_encrypt_public(self._h, _ptr(key), key.size, le32(value % (1 << max(bit_width, 1))),
                                             bit_width, _ptr(buf), C.byref(n)))
        return buf.tobytes()

    def set_pke_key(self, pke_sk: np.ndarray):
        """Secret key of public-key encryptions made elsewhere; builds the cast key."""
        k = _u64(pke_sk)
        assert k.size == self.N
        self._ck(self._L.lfhe_pke_key_set(self._h, _ptr(k)))

    def cast_from_pke(self, ct: bytes) -> bytes:
        """Ciphertext under the public-key secret -> same values under the compute key."""
        buf = np.frombuffer(ct, dtype=np.uint8)
        n = C.c_uint64(buf.size)
        out = np.zeros(buf.size, dtype=np.uint8)
        self._ck(self._L.lfhe_cast_from_pke(self._h, _ptr(buf), buf.size, _ptr(out), C.byref(n)))
        return out.tobytes()

    def expand_compact_list(self, mask: np.ndarray, bodies: np.ndarray, widths: list, packed: bool = True) -> list:
        """A client's compact ciphertext list (under the public-key secret given to set_pke_key) ->
        one ciphertext under the compute key per entry of `widths`."""
        m, b = _u64(mask).reshape(-1), _u64(bodies).reshape(-1)
        if m.size != self.N:
            raise ValueError("the mask of a compact list has 2048 words")
        w = (C.c_uint32 * len(widths))(*widths)
        n = C.c_uint64(0)
        self._ck(self._L.lfhe_expand_compact_list(self._h, _ptr(m), _ptr(b), b.size, 1 if packed else 0, w, len(widths),
                                                  None, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_expand_compact_list(self._h, _ptr(m), _ptr(b), b.size, 1 if packed else 0, w, len(widths),
                                                  _ptr(out), C.byref(n)))
        res, off = [], 0
        for wd in widths:
            sz = self.ciphertext_size(wd)
            res.append(out[off:off + sz].tobytes())
            off += sz
        return res

    def decrypt(self, ct: bytes) -> int:
        buf = np.frombuffer(ct, dtype=np.uint8)
        out = np.zeros(32, dtype=np.uint8)
        w = C.c_uint32()
        self._ck(self._L.lfhe_decrypt(self._h, _ptr(buf), buf.size, _ptr(out), C.byref(w)))
        return int.from_bytes(out.tobytes(), "little")

    def decrypt_blocks(self, blocks: np.ndarray) -> list:
        """Decrypt single radix blocks (values 0..15: message and carry bits)."""
        blocks = _u64(blocks).reshape(-1, self.N + 1)
        hdr = self.encrypt(0, 8, trivial=True)[:16]
        pad = bytes(3 * (self.N + 1) * 8)
        return [self.decrypt(hdr + b.tobytes() + pad) & 0xF for b in blocks]

    def evaluate(self, op: str, bit_width: int, lhs: bytes, rhs=None, cond: bytes | None = None) -> bytes:
        """rhs: ciphertext bytes, or a Python int for a clear right operand."""
        a = np.frombuffer(lhs, dtype=np.uint8)
        scalar = isinstance(rhs, int)
        b = np.frombuffer(le32(rhs % (1 << 256)) if scalar else (rhs or b""), dtype=np.uint8)
        c = np.frombuffer(cond or b"", dtype=np.uint8)
        out_w = 1 if op in COMPARISONS else bit_width
        n = C.c_uint64(self.ciphertext_size(out_w))
        out = np.zeros(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_evaluate(self._h, OPS[op], bit_width, _ptr(a), a.size, _ptr(b) if b.size else None,
                                       b.size, 1 if scalar else 0, _ptr(c) if c.size else None, c.size, _ptr(out),
                                       C.byref(n)))
        return out[: n.value].tobytes()

    @staticmethod
    def _column(col) -> np.ndarray:
        """n ciphertexts as the C ABI wants them, back to back: a list of byte strings is joined, an
        (n, size) byte array is used where it lies."""
        if isinstance(col, np.ndarray):
            return np.ascontiguousarray(col, dtype=np.uint8).reshape(-1)
        return np.frombuffer(b"".join(col), dtype=np.uint8)

    def _sized_column(self, col, n: int, bit_width: int, what: str) -> np.ndarray:
        """A column of n ciphertexts of one width.  The C entry points take no lengths and read
        n * ciphertext_size(width) bytes: the length is checked here."""
        a = self._column(col)
        want = n * self.ciphertext_size(bit_width)
        if a.size != want:
            raise ValueError(f"{what}: {a.size} bytes for {n} ciphertexts of width {bit_width}, expected {want}")
        return a

    @staticmethod
    def _rows(flat: np.ndarray, n: int, sz: int, as_array: bool):
        if as_array:
            return flat.reshape(n, sz)
        view = memoryview(flat)
        return [bytes(view[i * sz:(i + 1) * sz]) for i in range(n)]

    def evaluate_batch(self, op: str, bit_width: int, lhs, rhs=None, cond=None):
        """n independent instances of one operator.  Operands: lists of ciphertext byte strings, or
        (n, size) byte arrays (no copy on the way in, an array comes back); rhs may be a list of ints."""
        n = len(lhs)
        as_array = isinstance(lhs, np.ndarray)
        scalar = rhs is not None and len(rhs) > 0 and isinstance(rhs[0], int)
        A = self._sized_column(lhs, n, bit_width, "lhs")
        if rhs is None:
            B = None
        elif scalar:
            if len(rhs) != n:
                raise ValueError(f"rhs: {len(rhs)} clear operands for {n} instances")
            B = np.frombuffer(b"".join(le32(v % (1 << 256)) for v in rhs), dtype=np.uint8)
        else:   # shift and rotation amounts are euint8
            B = self._sized_column(rhs, n, 8 if op in ("shl", "shr", "rotl", "rotr") else bit_width, "rhs")
        Cn = None if cond is None else self._sized_column(cond, n, 1, "cond")
        out_w = 1 if op in COMPARISONS else bit_width
        sz = self.ciphertext_size(out_w)
        out = np.empty(n * sz, dtype=np.uint8)   # every byte is written by the library
        self._ck(self._L.lfhe_evaluate_batch(self._h, OPS[op], bit_width, n, _ptr(A), _ptr(B), 1 if scalar else 0,
                                             _ptr(Cn), _ptr(out)))
        return self._rows(out, n, sz, as_array)

    def rand(self, bit_width: int, seed: bytes, random_bits: int | None = None) -> bytes:
        """Oblivious pseudo-random ciphertext: uniform below 2^random_bits (default: the full width)."""
        assert len(seed) == 16
        n = C.c_uint64(self.ciphertext_size(bit_width))
        buf = np.zeros(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_rand(self._h, bit_width, seed, bit_width if random_bits is None else random_bits,
                                   _ptr(buf), C.byref(n)))
        return buf.tobytes()

    def evaluate_fused(self, steps: list, in_widths: list, inputs: list, outputs: list, out_widths: list) -> list:
        """Several operations as one schedule.  inputs: per input, a list of n ciphertexts or an
        (n, size) byte array; outputs: value references; out_widths: the width of each (1 for
        comparisons).  Returns per output a list of n ciphertexts (an array when the inputs were arrays)."""
        n = len(inputs[0])
        as_array = isinstance(inputs[0], np.ndarray)
        if n == 0:
            return [[] for _ in outputs]
        arr = make_steps(steps)
        w = (C.c_uint32 * len(in_widths))(*in_widths)
        o = (C.c_int32 * len(outputs))(*outputs)
        if len(inputs) != len(in_widths):
            raise ValueError("one width per input")
        ins = [self._sized_column(col, n, wd, f"input {i}") for i, (col, wd) in enumerate(zip(inputs, in_widths))]
        in_ptrs = (C.c_void_p * len(ins))(*[a.ctypes.data for a in ins])
        sizes = [self.ciphertext_size(x) for x in out_widths]
        outs = [np.empty(n * sz, dtype=np.uint8) for sz in sizes]   # every byte is written by the library
        out_ptrs = (C.c_void_p * len(outs))(*[a.ctypes.data for a in outs])
        self._ck(self._L.lfhe_evaluate_fused(self._h, arr, len(steps), len(in_widths), w, n, in_ptrs, o, len(outputs),
                                             out_ptrs))
        return [self._rows(a, n, sz, as_array) for a, sz in zip(outs, sizes)]

    # ---- ciphertexts resident on the GPU, addressed by 32-byte handles
    @staticmethod
    def _handles(hs, n: int | None = None) -> np.ndarray:
        a = np.frombuffer(b"".join(bytes(h) for h in hs), dtype=np.uint8)
        if a.size != 32 * len(hs) or (n is not None and len(hs) != n):
            raise ValueError("handles are 32 bytes each, one per instance")
        return a

    def store_put(self, handle: bytes, ct: bytes):
        buf = np.frombuffer(ct, dtype=np.uint8)
        self._ck(self._L.lfhe_store_put(self._h, bytes(handle), _ptr(buf), buf.size))

    def store_get(self, handle: bytes) -> bytes:
        n = C.c_uint64(0)
        self._ck(self._L.lfhe_store_get(self._h, bytes(handle), None, C.byref(n)))
        buf = np.empty(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_store_get(self._h, bytes(handle), _ptr(buf), C.byref(n)))
        return buf.tobytes()

    def store_drop(self, handle: bytes):
        self._ck(self._L.lfhe_store_drop(self._h, bytes(handle)))

    def store_has(self, handle: bytes) -> bool:
        return self._L.lfhe_store_stat(self._h, bytes(handle), None, None, None) == LFHE_OK

    def store_stat(self):
        """(entries, device bytes held by the store)."""
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.lfhe_store_stat(self._h, None, C.byref(a), C.byref(b), None))
        return a.value, b.value

    def evaluate_batch_handles(self, op: str, bit_width: int, lhs: list, rhs, cond, out: list):
        """evaluate_batch on stored ciphertexts: lhs / rhs / cond / out are lists of handles (rhs may be a
        list of ints: clear operands); the results are entered under the `out` handles."""
        n = len(lhs)
        if n == 0:
            return
        scalar = rhs is not None and len(rhs) > 0 and isinstance(rhs[0], int)
        A = self._handles(lhs)
        if rhs is None:
            B = None
        elif scalar:
            if len(rhs) != n:
                raise ValueError(f"rhs: {len(rhs)} clear operands for {n} instances")
            B = np.frombuffer(b"".join(le32(v % (1 << 256)) for v in rhs), dtype=np.uint8)
        else:
            B = self._handles(rhs, n)
        Cn = None if cond is None else self._handles(cond, n)
        O = self._handles(out, n)
        self._ck(self._L.lfhe_evaluate_batch_handles(self._h, OPS[op], bit_width, n, _ptr(A), _ptr(B), 1 if scalar else 0,
                                                     _ptr(Cn), _ptr(O)))

    def evaluate_fused_handles(self, steps: list, in_widths: list, inputs: list, outputs: list, out_handles: list):
        """evaluate_fused on stored ciphertexts: inputs[i] and out_handles[o] are lists of n handles."""
        n = len(inputs[0])
        if n == 0:
            return
        arr = make_steps(steps)
        w = (C.c_uint32 * len(in_widths))(*in_widths)
        o = (C.c_int32 * len(outputs))(*outputs)
        if len(inputs) != len(in_widths) or len(out_handles) != len(outputs):
            raise ValueError("one width per input, one handle list per output")
        ins = [self._handles(col, n) for col in inputs]
        outs = [self._handles(col, n) for col in out_handles]
        in_ptrs = (C.c_void_p * len(ins))(*[a.ctypes.data for a in ins])
        out_ptrs = (C.c_void_p * len(outs))(*[a.ctypes.data for a in outs])
        self._ck(self._L.lfhe_evaluate_fused_handles(self._h, arr, len(steps), len(in_widths), w, n, in_ptrs, o,
                                                     len(outputs), out_ptrs))

    # ---- compression for storage
    def compression_keygen(self):
        self._ck(self._L.lfhe_compression_keygen(self._h))

    def import_compression_keys_raw(self, packing_key: np.ndarray, decompression_bsk: np.ndarray):
        a, b = _u64(packing_key).reshape(-1), _u64(decompression_bsk).reshape(-1)
        self._ck(self._L.lfhe_compression_keys_import_raw(self._h, _ptr(a), a.size, _ptr(b), b.size))

    def compress(self, cts: list) -> bytes:
        """Ciphertexts (any widths) -> one compressed list."""
        src = np.frombuffer(b"".join(cts), dtype=np.uint8)
        n = C.c_uint64(0)
        self._ck(self._L.lfhe_compress(self._h, _ptr(src), src.size, None, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_compress(self._h, _ptr(src), src.size, _ptr(out), C.byref(n)))
        return out[:n.value].tobytes()

    def decompress(self, blob: bytes) -> list:
        """A compressed list -> its ciphertexts, one bootstrap per block."""
        src = np.frombuffer(blob, dtype=np.uint8)
        n = C.c_uint64(0)
        self._ck(self._L.lfhe_decompress(self._h, _ptr(src), src.size, None, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint8)
        self._ck(self._L.lfhe_decompress(self._h, _ptr(src), src.size, _ptr(out), C.byref(n)))
        raw, res, off = memoryview(out), [], 0
        while off < n.value:
            width = int.from_bytes(raw[off + 6:off + 8], "little")
            size = self.ciphertext_size(width)
            res.append(bytes(raw[off:off + size]))
            off += size
        return res

    def bsk_factored(self) -> bool:
        """Whether the device holds the grouped bootstrapping key in factored form (two modular
        products per entry of a step's GGSW instead of three)."""
        f = C.c_uint32(0)
        self._ck(self._L.lfhe_bsk_image_form(self._h, C.byref(f)))
        return bool(f.value)

    def lift_fallbacks(self) -> int:
        """Ciphertexts bootstrapped again on the slow exact path because a lift came near the end of its
        window (grouped key over 2^64; 0 for the other keys)."""
        n = C.c_uint64()
        self._ck(self._L.lfhe_lift_fallback_count(self._h, C.byref(n)))
        return n.value

    def pbs_count(self) -> int:
        n = C.c_uint64()
        self._ck(self._L.lfhe_pbs_count(self._h, C.byref(n)))
        return n.value

    # ---- bootstrap path
    def make_lut(self, f) -> np.ndarray:
        space = self.params.message_modulus * self.params.carry_modulus
        table = np.array([f(i) % (2 * space) for i in range(space)], dtype=np.uint64)
        body = np.zeros(self.N, dtype=np.uint64)
        self._ck(self._L.lfhe_make_lut(self._h, _ptr(table), _ptr(body)))
        return body

    def pbs_batch(self, cts: np.ndarray, lut_bodies: np.ndarray, lut_index=None) -> np.ndarray:
        cts = _u64(cts).reshape(-1, self.N + 1)
        luts = _u64(lut_bodies).reshape(-1, self.N)
        idx = None if lut_index is None else np.ascontiguousarray(lut_index, dtype=np.uint32)
        out = np.zeros_like(cts)
        self._ck(self._L.lfhe_pbs_batch(self._h, cts.shape[0], _ptr(cts), _ptr(luts), luts.shape[0], _ptr(idx),
                                        _ptr(out)))
        return out

    def pbs_batch_device(self, count: int, d_in: int, d_luts: int, d_idx: int | None, d_out: int):
        self._ck(self._L.lfhe_pbs_batch_device(self._h, count, C.c_void_p(d_in), C.c_void_p(d_luts),
                                               C.c_void_p(d_idx) if d_idx else None, C.c_void_p(d_out)))

    def pbs_batch_device_noalloc(self, count: int, d_in: int, d_luts: int, d_idx: int | None, d_out: int):
        """pbs_batch_device on the library's static scratch array: no device allocation in the call."""
        self._ck(self._L.lfhe_pbs_batch_device_noalloc(self._h, count, C.c_void_p(d_in), C.c_void_p(d_luts),
                                                       C.c_void_p(d_idx) if d_idx else None, C.c_void_p(d_out)))

    def pbs_batch_device_timed(self, count: int, d_in: int, d_luts: int, d_idx: int | None, d_out: int, iters: int):
        ks, br = C.c_double(), C.c_double()
        self._ck(self._L.lfhe_pbs_batch_device_timed(self._h, count, C.c_void_p(d_in), C.c_void_p(d_luts),
                                                     C.c_void_p(d_idx) if d_idx else None, C.c_void_p(d_out), iters,
                                                     C.byref(ks), C.byref(br)))
        return ks.value, br.value

    def sync(self):
        self._ck(self._L.lfhe_sync(self._h))

    # ---- stages
    def ntt_info(self):
        p = (C.c_uint64 * 2)()
        psi = (C.c_uint64 * 2)()
        self._ck(self._L.lfhe_ntt_info(self._h, p, psi))
        return [int(p[0]), int(p[1])], [int(psi[0]), int(psi[1])]

    def ntt_forward(self, polys: np.ndarray) -> np.ndarray:
        polys = _u64(polys).reshape(-1, self.N)
        out = np.zeros((polys.shape[0], 2, self.N), dtype=np.uint64)
        self._ck(self._L.lfhe_ntt_forward(self._h, polys.shape[0], _ptr(polys), _ptr(out)))
        return out

    def ntt_inverse(self, residues: np.ndarray) -> np.ndarray:
        res = _u64(residues).reshape(-1, 2, self.N)
        out = np.zeros((res.shape[0], self.N), dtype=np.uint64)
        self._ck(self._L.lfhe_ntt_inverse(self._h, res.shape[0], _ptr(res), _ptr(out)))
        return out

    def external_product(self, ggsw_index, glwe: np.ndarray) -> np.ndarray:
        glwe = _u64(glwe).reshape(-1, 2, self.N)
        idx = np.ascontiguousarray(ggsw_index, dtype=np.uint32)
        out = np.zeros_like(glwe)
        self._ck(self._L.lfhe_external_product(self._h, glwe.shape[0], _ptr(idx), _ptr(glwe), _ptr(out)))
        return out

    def keyswitch(self, cts: np.ndarray) -> np.ndarray:
        cts = _u64(cts).reshape(-1, self.N + 1)
        out = np.zeros((cts.shape[0], self.n + 1), dtype=np.uint64)
        self._ck(self._L.lfhe_keyswitch(self._h, cts.shape[0], _ptr(cts), _ptr(out)))
        return out

    def ms_noise_reduce(self, small: np.ndarray):
        """(reduced small ciphertexts, index of the encryption of zero added to each; -1 = none)."""
        small = np.array(_u64(small).reshape(-1, self.n + 1), copy=True)
        chosen = np.zeros(small.shape[0], dtype=np.int32)
        self._ck(self._L.lfhe_ms_noise_reduce(self._h, small.shape[0], _ptr(small), _ptr(chosen)))
        return small, chosen

    def blind_rotate(self, small: np.ndarray, lut_bodies: np.ndarray, lut_index=None) -> np.ndarray:
        small = _u64(small).reshape(-1, self.n + 1)
        luts = _u64(lut_bodies).reshape(-1, self.N)
        idx = None if lut_index is None else np.ascontiguousarray(lut_index, dtype=np.uint32)
        out = np.zeros((small.shape[0], 2, self.N), dtype=np.uint64)
        self._ck(self._L.lfhe_blind_rotate(self._h, small.shape[0], _ptr(small), _ptr(luts), luts.shape[0], _ptr(idx),
                                           _ptr(out)))
        return out
