"""Plaintext stand-in for the engine behind the HTTP surface, for CPU-side tests of the server and
client plumbing (the ecosystem's own mock strategy: scaffold-eth/packages/hardhat/test/
FHECounter.test.ts:53-152 checks plaintexts through a mock backend).  Test infrastructure only."""
import json

from oracle.radix import clear_op


class MockBackend:
    def health(self):
        return {"backend": "mock"}

    def public_key(self):
        return b"mock-public-key"

    def encrypt(self, value, width, trivial=False):
        return json.dumps({"w": width, "v": str(value % (1 << width))}).encode()

    def decrypt(self, ct):
        return int(json.loads(ct)["v"])

    def evaluate(self, op, width, left, right, cond):
        a = int(json.loads(left)["v"])
        b = right if isinstance(right, int) or right is None else int(json.loads(right)["v"])
        c = None if cond is None else int(json.loads(cond)["v"])
        if op == "cast":
            return self.encrypt(a, width)
        if json.loads(left)["w"] != width:
            raise ValueError("width mismatch")
        out_w = 1 if op in ("eq", "ne", "ge", "gt", "le", "lt") else width
        return self.encrypt(clear_op(op, width, a, b, c), out_w)

    # ---- batch / coprocessor surface (fhe_amd.native.Context signatures)
    def evaluate_batch(self, op, width, lhs, rhs=None, cond=None):
        self.batches = getattr(self, "batches", 0) + 1
        out = []
        for i, l in enumerate(lhs):
            r = None if rhs is None else rhs[i]
            c = None if cond is None else cond[i]
            if op in ("shl", "shr", "rotl", "rotr") and not isinstance(r, int):
                assert json.loads(r)["w"] == 8
            out.append(self.evaluate(op, width, l, r, c))
        return out

    def rand(self, width, seed, random_bits):
        import hashlib
        v = int.from_bytes(hashlib.sha256(seed).digest(), "big") & ((1 << random_bits) - 1)
        return self.encrypt(v, width)


class FusingMockBackend(MockBackend):
    """The stand-in with a fused entry: steps are evaluated one after the other on clear values."""

    def evaluate_fused(self, steps, in_widths, inputs, outputs, out_widths):
        self.fused_calls = getattr(self, "fused_calls", 0) + 1
        n = len(inputs[0])
        res = [[] for _ in outputs]
        for k in range(n):
            vals = [int(json.loads(col[k])["v"]) for col in inputs]
            for col, w in zip(inputs, in_widths):
                assert json.loads(col[k])["w"] == w
            for op, w, lhs, rhs, cond in steps:
                b = rhs[1] if isinstance(rhs, tuple) else (None if rhs is None else vals[rhs])
                c = None if cond is None else vals[cond]
                vals.append(clear_op(op, w, vals[lhs], b, c))
            for o, (ref, w) in enumerate(zip(outputs, out_widths)):
                res[o].append(self.encrypt(vals[ref], w))
        return res


class ResidentMockBackend(FusingMockBackend):
    """The stand-in with a handle store of its own (the surface of the library's device store): what
    crosses the "bus" is counted, so a test can see that a chain of events moves only its inputs in
    and the requested results out."""

    def __init__(self):
        self._store = {}
        self.uploads = self.downloads = 0

    def store_put(self, handle, ct):
        assert len(handle) == 32
        self._store[bytes(handle)] = ct
        self.uploads += 1

    def store_get(self, handle):
        self.downloads += 1
        return self._store[bytes(handle)]

    def store_drop(self, handle):
        del self._store[bytes(handle)]

    def evaluate_batch_handles(self, op, width, lhs, rhs, cond, out):
        g = lambda col: None if col is None else [c if isinstance(c, int) else self._store[bytes(c)] for c in col]
        for h, r in zip(out, self.evaluate_batch(op, width, g(lhs), g(rhs), g(cond))):
            self._store[bytes(h)] = r

    def evaluate_fused_handles(self, steps, in_widths, inputs, outputs, out_handles):
        cols = [[self._store[bytes(h)] for h in col] for col in inputs]
        out_w = []
        for ref in outputs:
            op, w = steps[ref - len(in_widths)][:2]
            out_w.append(1 if op in ("eq", "ne", "ge", "gt", "le", "lt") else w)
        res = self.evaluate_fused(steps, in_widths, cols, outputs, out_w)
        for hs, col in zip(out_handles, res):
            for h, r in zip(hs, col):
                self._store[bytes(h)] = r


# ---- a stand-in for fhe_amd.native.Context in the real ciphertext layout --------------------------
import ctypes as _C
import hashlib as _hashlib
import struct as _struct

import numpy as _np

_N = 2048
_DELTA = 1 << 59
_HDR = "<IHHIBBH"
_MAGIC = 0x5443464C


class MockContext:
    """Plaintext stand-in with the surface and the byte layouts of fhe_amd.native.Context, for CPU
    runs of the benchmark drivers' control flow (ranks, shards, collectives, verification flags, exit
    status).  Ciphertexts are what lfhe_trivial_encrypt writes -- header, zero masks, body =
    digit * delta -- so sizes, strides and block slicing are the real ones; operators decode, apply
    the clear semantics and encode again.  The two "device key images" are host arrays whose digest
    the client key blob carries: decrypting on a rank whose images did not arrive intact fails.
    Test infrastructure only."""

    N = _N

    def __init__(self, seed=None, device=0, params=None, params_id=1):
        self.seed = seed or bytes(32)
        self.params_id = params_id
        self.n = 918
        self._bsk = self._ksk = None
        self._client = None
        self._pbs = 0
        self.corrupt = False      # tests flip this to make a rank's results wrong

    # ---- keys
    def keygen(self):
        rng = _np.random.default_rng(list(self.seed) + [self.params_id])
        self._bsk = rng.integers(0, 256, size=1 << 16, dtype=_np.uint8)
        self._ksk = rng.integers(0, 256, size=1 << 14, dtype=_np.uint8)
        self._client = b"MOCKCK" + self._digest()

    def _digest(self):
        return _hashlib.sha256(self._bsk.tobytes() + self._ksk.tobytes()).digest()

    def export_keys(self, which):
        assert self._client is not None
        return self._client

    def import_keys(self, which, blob):
        assert blob.startswith(b"MOCKCK")
        self._client = blob

    def device_keys_size(self):
        return 1 << 16, 1 << 14

    def device_keys(self):
        return self._bsk.ctypes.data, self._bsk.size, self._ksk.ctypes.data, self._ksk.size

    def attach_device_keys(self, bsk_ptr, ksk_ptr):
        self._bsk = _np.ctypeslib.as_array((_C.c_uint8 * (1 << 16)).from_address(bsk_ptr))
        self._ksk = _np.ctypeslib.as_array((_C.c_uint8 * (1 << 14)).from_address(ksk_ptr))

    def _require_keys(self):
        if self._bsk is None or self._client is None or self._client[6:] != self._digest():
            raise RuntimeError("mock: key images and client key do not belong together")

    # ---- ciphertexts
    @staticmethod
    def _blocks_of(width):
        return 1 if width == 1 else width // 2

    def ciphertext_size(self, width):
        return 16 + self._blocks_of(width) * (_N + 1) * 8

    def encrypt(self, value, width, trivial=False):
        nb = self._blocks_of(width)
        out = _np.zeros(2 + nb * (_N + 1), dtype=_np.uint64)
        out[:2] = _np.frombuffer(_struct.pack(_HDR, _MAGIC, 1, width, nb, 1, 1 if width == 1 else 3, 0), dtype=_np.uint64)
        v = int(value) % (1 << width)
        for i in range(nb):
            out[2 + i * (_N + 1) + _N] = ((v >> (2 * i)) & (1 if width == 1 else 3)) * _DELTA
        return out.tobytes()

    def _decode(self, ct):
        raw = bytes(ct)
        magic, version, width, nb, _, _, _ = _struct.unpack(_HDR, raw[:16])
        assert magic == _MAGIC and version == 1 and len(raw) == self.ciphertext_size(width), "mock: malformed ciphertext"
        words = _np.frombuffer(raw[16:], dtype=_np.uint64).reshape(nb, _N + 1)
        v = 0
        for i in range(nb):
            v += (((int(words[i, _N]) + _DELTA // 2) // _DELTA) % 16) << (2 * i)
        return width, v % (1 << width)

    def decrypt(self, ct):
        self._require_keys()
        return self._decode(ct)[1]

    def decrypt_blocks(self, blocks):
        self._require_keys()
        blocks = _np.asarray(blocks, dtype=_np.uint64).reshape(-1, _N + 1)
        return [int(((int(b[_N]) + _DELTA // 2) // _DELTA) % 16) for b in blocks]

    # ---- bootstraps
    def make_lut(self, f):
        body = _np.zeros(_N, dtype=_np.uint64)
        for j in range(_N):
            k = (j + 64) // 128
            val = (f(k % 16) % 32) * _DELTA
            body[j] = (-val) % (1 << 64) if k == 16 else val
        return body

    def pbs_batch(self, cts, lut_bodies, lut_index=None):
        self._require_keys()
        cts = _np.asarray(cts, dtype=_np.uint64).reshape(-1, _N + 1)
        luts = _np.asarray(lut_bodies, dtype=_np.uint64).reshape(-1, _N)
        out = _np.zeros_like(cts)
        for i, b in enumerate(cts):
            m = ((int(b[_N]) + _DELTA // 2) // _DELTA) % 16
            lut = luts[0 if lut_index is None else int(lut_index[i])]
            out[i, _N] = lut[m * 128]
            if self.corrupt and i == 0:
                out[i, _N] += _np.uint64(_DELTA)
        self._pbs += len(cts)
        return out

    def pbs_batch_device_timed(self, count, d_in, d_luts, d_idx, d_out, iters):
        view = lambda ptr, n: _np.ctypeslib.as_array((_C.c_uint64 * n).from_address(ptr))
        cts = view(d_in, count * (_N + 1)).reshape(count, _N + 1)
        idx = None if not d_idx else _np.ctypeslib.as_array((_C.c_uint32 * count).from_address(d_idx))
        nl = 1 if idx is None else int(idx.max()) + 1
        out = None
        for _ in range(iters):
            out = self.pbs_batch(cts, view(d_luts, nl * _N).reshape(nl, _N), idx)
        view(d_out, count * (_N + 1)).reshape(count, _N + 1)[:] = out
        return 0.01, 1.0

    def pbs_count(self):
        return self._pbs

    def sync(self):
        pass

    def close(self):
        pass

    # ---- operators
    def evaluate(self, op, width, lhs, rhs=None, cond=None):
        self._require_keys()
        w, a = self._decode(lhs)
        assert w == width, "mock: width mismatch"
        b = rhs if isinstance(rhs, int) or rhs is None else self._decode(rhs)[1]
        c = None if cond is None else self._decode(cond)[1]
        out_w = 1 if op in ("eq", "ne", "ge", "gt", "le", "lt") else width
        self._pbs += 1
        return self.encrypt(clear_op(op, width, a, b, c), out_w)

    @staticmethod
    def _rows(col):
        return [bytes(r) for r in col]

    def evaluate_batch(self, op, width, lhs, rhs=None, cond=None):
        as_array = isinstance(lhs, _np.ndarray)
        L = self._rows(lhs)
        R = [None] * len(L) if rhs is None else (list(rhs) if len(rhs) and isinstance(rhs[0], int) else self._rows(rhs))
        Cn = [None] * len(L) if cond is None else self._rows(cond)
        out = [self.evaluate(op, width, l, r, c) for l, r, c in zip(L, R, Cn)]
        return _np.stack([_np.frombuffer(o, dtype=_np.uint8) for o in out]) if as_array and out else out

    def evaluate_fused(self, steps, in_widths, inputs, outputs, out_widths):
        self._require_keys()
        as_array = isinstance(inputs[0], _np.ndarray)
        n = len(inputs[0])
        res = [[] for _ in outputs]
        for k in range(n):
            vals = []
            for col, w in zip(inputs, in_widths):
                cw, v = self._decode(col[k])
                assert cw == w
                vals.append(v)
            for op, w, lhs, rhs, cond in steps:
                b = rhs[1] if isinstance(rhs, tuple) else (None if rhs is None else vals[rhs])
                c = None if cond is None else vals[cond]
                vals.append(clear_op(op, w, vals[lhs], b, c))
                self._pbs += 1
            for o, (ref, w) in enumerate(zip(outputs, out_widths)):
                res[o].append(self.encrypt(vals[ref], w))
        if as_array:
            return [_np.stack([_np.frombuffer(c, dtype=_np.uint8) for c in col]) for col in res]
        return res


class MockBenchBackend:
    """bench.py / bench_workloads.py on the CPU: host tensors, gloo, MockContext behind the real
    key-replication code of fhe_amd.sharding."""
    collective = "gloo"

    def __init__(self, corrupt_rank=None):
        self.corrupt_rank = corrupt_rank

    def device(self, local):
        import torch
        return torch.device("cpu")

    def context(self, dist, rank, world, dev, seed, params_id):
        from fhe_amd.sharding import replicated_context
        ctx, keep = replicated_context(dist, rank, world, dev, seed, params_id, context_cls=MockContext)
        ctx.corrupt = self.corrupt_rank == rank
        return ctx, keep

    def synchronize(self):
        pass
