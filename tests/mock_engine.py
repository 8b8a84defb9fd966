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
