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

    def encrypt(self, value, width):
        return json.dumps({"w": width, "v": str(value)}).encode()

    def decrypt(self, ct):
        return int(json.loads(ct)["v"])

    def evaluate(self, op, width, left, right, cond):
        a = int(json.loads(left)["v"])
        b = right if isinstance(right, int) or right is None else int(json.loads(right)["v"])
        c = None if cond is None else int(json.loads(cond)["v"])
        if json.loads(left)["w"] != width:
            raise ValueError("width mismatch")
        out_w = 1 if op in ("eq", "ne", "ge", "gt", "le", "lt") else width
        return self.encrypt(clear_op(op, width, a, b, c), out_w)
