"""Host-side client with the reference SDK's method names.

LuxFHEClient mirrors packages/luxfhejs/src/index.ts:42-200 (encrypt_uint8 ... encrypt_address,
getPublicKey, initialize) and adds the decrypt / evaluate calls the hardhat plugin and the e2e
suite make (packages/hardhat-plugin/src/index.ts:57-87, e2e/test/fhe.test.ts:105-175).  It talks
either to a server speaking the engine's HTTP surface, or directly to the in-process library
(transport="native"), which is what a Node N-API binding of include/lfhe.h would do."""
from __future__ import annotations

import json
import urllib.error
import urllib.request

DEFAULT_FHE_SERVER = "http://localhost:8448"


class LuxFHEClient:
    def __init__(self, serverUrl: str | None = None, context=None, hex_payloads: bool = False):
        """serverUrl: HTTP engine (default http://localhost:8448); context: a fhe_amd.Context to call
        the library in-process instead.  hex_payloads: send ciphertexts as "0x..." strings instead of
        the number arrays the reference's clients send (e2e/test/fhe.test.ts:105-121): the server takes
        both, and a 128 KB ciphertext is 0.5 MB of JSON numbers that cost it ten milliseconds to parse,
        against 0.25 MB of hex that cost it none."""
        self.serverUrl = (serverUrl or DEFAULT_FHE_SERVER).rstrip("/")
        self._ctx = context
        self._publicKey = None
        self._wire = (lambda b: "0x" + bytes(b).hex()) if hex_payloads else (lambda b: list(b))

    # ---- transport
    def _http(self, method: str, path: str, body=None, accept: str | None = None, raw_body: bytes | None = None):
        data = raw_body if raw_body is not None else (None if body is None else json.dumps(body).encode())
        req = urllib.request.Request(self.serverUrl + path, data=data, method=method)
        if body is not None:
            req.add_header("Content-Type", "application/json")
        if raw_body is not None:
            req.add_header("Content-Type", "application/octet-stream")
        if accept:
            req.add_header("Accept", accept)
        try:
            with urllib.request.urlopen(req, timeout=600) as r:
                return r.read()
        except urllib.error.HTTPError as e:
            raise RuntimeError(f"{path} failed: {e.code} {e.read().decode(errors='replace')}") from None

    # ---- reference method names
    def initialize(self) -> None:
        self.getPublicKey()

    def getPublicKey(self) -> bytes:
        if self._publicKey is None:
            self._publicKey = self._ctx.public_key() if self._ctx else self._http("GET", "/publickey")
        return self._publicKey

    def encrypt_uint8(self, value: int) -> bytes: return self.encrypt(value, 8)
    def encrypt_uint16(self, value: int) -> bytes: return self.encrypt(value, 16)
    def encrypt_uint32(self, value: int) -> bytes: return self.encrypt(value, 32)
    def encrypt_uint64(self, value: int) -> bytes: return self.encrypt(value, 64)
    def encrypt_uint128(self, value: int) -> bytes: return self.encrypt(value, 128)
    def encrypt_uint256(self, value: int) -> bytes: return self.encrypt(value, 256)
    def encrypt_bool(self, value: bool) -> bytes: return self.encrypt(int(bool(value)), 1)

    def encrypt_address(self, address: str) -> bytes:
        return self.encrypt(int(address, 16), 160)

    def encrypt(self, value: int, bitWidth: int) -> bytes:
        if self._ctx:
            return self._ctx.encrypt(int(value), bitWidth)
        try:
            return self._http("POST", "/encrypt", {"value": int(value), "bitWidth": bitWidth})
        except RuntimeError as e:
            raise RuntimeError(f"Encryption failed: {e}") from None

    # ---- what the hardhat plugin and the e2e suite call
    def decrypt(self, ciphertext: bytes) -> int:
        if self._ctx:
            return self._ctx.decrypt(ciphertext)
        return int(json.loads(self._http("POST", "/decrypt", {"ciphertext": self._wire(ciphertext)}))["value"])

    def evaluate(self, op: str, left: bytes, right, bitWidth: int, cond: bytes | None = None) -> bytes:
        """right: ciphertext bytes, or an int for a clear right operand."""
        if self._ctx:
            return self._ctx.evaluate(op, bitWidth, left, right, cond)
        body = {"op": op, "left": self._wire(left), "bitWidth": bitWidth}
        if isinstance(right, int):
            body["scalar"] = str(right)
        elif right is not None:
            body["right"] = self._wire(right)
        if cond is not None:
            body["cond"] = self._wire(cond)
        return self._http("POST", "/evaluate", body)

    def health(self) -> dict:
        if self._ctx:
            return json.loads(self._ctx.health())
        return json.loads(self._http("GET", "/health"))

    def verify(self, proof: bytes) -> bool:
        if self._ctx:
            return True
        return bool(json.loads(self._http("POST", "/verify", raw_body=proof))["verified"])
