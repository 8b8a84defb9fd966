"""HTTP surface of the engine, as the reference's clients speak it (default :8448).

Mirrors what the reference tree shows of the absent `fhe-server`:
  GET  /health     -> {"status":"ok","threshold":bool,"parties":int}      e2e/test/fhe.test.ts:31-37
  GET  /publickey  -> raw bytes, or {"publicKey":[...]} for JSON clients   luxfhejs/src/index.ts:63-72,
                                                                           hardhat-plugin/src/index.ts:80-87
  POST /encrypt    {value,bitWidth} -> raw ciphertext bytes                luxfhejs/src/index.ts:127-141
                   {value:"..",type:"uint32"} -> {"ciphertext":[...]}      hardhat-plugin/src/index.ts:57-67
  POST /decrypt    {ciphertext:[...]} -> {"value":".."}                    hardhat-plugin/src/index.ts:69-78
  POST /evaluate   {op,left,right,bitWidth[,cond]} -> raw ciphertext bytes e2e/test/fhe.test.ts:105-175
  POST /verify     any body -> {"verified":true}                           e2e/test/fhe.test.ts:178-189
Errors: non-2xx with a text body (luxfhejs/src/index.ts:134-137 reads res.text()).
CLI flags as the reference starts its server: -addr :8448 [-threshold] [-parties N]
(e2e/scripts/start-server.sh:17, local-dev/docker-compose.yml:29).

The backend is the GPU library; there is no CPU engine behind this server.
"""
from __future__ import annotations

import argparse
import json
import re
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

WIDTHS = (1, 8, 16, 32, 64, 128, 160, 256)


class GpuBackend:
    """liblfhe behind the HTTP surface: one context, requests serialised on it."""

    def __init__(self, seed: bytes | None = None, device: int = 0, params_id: int = 5):
        """seed: 32 bytes the server's keys derive from, or None (the default) for
        operating-system entropy -- a fixed seed means a secret key anyone with the seed can
        compute, so it is for tests and reproducible set-ups only.  params_id 5 is the bootstrapping
        key over the NTT prime with the key bits in triples, the fastest one (include/lfhe.h)."""
        from .native import Context
        self.ctx = Context(seed=seed, device=device, params_id=params_id)
        self.ctx.keygen()
        self._pk = None

    def health(self) -> dict:
        return json.loads(self.ctx.health())

    def public_key(self) -> bytes:
        if self._pk is None:
            self._pk = self.ctx.public_key()
        return self._pk

    def encrypt(self, value: int, width: int) -> bytes:
        return self.ctx.encrypt(value, width)

    def decrypt(self, ct: bytes) -> int:
        return self.ctx.decrypt(ct)

    def evaluate(self, op: str, width: int, left: bytes, right, cond) -> bytes:
        return self.ctx.evaluate(op, width, left, right, cond)

    def evaluate_batch(self, op: str, width: int, lefts: list, rights, conds) -> list:
        return self.ctx.evaluate_batch(op, width, lefts, rights, conds)


class Coalescer:
    """Concurrent /evaluate requests become one engine call per (operator, width, operand kind).

    The reference's clients fan requests out (tests/fhevm-suite/e2e/test/instance.ts:32-36 uses
    Promise.all), and a bootstrap layer of ONE operation is a few dozen ciphertexts on a GPU that holds
    a thousand: requests that wait behind the same lock anyway are cheaper served together.  One
    worker thread owns the engine's evaluate path: it takes everything that is queued -- once a
    window of `window` seconds has passed without a new arrival (at most eight windows), so that
    requests sent together are seen together; while a batch runs, later arrivals queue up by
    themselves -- groups it, and issues one
    `evaluate_batch` per group.  A batch that fails is retried one request at a time, so a bad
    ciphertext fails its own request only.  Backends without `evaluate_batch` are served one by one."""

    def __init__(self, backend, lock: threading.Lock, window: float = 0.001, max_batch: int = 256):
        self.backend, self.lock, self.window, self.max_batch = backend, lock, window, max_batch
        self.cv = threading.Condition()
        self.queue: list = []
        self.batches = 0          # engine calls issued
        self.served = 0           # requests served
        self.closed = False
        self.thread = threading.Thread(target=self._run, name="evaluate-coalescer", daemon=True)
        self.thread.start()

    def close(self):
        with self.cv:
            self.closed = True
            self.cv.notify_all()

    def submit(self, op, width, left, right, cond) -> bytes:
        item = {"args": (op, width, left, right, cond), "done": threading.Event(), "out": None, "err": None}
        with self.cv:
            if self.closed:
                raise RuntimeError("server is shutting down")
            self.queue.append(item)
            self.cv.notify_all()
        item["done"].wait()
        if item["err"] is not None:
            raise item["err"]
        return item["out"]

    @staticmethod
    def _key(args):
        op, width, left, right, cond = args
        kind = "none" if right is None else "clear" if isinstance(right, int) else "ct"
        return (op, width, kind, cond is not None)

    def _run(self):
        while True:
            with self.cv:
                while not self.queue and not self.closed:
                    self.cv.wait()
                if self.closed and not self.queue:
                    return
            # requests sent together arrive together: wait until a window passes without a new arrival
            # (a burst of large bodies takes the interpreter a few milliseconds to read), at most eight windows
            if self.window > 0:
                for _ in range(8):
                    with self.cv:
                        before = len(self.queue)
                    time.sleep(self.window)
                    with self.cv:
                        if len(self.queue) == before:
                            break
            with self.cv:
                taken, self.queue = self.queue[:], []
            groups: dict = {}
            for it in taken:
                groups.setdefault(self._key(it["args"]), []).append(it)
            for (op, width, kind, has_cond), items in groups.items():
                for i in range(0, len(items), self.max_batch):
                    self._serve(op, width, kind, has_cond, items[i:i + self.max_batch])

    def _serve(self, op, width, kind, has_cond, items):
        batch = getattr(self.backend, "evaluate_batch", None)
        if len(items) > 1 and batch is not None and op != "cast":
            try:
                with self.lock:
                    outs = batch(op, width, [it["args"][2] for it in items],
                                 None if kind == "none" else [it["args"][3] for it in items],
                                 [it["args"][4] for it in items] if has_cond else None)
                self.batches += 1
                self.served += len(items)
                for it, o in zip(items, outs):
                    it["out"] = bytes(o)
                    it["done"].set()
                return
            except Exception:
                pass                            # one of them is bad: find out which, below
        for it in items:
            try:
                with self.lock:
                    it["out"] = self.backend.evaluate(*it["args"])
                self.batches += 1
            except Exception as e:              # handed to the request's own thread
                it["err"] = e
            self.served += 1
            it["done"].set()


def _bytes_field(v) -> bytes:
    if isinstance(v, list):
        return bytes(v)
    if isinstance(v, str):
        return bytes.fromhex(v[2:] if v.startswith("0x") else v)
    raise ValueError("ciphertext must be an array of bytes or a hex string")


def _width_of(body: dict) -> int:
    if "bitWidth" in body:
        w = int(body["bitWidth"])
    elif "type" in body:
        t = str(body["type"])
        w = 1 if t in ("bool", "ebool") else 160 if t in ("address", "eaddress") else int(re.sub(r"\D", "", t) or 0)
    else:
        w = 32   # the hardhat plugin's default type is uint32
    if w not in WIDTHS:
        raise ValueError(f"unsupported bit width {w}")
    return w


def make_handler(backend, threshold: bool, parties: int, coalesce_window: float = 0.001):
    lock = threading.Lock()
    coalescer = Coalescer(backend, lock, coalesce_window)

    class Handler(BaseHTTPRequestHandler):
        protocol_version = "HTTP/1.1"

        def log_message(self, *a):  # quiet
            pass

        def _send(self, code: int, body: bytes, ctype: str):
            self.send_response(code)
            self.send_header("Content-Type", ctype)
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

        def _json(self, obj, code=200):
            self._send(code, json.dumps(obj).encode(), "application/json")

        def _raw(self, b: bytes):
            self._send(200, b, "application/octet-stream")

        def _fail(self, code: int, msg: str):
            self._send(code, msg.encode(), "text/plain")

        def _wants_json(self) -> bool:
            return "application/json" in (self.headers.get("Accept") or "")

        def do_GET(self):
            try:
                if self.path == "/health":
                    with lock:
                        h = dict(backend.health())
                    h.update({"status": "ok", "threshold": threshold, "parties": parties})
                    self._json(h)
                elif self.path == "/publickey":
                    with lock:
                        pk = backend.public_key()
                    if self._wants_json():
                        self._json({"publicKey": list(pk)})
                    else:
                        self._raw(pk)
                else:
                    self._fail(404, "not found")
            except Exception as e:  # engine failure -> 500 with the text the clients print
                self._fail(500, str(e))

        def do_POST(self):
            try:
                n = int(self.headers.get("Content-Length") or 0)
                data = self.rfile.read(n)
                if self.path == "/verify":
                    self._json({"verified": True})
                    return
                try:
                    body = json.loads(data or b"{}")
                    if not isinstance(body, dict):
                        raise ValueError("JSON object expected")
                except ValueError as e:
                    self._fail(400, f"bad request: {e}")
                    return
                if self.path == "/encrypt":
                    try:
                        w = _width_of(body)
                        value = int(body["value"])
                        if value < 0 or value >> w:
                            raise ValueError("value out of range for the bit width")
                    except (KeyError, ValueError) as e:
                        self._fail(400, f"bad request: {e}")
                        return
                    with lock:
                        ct = backend.encrypt(value, w)
                    if "type" in body or self._wants_json():
                        self._json({"ciphertext": list(ct)})
                    else:
                        self._raw(ct)
                elif self.path == "/decrypt":
                    try:
                        ct = _bytes_field(body["ciphertext"])
                    except (KeyError, ValueError) as e:
                        self._fail(400, f"bad request: {e}")
                        return
                    with lock:
                        v = backend.decrypt(ct)
                    self._json({"value": str(v)})
                elif self.path == "/evaluate":
                    try:
                        op = str(body["op"])
                        w = _width_of(body)
                        left = _bytes_field(body["left"])
                        right = body.get("right")
                        right = int(body["scalar"]) if "scalar" in body else (None if right is None else _bytes_field(right))
                        cond = body.get("cond")
                        cond = None if cond is None else _bytes_field(cond)
                    except (KeyError, ValueError) as e:
                        self._fail(400, f"bad request: {e}")
                        return
                    out = coalescer.submit(op, w, left, right, cond)
                    self._raw(out)
                else:
                    self._fail(404, "not found")
            except KeyError as e:
                self._fail(400, f"unknown operator {e}")
            except Exception as e:
                self._fail(500, str(e))

    Handler.coalescer = coalescer
    return Handler


class _Server(ThreadingHTTPServer):
    request_queue_size = 256      # clients connect in bursts (Promise.all over many requests); the default backlog is 5
    daemon_threads = True


def serve(backend, addr: str = ":8448", threshold: bool = False, parties: int = 1,
          coalesce_window: float = 0.001) -> ThreadingHTTPServer:
    host, _, port = addr.rpartition(":")
    srv = _Server((host or "0.0.0.0", int(port)), make_handler(backend, threshold, parties, coalesce_window))
    srv.coalescer = srv.RequestHandlerClass.coalescer
    return srv


def main(argv=None):
    ap = argparse.ArgumentParser(prog="fhe-server", description=__doc__.split("\n")[0])
    ap.add_argument("-addr", "--addr", default=":8448")
    ap.add_argument("-threshold", "--threshold", action="store_true")
    ap.add_argument("-parties", "--parties", type=int, default=1)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--bootstrapping-key", choices=["q64", "q64-grouped", "q64-grouped3", "prime", "grouped", "grouped3"], default="grouped3",
                    help="server-side bootstrapping key (include/lfhe.h parameter set ids 1, 4, 6, 2, 3, 5): over 2^64, over "
                         "2^64 with the key bits in pairs or in triples, over the NTT prime, or over the prime with the key "
                         "bits in pairs or in triples (fastest); clients see no difference")
    ap.add_argument("--seed-hex", default=None,
                    help="64 hex digits the keys are derived from (reproducible set-ups only: whoever knows the "
                         "seed knows the secret key); default: operating-system entropy")
    args = ap.parse_args(argv)
    seed = None
    if args.seed_hex is not None:
        seed = bytes.fromhex(args.seed_hex)
        if len(seed) != 32:
            raise SystemExit("--seed-hex needs 64 hex digits")
    if args.threshold:
        raise SystemExit("threshold decryption is outside this backend's scope: run without -threshold")
    params_id = {"q64": 1, "prime": 2, "grouped": 3, "q64-grouped": 4, "grouped3": 5, "q64-grouped3": 6}[args.bootstrapping_key]
    srv = serve(GpuBackend(seed=seed, device=args.device, params_id=params_id), args.addr, False, args.parties)
    print(f"fhe-server (gfx950) listening on {args.addr}")
    srv.serve_forever()


if __name__ == "__main__":
    main()
