"""The engine's HTTP surface and the SDK-named client, restating the reference's black-box suite
(e2e/test/fhe.test.ts:17-191) and the two client encodings (luxfhejs raw bytes, hardhat-plugin
JSON).  CPU: against the plaintext mock backend; GPU: against the real library, where the results
are also decrypted (the reference suite never decrypts)."""
import json
import threading
import urllib.request

import pytest

from fhe_amd.client import LuxFHEClient
from fhe_amd.server import serve


def _start(backend):
    srv = serve(backend, "127.0.0.1:0")
    th = threading.Thread(target=srv.serve_forever, daemon=True)
    th.start()
    return srv, f"http://127.0.0.1:{srv.server_address[1]}"


def _post(url, body, accept=None):
    req = urllib.request.Request(url, data=json.dumps(body).encode(), method="POST",
                                 headers={"Content-Type": "application/json", **({"Accept": accept} if accept else {})})
    with urllib.request.urlopen(req, timeout=600) as r:
        return r.status, r.read()


def _suite(url, decrypts: bool):
    c = LuxFHEClient(serverUrl=url)
    assert c.health()["status"] == "ok" and "threshold" in c.health() and "parties" in c.health()
    c.initialize()
    assert len(c.getPublicKey()) > 0
    for w in (8, 16, 32, 64):
        assert len(c.encrypt(123, w)) > 0
    a, b = c.encrypt_uint32(42), c.encrypt_uint32(17)
    res = {op: c.evaluate(op, a, b, 32) for op in ("add", "sub", "lt")}
    res["eq"] = c.evaluate("eq", a, a, 32)
    assert all(len(v) > 0 for v in res.values())
    assert c.verify(bytes([1, 2, 3])) is True
    if decrypts:
        assert (c.decrypt(res["add"]), c.decrypt(res["sub"]), c.decrypt(res["lt"]), c.decrypt(res["eq"])) == (59, 25, 0, 1)
        assert c.decrypt(c.evaluate("mul", a, 3, 32)) == 126                       # clear right operand
        assert c.decrypt(c.evaluate("select", a, b, 32, cond=res["lt"])) == 17
        assert c.decrypt(c.encrypt_address("0x8ba1f109551bd432803012645ac136ddd64dba72")) == 0x8ba1f109551bd432803012645ac136ddd64dba72
    # the hardhat plugin's encoding: value as string + type, JSON array back; decrypt -> {"value": "..."}
    st, body = _post(url + "/encrypt", {"value": "77", "type": "uint16"})
    ct = bytes(json.loads(body)["ciphertext"])
    st, body = _post(url + "/decrypt", {"ciphertext": list(ct)})
    assert json.loads(body)["value"] == "77"
    req = urllib.request.Request(url + "/publickey", headers={"Accept": "application/json"})
    with urllib.request.urlopen(req) as r:
        assert len(json.loads(r.read())["publicKey"]) > 0
    # errors: non-2xx with a text body
    for path, bad in (("/encrypt", {"value": 1, "bitWidth": 12}), ("/encrypt", {"bitWidth": 8}),
                      ("/evaluate", {"op": "add", "left": [1, 2], "bitWidth": 8}), ("/decrypt", {})):
        with pytest.raises(urllib.error.HTTPError) as e:
            _post(url + path, bad)
        assert e.value.code >= 400 and len(e.value.read()) > 0
    with pytest.raises(RuntimeError, match="Encryption failed"):
        c.encrypt(1 << 40, 8)


def test_http_surface_with_mock_backend():
    from tests.mock_engine import MockBackend
    srv, url = _start(MockBackend())
    try:
        _suite(url, decrypts=True)
    finally:
        srv.shutdown()


@pytest.mark.gpu
def test_http_surface_on_gpu():
    from fhe_amd.server import GpuBackend
    be = GpuBackend(seed=bytes(range(32)))
    srv, url = _start(be)
    try:
        _suite(url, decrypts=True)
        # public-key encryption by a client that holds no secret key, computed on by the server
        from fhe_amd.native import Context
        pk = LuxFHEClient(serverUrl=url).getPublicKey()
        cli = Context(seed=bytes(31) + b"\x01")
        x = cli.encrypt_public(pk, 200, 8)
        y = cli.encrypt_public(pk, 100, 8)
        c = LuxFHEClient(serverUrl=url)
        assert c.decrypt(x) == 200
        assert c.decrypt(c.evaluate("add", x, y, 8)) == 44
        cli.close()
        # in-process transport: same client API, no HTTP
        direct = LuxFHEClient(context=be.ctx)
        assert direct.decrypt(direct.evaluate("max", direct.encrypt_uint8(9), direct.encrypt_uint8(200), 8)) == 200
    finally:
        srv.shutdown()
        be.ctx.close()


def _concurrent(url, n, a, b, op="add", width=16, hex_payloads=False):
    """n clients post /evaluate at the same moment; returns (results in order, seconds).  Bodies are
    built and connections opened before the start signal, so that the time measured is the server's
    (the clients share this process's interpreter with it)."""
    import http.client
    import time
    from urllib.parse import urlparse
    c = LuxFHEClient(serverUrl=url)
    wire = (lambda v: "0x" + bytes(v).hex()) if hex_payloads else (lambda v: list(v))
    bodies = [json.dumps({"op": op, "left": wire(a[i]), "right": wire(b[i]), "bitWidth": width}).encode() for i in range(n)]
    u = urlparse(url)
    conns = [http.client.HTTPConnection(u.hostname, u.port, timeout=600) for _ in range(n)]
    for k in conns:
        k.connect()
    outs, errs = [None] * n, []
    gate = threading.Barrier(n + 1)

    def one(i):
        try:
            gate.wait()
            conns[i].request("POST", "/evaluate", body=bodies[i], headers={"Content-Type": "application/json"})
            r = conns[i].getresponse()
            data = r.read()
            if r.status != 200:
                raise RuntimeError(f"{r.status} {data[:200]!r}")
            outs[i] = data
        except Exception as e:       # noqa: BLE001 -- reported by the caller
            errs.append(e)
    ths = [threading.Thread(target=one, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    for k in conns:
        k.close()
    assert not errs, errs
    return outs, dt, c


def test_concurrent_requests_are_coalesced_on_the_mock():
    """Sixteen clients at once: one engine call for the lot (the reference's clients fan out with
    Promise.all, tests/fhevm-suite/e2e/test/instance.ts:32-36); a bad request among them fails alone."""
    from tests.mock_engine import MockBackend
    be = MockBackend()
    srv = serve(be, "127.0.0.1:0", coalesce_window=0.05)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    url = f"http://127.0.0.1:{srv.server_address[1]}"
    try:
        c = LuxFHEClient(serverUrl=url)
        a = [c.encrypt(100 * i, 16) for i in range(16)]
        b = [c.encrypt(7 + i, 16) for i in range(16)]
        before = srv.coalescer.batches
        outs, _, _ = _concurrent(url, 16, a, b)
        assert [c.decrypt(o) for o in outs] == [100 * i + 7 + i for i in range(16)]
        assert srv.coalescer.batches - before <= 3 and srv.coalescer.served >= 16     # not sixteen engine calls
        # a malformed ciphertext among good ones: its request fails, the others are served
        results, gate = {}, threading.Barrier(3)

        def post(name, left):
            gate.wait()
            try:
                results[name] = _post(url + "/evaluate", {"op": "add", "left": list(left), "right": list(b[0]), "bitWidth": 16})[0]
            except urllib.error.HTTPError as e:
                results[name] = e.code
        ths = [threading.Thread(target=post, args=("good", a[1])), threading.Thread(target=post, args=("bad", b"\x01\x02"))]
        for t in ths:
            t.start()
        gate.wait()
        for t in ths:
            t.join()
        assert results["good"] == 200 and results["bad"] >= 400
    finally:
        srv.shutdown()


@pytest.mark.gpu
def test_sixteen_concurrent_clients_cost_less_than_two_single_requests():
    """On the GPU the coalesced sixteen share their bootstrap launches: under twice one request's time."""
    import time
    from fhe_amd.server import GpuBackend
    be = GpuBackend(seed=bytes(range(32)))
    srv, url = _start(be)
    try:
        # euint8 products: 17 ms of bootstrap layers that sixteen requests share, on ciphertexts small enough (64 KB)
        # that reading sixteen bodies does not cost the interpreter more than that.  Ciphertexts go as hex
        # strings: as number arrays (the reference clients' encoding, also accepted) the bodies are megabytes of
        # JSON numbers that the server's interpreter parses one after the other.
        c = LuxFHEClient(serverUrl=url, hex_payloads=True)
        a = [c.encrypt(10 * i + 3, 8) for i in range(16)]
        b = [c.encrypt(7 + i, 8) for i in range(16)]
        c.evaluate("mul", a[0], b[0], 8)                      # schedule built and uploaded
        _concurrent(url, 16, a, b, op="mul", width=8, hex_payloads=True)         # ... for the batch shape too
        # a timing statement about a shared machine: up to four attempts, every one checked for its values, the
        # claim has to hold in one of them (a slow attempt says something about the box, not about the batching)
        attempts = []
        for _ in range(4):
            single = min(_timed(lambda: c.evaluate("mul", a[0], b[0], 8)) for _ in range(3))
            before = srv.coalescer.batches
            outs, dt, _ = _concurrent(url, 16, a, b, op="mul", width=8, hex_payloads=True)
            assert [c.decrypt(o) for o in outs] == [((10 * i + 3) * (7 + i)) & 0xFF for i in range(16)]
            calls = srv.coalescer.batches - before
            attempts.append((dt / single, dt, single, calls))
            print(f"http: one euint8 mul {single * 1e3:.1f} ms, sixteen concurrent {dt * 1e3:.1f} ms in {calls} engine calls")
            if dt < 2 * single and calls <= 3:
                break
        assert any(r < 2 and calls <= 3 for r, _, _, calls in attempts), attempts
    finally:
        srv.shutdown()
        be.ctx.close()


def _timed(fn):
    import time
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0
