"""The Node N-API binding over the C ABI (bindings/node): the host-language side the reference's
TypeScript SDK would call in process.  Runs first in the suite: its GPU case starts `node` as a
child process, which must happen before this process has touched the GPU."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(ROOT, "bindings", "node")
ADDON = os.path.join(NODE_DIR, "lfhe_napi.node")

needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists(ADDON),
                                reason="node or the built addon is missing")


@needs_node
def test_addon_loads_and_exports_the_engine_calls():
    js = ("const m=require('./index.js');"
          "console.log(JSON.stringify({fns:Object.keys(m.native).sort(),"
          "sdk:Object.getOwnPropertyNames(m.LuxFHEClient.prototype).sort(),"
          "rt:m.fromLE32(m.toLE32(18434227713315959936n)).toString()}))")
    out = subprocess.run(["node", "-e", js], cwd=NODE_DIR, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    for fn in ("open", "createContext", "keygen", "publicKey", "encrypt", "decrypt", "evaluate", "evaluateBatch", "evaluateAsync",
               "compressionKeygen", "compress", "decompress", "storePut", "storeGet", "storeDrop", "storeStat",
               "evaluateBatchHandles"):
        assert fn in d["fns"]
    # the reference SDK's method names (packages/luxfhejs/src/index.ts:56-122)
    for m in ("initialize", "getPublicKey", "encrypt_uint8", "encrypt_uint16", "encrypt_uint32", "encrypt_uint64",
              "encrypt_uint128", "encrypt_uint256", "encrypt_address"):
        assert m in d["sdk"]
    assert d["rt"] == "18434227713315959936"


@needs_node
def test_context_creation_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    js = "const m=require('./index.js'); try{new m.NativeEngine({});process.exit(3)}catch(e){console.log(e.message)}"
    out = subprocess.run(["node", "-e", js], cwd=NODE_DIR, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "lfhe_ctx_create failed" in out.stdout


@pytest.mark.gpu
@needs_node
def test_plumbing_case_through_node():
    """42 and 17 as euint8 -> add/sub/lt/eq, a clear operand, a shift, the reference's euint64
    product vector (fhevmOperations6.ts:647-660) and a batch, all from JavaScript."""
    # and a compact input list built under the reference's public key (tests/golden), expanded from JavaScript
    import base64
    import tempfile
    import numpy as np
    from fhe_amd import tfhers_wire as W
    golden = os.path.join(ROOT, "tests", "golden")
    k = W.parse_client_key(open(os.path.join(golden, "privateKey.bin"), "rb").read())
    mask, body = W.parse_compact_public_key(open(os.path.join(golden, "publicKey.bin"), "rb").read())
    vals, widths = [0xFEEDFACE12345678, 1, 77], [64, 1, 8]
    c0, bodies = W.build_compact_list(mask, body, vals, widths, bytes(range(7, 39)))
    b64 = lambda a: base64.b64encode(np.ascontiguousarray(a, dtype="<u8").tobytes()).decode()
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"lweSk": b64(k["lwe_sk"]), "glweSk": b64(k["glwe_sk"]), "pkeSk": b64(k["pke_sk"]),
                   "list": b64(np.concatenate([c0, bodies])), "widths": widths, "packed": True,
                   "values": [str(v) for v in vals]}, f)
    try:
        out = subprocess.run(["node", "plumbing.js"], cwd=NODE_DIR, capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, LFHE_COMPACT_CASE=f.name))
    finally:
        os.unlink(f.name)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["compact"] == [str(v) for v in vals]
    assert d["add"] == "59" and d["sub"] == "25" and d["lt"] == "0" and d["eq"] == "1"
    assert d["mul64"] == "18434227713315959936" and d["batch"] == ["1", "0"]
    assert d["restored"] == ["42", "17"] and d["blobBytes"] < 8192      # compressed for storage and back
    # sixteen concurrent evaluate promises: right results, under twice the time of one, the event loop kept turning
    c = d["concurrent"]
    assert c["ok"] and c["all_ms"] < 2 * c["single_ms"] and c["ticks"] > 2, c
    m = d["concurrentMul"]          # 16 x 1814 bootstraps cannot hide in one product's launches: batch rate instead
    assert m["ok"] and m["all_ms"] < 13 * m["single_ms"], m      # sixteen one after the other would be 16 x
    assert d["mixed"] == ["59", "rejected"]
    # le -> select -> sub -> add on ciphertexts that stay on the GPU under 32-byte handles
    assert d["store"] == {"chain": "125", "cond": "1", "entries": 5, "bitWidth": 8, "condWidth": 1, "afterDrop": 4,
                          "dropped": 0, "unknown": "rejected"}, d["store"]
    print(f"node: one euint16 add {c['single_ms']:.1f} ms, sixteen concurrent {c['all_ms']:.1f} ms; "
          f"one euint64 mul {m['single_ms']:.1f} ms, sixteen concurrent {m['all_ms']:.1f} ms")
