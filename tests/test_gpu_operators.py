"""Integer operators on the GPU through the C ABI: encrypt -> evaluate -> decrypt must equal the
decrypted results the reference's operator suite expects (tests/golden/*.json, extracted from
tests/fhevm-suite/e2e/test/fhevmOperations/), at the reference's parameter set."""
import json
import os
import time

import pytest

from oracle import radix

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "fhevm_operator_kats.json")))["kats"]
SELECT_KATS = json.load(open(os.path.join(GOLDEN, "fhevm_select_kats.json")))["kats"]
GPU_OPS = {"add", "sub", "mul", "eq", "ne", "ge", "gt", "le", "lt", "min", "max", "and", "or", "xor", "not", "neg",
           "shl", "shr", "rotl", "rotr", "div", "rem"}
SPECIALISED = {"shl", "shr", "rotl", "rotr", "div", "rem"}


@pytest.fixture(scope="module", params=[1, 2, 3, 4, 5, 6],
                ids=["bsk_q64", "bsk_prime", "bsk_prime_grouped", "bsk_q64_grouped", "bsk_prime_grouped3", "bsk_q64_grouped3"])
def ctx(request):
    """Library-generated keys, once per ring modulus of the bootstrapping key."""
    from fhe_amd.native import Context
    c = Context(seed=bytes(range(32)), params_id=request.param)
    c.keygen()
    yield c
    c.close()


def _operands(ctx, k):
    """Encrypt a vector's inputs at the promoted width: encrypted operands fresh, clear ones trivial
    (left clear operand) or passed as a scalar (right clear operand)."""
    op, w, ins, exp = radix.kat_case(k)
    enc = [t.startswith("e") for t in k["types"]]
    if op in SPECIALISED:      # amounts are euint8 / clear, divisors clear: they do not widen the operand
        w = radix.type_width(k["types"][0])
        return op, w, ctx.encrypt(ins[0], w), ctx.encrypt(ins[1], 8) if enc[1] else ins[1], exp
    lhs = ctx.encrypt(ins[0], w, trivial=not enc[0])
    rhs = None
    if len(ins) > 1:
        rhs = ctx.encrypt(ins[1], w) if enc[1] else ins[1]
    return op, w, lhs, rhs, exp


def _groups():
    """Every reference vector, grouped into batches of one shape: (operator, working width, which
    operands are encrypted).  A group is one `evaluate_batch` call."""
    groups = {}
    for k in KATS:
        op, w, ins, exp = radix.kat_case(k)
        assert op in GPU_OPS, op
        enc = tuple(t.startswith("e") for t in k["types"])
        if op in SPECIALISED:
            w = radix.type_width(k["types"][0])
        groups.setdefault((op, w, enc), []).append((k, ins, exp))
    return groups


def test_every_reference_operator_vector(ctx):
    """All 2394 decrypted-result vectors of tests/fhevm-suite/e2e/test/fhevmOperations/
    fhevmOperations1..13.ts (every overload, euint8 ... euint256, encrypted, clear-right and
    clear-left operands), end to end on the GPU: encrypt -> evaluate_batch -> decrypt."""
    groups = _groups()
    assert sum(len(v) for v in groups.values()) == len(KATS) == 2394
    t0 = time.time()
    n0 = ctx.pbs_count()
    for (op, w, enc), cases in sorted(groups.items()):
        lhs = [ctx.encrypt(ins[0], w, trivial=not enc[0]) for _, ins, _ in cases]
        rhs = None
        if len(enc) > 1:
            if not enc[1]:
                rhs = [ins[1] for _, ins, _ in cases]                     # clear right operands
            else:                                                         # shift amounts are euint8
                rhs = [ctx.encrypt(ins[1], 8 if op in SPECIALISED else w) for _, ins, _ in cases]
        out = ctx.evaluate_batch(op, w, lhs, rhs)
        got = [ctx.decrypt(o) for o in out]
        for (k, _, exp), g in zip(cases, got):
            assert g == exp, (k, g)
    print(f"{len(KATS)} reference vectors in {len(groups)} batches, {ctx.pbs_count() - n0} bootstraps, "
          f"{time.time() - t0:.1f} s")


def test_single_evaluate_on_one_vector_per_operator(ctx):
    """The one-operation entry point (POST /evaluate's shape) on the first vector of every operator."""
    seen = set()
    for k in KATS:
        if k["op"] in seen or radix.kat_case(k)[1] > 64:
            continue
        seen.add(k["op"])
        op, w, lhs, rhs, exp = _operands(ctx, k)
        assert ctx.decrypt(ctx.evaluate(op, w, lhs, rhs)) == exp, k
    assert seen == GPU_OPS


def test_select_vectors_and_noise_tracking(ctx):
    for k in SELECT_KATS:
        w = k["width"]
        c = ctx.encrypt(k["cond"], 1)
        out = ctx.evaluate("select", w, ctx.encrypt(k["a"], w), ctx.encrypt(k["b"], w), c)
        assert ctx.decrypt(out) == k["expected"], k
    # a select result carries noise level 2; operators that pack two blocks clean it first
    a, b = ctx.encrypt(0xDEADBEEF, 32), ctx.encrypt(0x12345678, 32)
    s = ctx.evaluate("select", 32, a, b, ctx.encrypt(1, 1))
    assert s[12] == 2
    assert ctx.decrypt(ctx.evaluate("mul", 32, s, s)) == (0xDEADBEEF * 0xDEADBEEF) & 0xFFFFFFFF
    assert ctx.decrypt(ctx.evaluate("add", 32, s, s)) == (2 * 0xDEADBEEF) & 0xFFFFFFFF
    assert ctx.decrypt(ctx.evaluate("gt", 32, s, b)) == 1


def test_plumbing_u8_add_sub_lt_eq(ctx):
    """The reference's plumbing case (e2e/test/fhe.test.ts:88-175, local-dev/demo.ts:44-81)."""
    a, b = ctx.encrypt(42, 8), ctx.encrypt(17, 8)
    assert ctx.decrypt(ctx.evaluate("add", 8, a, b)) == 59
    assert ctx.decrypt(ctx.evaluate("sub", 8, a, b)) == 25
    assert ctx.decrypt(ctx.evaluate("lt", 8, a, b)) == 0
    assert ctx.decrypt(ctx.evaluate("eq", 8, a, ctx.encrypt(42, 8))) == 1
    assert ctx.decrypt(ctx.evaluate("cast", 32, a)) == 42


def test_batch_matches_single_and_clear_model(ctx):
    import random
    rnd = random.Random(5)
    n = 24
    xs = [rnd.randrange(1 << 64) for _ in range(n)]
    ys = [rnd.randrange(1 << 64) for _ in range(n)]
    A = [ctx.encrypt(x, 64) for x in xs]
    B = [ctx.encrypt(y, 64) for y in ys]
    for op in ("mul", "gt", "sub"):
        out = ctx.evaluate_batch(op, 64, A, B)
        assert [ctx.decrypt(o) for o in out] == [radix.clear_op(op, 64, x, y) for x, y in zip(xs, ys)], op
    out = ctx.evaluate_batch("add", 64, A, [y & 0xFFFF for y in ys])  # clear right operands
    assert [ctx.decrypt(o) for o in out] == [(x + (y & 0xFFFF)) & ((1 << 64) - 1) for x, y in zip(xs, ys)]
    assert ctx.evaluate_batch("add", 64, [], []) == []
    # a clear factor shared by the whole batch runs the schedule built on the factor's digits (0, 1 and
    # powers of four need no bootstrap at all); factors that differ run the general schedule on constants
    m64 = (1 << 64) - 1
    for c in (0, 1, 3, 16, 1000003, 0xF123456789ABCDEF):
        n0 = ctx.pbs_count()
        out = ctx.evaluate_batch("mul", 64, A[:8], [c] * 8)
        assert [ctx.decrypt(o) for o in out] == [(x * c) & m64 for x in xs[:8]], c
        assert ctx.pbs_count() - n0 <= 8 * 900, c
    out = ctx.evaluate_batch("mul", 64, A[:8], ys[:8])
    assert [ctx.decrypt(o) for o in out] == [(x * y) & m64 for x, y in zip(xs[:8], ys[:8])]
    assert ctx.decrypt(ctx.evaluate("mul", 64, A[0], 12345)) == (xs[0] * 12345) & m64
    assert ctx.decrypt(ctx.evaluate("add", 64, ctx.evaluate("mul", 64, A[1], 1), B[1])) == (xs[1] + ys[1]) & m64
    # and / or / xor with a clear operand shared by the batch: no bootstrap where its digit is 0 or 3
    for op, f in (("and", lambda a, c: a & c), ("or", lambda a, c: a | c), ("xor", lambda a, c: a ^ c)):
        for c in (0xFFFFFFFF, 0x5A5A5A5A12345678, m64):
            out = ctx.evaluate_batch(op, 64, A[:6], [c] * 6)
            assert [ctx.decrypt(o) for o in out] == [f(x, c) for x in xs[:6]], (op, c)
        out = ctx.evaluate_batch(op, 64, A[:6], ys[:6])          # operands that differ: the general schedule
        assert [ctx.decrypt(o) for o in out] == [f(x, y) for x, y in zip(xs[:6], ys[:6])], op
    n0 = ctx.pbs_count()
    masked = ctx.evaluate("and", 64, A[2], 0xFFFF)
    assert ctx.pbs_count() == n0 and ctx.decrypt(masked) == xs[2] & 0xFFFF
    assert ctx.decrypt(ctx.evaluate("add", 64, masked, B[2])) == ((xs[2] & 0xFFFF) + ys[2]) & m64
    # clear amounts and divisors that differ between instances: grouped into specialised runs
    amounts = [(3 * i) % 5 + 60 * (i % 2) for i in range(n)]
    out = ctx.evaluate_batch("rotl", 64, A, amounts)
    assert [ctx.decrypt(o) for o in out] == [radix.clear_op("rotl", 64, x, t) for x, t in zip(xs, amounts)]
    divs = [1000003 if i % 3 else 4096 for i in range(n)]
    out = ctx.evaluate_batch("rem", 64, A[:6], divs[:6])
    assert [ctx.decrypt(o) for o in out] == [x % d for x, d in zip(xs[:6], divs[:6])]
    S = [ctx.encrypt(t, 8) for t in amounts]            # encrypted amounts: one run
    out = ctx.evaluate_batch("shr", 64, A, S)
    assert [ctx.decrypt(o) for o in out] == [radix.clear_op("shr", 64, x, t) for x, t in zip(xs, amounts)]


def test_erc20_transfer_semantics(ctx):
    """tests/fhevm-suite/e2e/contracts/EncryptedERC20.sol:87-93,208-220 and the expectations of
    test/encryptedERC20/EncryptedERC20.ts:64-115: 10000 - 1337 / 1337, and an amount above the
    balance leaves both balances unchanged."""
    def transfer(bal_from, bal_to, amount):
        ok = ctx.evaluate("le", 64, amount, bal_from)
        v = ctx.evaluate("select", 64, amount, ctx.encrypt(0, 64, trivial=True), ok)
        return ctx.evaluate("sub", 64, bal_from, v), ctx.evaluate("add", 64, bal_to, v)
    alice, bob = ctx.encrypt(10000, 64), ctx.encrypt(0, 64)
    alice, bob = transfer(alice, bob, ctx.encrypt(1337, 64))
    assert (ctx.decrypt(alice), ctx.decrypt(bob)) == (10000 - 1337, 1337)
    alice2, bob2 = transfer(alice, bob, ctx.encrypt(10000, 64))
    assert (ctx.decrypt(alice2), ctx.decrypt(bob2)) == (10000 - 1337, 1337)


def test_fused_transfer_matches_the_separate_operators(ctx):
    from fhe_amd import workloads
    cases = [(1000, 5, 400), (1000, 5, 1001), (0, 0, 0), ((1 << 64) - 1, 7, (1 << 64) - 1), (77, (1 << 64) - 1, 3)]
    F = [ctx.encrypt(c[0], 64) for c in cases]
    T = [ctx.encrypt(c[1], 64) for c in cases]
    A = [ctx.encrypt(c[2], 64) for c in cases]
    zero = ctx.encrypt(0, 64, trivial=True)
    n0 = ctx.pbs_count()
    nf, nt = workloads.transfers(ctx, F, T, A, zero)
    fused_pbs = ctx.pbs_count() - n0
    sf, st = workloads.transfers(ctx, F, T, A, zero, fused=False)
    want = [(f - (a if a <= f else 0), (t + (a if a <= f else 0)) & ((1 << 64) - 1)) for f, t, a in cases]
    assert [(ctx.decrypt(x), ctx.decrypt(y)) for x, y in zip(nf, nt)] == want
    assert [(ctx.decrypt(x), ctx.decrypt(y)) for x, y in zip(sf, st)] == want
    assert fused_pbs <= (ctx.pbs_count() - n0 - fused_pbs)
    # results of a fused schedule feed further operators
    assert ctx.decrypt(ctx.evaluate("add", 64, nf[0], nt[0])) == want[0][0] + want[0][1]


def test_batches_larger_than_the_arena_budget_run_in_chunks(ctx):
    import os
    import random
    rnd = random.Random(9)
    n = 11
    xs = [rnd.randrange(1 << 32) for _ in range(n)]
    ys = [rnd.randrange(1 << 32) for _ in range(n)]
    A, B = [ctx.encrypt(x, 32) for x in xs], [ctx.encrypt(y, 32) for y in ys]
    os.environ["LFHE_ARENA_BYTES"] = str(10 << 20)      # a few instances of add32 per arena
    try:
        out = ctx.evaluate_batch("add", 32, A, B)
        assert [ctx.decrypt(o) for o in out] == [(x + y) & 0xFFFFFFFF for x, y in zip(xs, ys)]
        out = ctx.evaluate_batch("lt", 32, A, [y & 0xFFFF for y in ys])         # clear operands, chunked too
        assert [ctx.decrypt(o) for o in out] == [int(x < (y & 0xFFFF)) for x, y in zip(xs, ys)]
        steps = [("sub", 32, 0, 1, None), ("max", 32, 2, 0, None)]
        (r,) = ctx.evaluate_fused(steps, [32, 32], [A, B], [3], [32])
        assert [ctx.decrypt(o) for o in r] == [max((x - y) & 0xFFFFFFFF, x) for x, y in zip(xs, ys)]
    finally:
        del os.environ["LFHE_ARENA_BYTES"]


@pytest.mark.parametrize("params_id", [2, 3], ids=["bsk_prime", "bsk_prime_grouped"])
def test_long_batches_cut_into_several_launches_give_the_same_bits(params_id):
    """Blind-rotation batches are cut into launches of a few rounds of workgroups (whole program
    instances each); forcing tiny launches must not change a single bit, for arena block maps and
    for contiguous batches with a table index per ciphertext."""
    import os
    import random
    import numpy as np
    from fhe_amd.native import Context
    seed = bytes(range(32))
    ref = Context(seed=seed, params_id=params_id)
    os.environ["LFHE_BR_CHUNK"] = "600"      # above the 512-thread kernel's limit, below every batch here
    try:
        cut = Context(seed=seed, params_id=params_id)
    finally:
        del os.environ["LFHE_BR_CHUNK"]
    try:
        ref.keygen()
        cut.keygen()                          # same seed, same keys
        rnd = random.Random(12)
        n = 40
        xs = [rnd.randrange(1 << 32) for _ in range(n)]
        ys = [rnd.randrange(1 << 32) for _ in range(n)]
        A, B = [ref.encrypt(x, 32) for x in xs], [ref.encrypt(y, 32) for y in ys]
        for op in ("add", "gt"):
            want = ref.evaluate_batch(op, 32, A, B)
            got = cut.evaluate_batch(op, 32, A, B)
            assert got == want, op
        assert [ref.decrypt(o) for o in cut.evaluate_batch("add", 32, A, B)] == [(x + y) & 0xFFFFFFFF for x, y in zip(xs, ys)]
        blocks = np.concatenate([np.frombuffer(a[16:], dtype=np.uint64).reshape(16, 2049) for a in A + B])   # 1280
        luts = np.stack([ref.make_lut(lambda x, j=j: (x + j) % 16) for j in range(3)])
        idx = np.array([i % 3 for i in range(len(blocks))], dtype=np.uint32)
        assert np.array_equal(cut.pbs_batch(blocks, luts, idx), ref.pbs_batch(blocks, luts, idx))
    finally:
        ref.close()
        cut.close()


def test_forged_headers_and_short_columns_are_refused(ctx):
    """Ciphertext headers and column lengths are caller data."""
    import struct
    from fhe_amd.native import LfheError
    a8 = ctx.encrypt(0x5A, 8)
    # a header that claims 128 blocks in front of an 8-bit ciphertext's 4: the cast must not read past the input
    magic, version, width, nblocks, noise, degree, rsv = struct.unpack("<IHHIBBH", a8[:16])
    assert (width, nblocks) == (8, 4)
    forged = struct.pack("<IHHIBBH", magic, version, width, 128, noise, degree, rsv) + a8[16:]
    with pytest.raises(LfheError):
        ctx.evaluate("cast", 256, forged)
    with pytest.raises(LfheError):
        ctx.evaluate("cast", 256, struct.pack("<IHHIBBH", magic, 7, width, nblocks, noise, degree, rsv) + a8[16:])
    assert ctx.decrypt(ctx.evaluate("cast", 256, a8)) == 0x5A
    # the batch entry points take no lengths: the binding checks every column against n and the width
    a16 = ctx.encrypt(3, 16)
    with pytest.raises(ValueError):
        ctx.evaluate_batch("add", 16, [a16, a16], [a16, a8])
    with pytest.raises(ValueError):
        ctx.evaluate_batch("add", 16, [a16, a16], [1])
    with pytest.raises(ValueError):
        ctx.evaluate_batch("select", 16, [a16], [a16], [a8])
    with pytest.raises(ValueError):
        ctx.evaluate_fused([("add", 16, 0, 1, None)], [16, 16], [[a16], [a8]], [2], [16])
    # key blobs: counts that overflow, or that do not match the parameter set, are a format error
    from fhe_amd.native import KEY_CLIENT, KEY_SERVER
    blob = bytearray(ctx.export_keys(KEY_CLIENT))
    hdr = len(blob) - (918 + 2048) * 8
    bad = bytes(blob[:hdr - 24]) + struct.pack("<QQQ", 1 << 61, 0, 0)
    with pytest.raises(LfheError) as e:
        ctx.import_keys(KEY_CLIENT, bad)
    assert e.value.code == -6
    with pytest.raises(LfheError) as e:
        ctx.import_keys(KEY_SERVER, bytes(blob))
    assert e.value.code == -6
    ctx.import_keys(KEY_CLIENT, bytes(blob))


def test_encryption_randomness_is_fresh_unless_reseeded():
    """Two contexts made from one key seed hold the same keys but never hand out the same masks;
    a reseeded generator is reproducible (tests and known-answer vectors only)."""
    from fhe_amd.native import Context, KEY_CLIENT
    seed = bytes(range(32))
    a, b = Context(seed=seed, params_id=3), Context(seed=seed, params_id=3)
    try:
        a.keygen()
        blob = a.export_keys(KEY_CLIENT)
        b.import_keys(KEY_CLIENT, blob)
        ca, cb = a.encrypt(77, 16), b.encrypt(77, 16)
        assert ca != cb and ca[16:48] != cb[16:48]
        assert b.decrypt(ca) == 77 and a.decrypt(cb) == 77
        a.reseed_encryption(bytes(32))
        b.reseed_encryption(bytes(32))
        assert a.encrypt(77, 16) == b.encrypt(77, 16)
        a.reseed_encryption()
        b.reseed_encryption()
        assert a.encrypt(77, 16) != b.encrypt(77, 16)
        # no seed at all: keys from operating-system entropy
        c = Context(params_id=3)
        c.close()
    finally:
        a.close()
        b.close()


def test_error_paths(ctx):
    from fhe_amd.native import LfheError
    a8, a16 = ctx.encrypt(1, 8), ctx.encrypt(1, 16)
    with pytest.raises(LfheError):
        ctx.evaluate("add", 8, a8, a16)            # width mismatch
    with pytest.raises(LfheError):
        ctx.evaluate("add", 8, a8[:-8], a8)        # truncated ciphertext
    with pytest.raises(LfheError):
        ctx.decrypt(b"\x00" * 40)                  # not a ciphertext
    with pytest.raises(LfheError):
        ctx.evaluate("select", 8, a8, a8, None)    # missing condition
    with pytest.raises(LfheError):
        ctx.encrypt(1, 12)                         # unsupported width
    with pytest.raises(LfheError):
        ctx.evaluate("div", 8, a8, 0)              # division by zero
    with pytest.raises(LfheError):
        ctx.evaluate("div", 8, a8, a8)             # encrypted divisor
    with pytest.raises(LfheError):
        ctx.evaluate("shl", 16, a16, a16)          # the amount must be a euint8


def test_programs_built_for_clear_operands_are_bounded_and_evicted():
    """Programs specialised on a clear operand are kept per value, the least recently used one goes
    beyond a limit (256; 2 here, in a process of its own because the limit is read once): results
    stay right while programs are evicted and rebuilt."""
    import subprocess
    import sys
    code = (
        "import os, sys\n"
        "sys.path.insert(0, os.getcwd())\n"
        "from fhe_amd.native import Context\n"
        "ctx = Context(seed=bytes(range(32)), params_id=3)\n"
        "ctx.keygen()\n"
        "m = (1 << 32) - 1\n"
        "a = ctx.encrypt(0xDEADBEEF, 32)\n"
        "for rep in range(2):\n"
        "    for c in (3, 5, 7, 11, 13):\n"
        "        assert ctx.decrypt(ctx.evaluate('mul', 32, a, c)) == (0xDEADBEEF * c) & m, c\n"
        "        assert ctx.decrypt(ctx.evaluate('shl', 32, a, c)) == (0xDEADBEEF << c) & m, c\n"
        "        assert ctx.decrypt(ctx.evaluate('and', 32, a, c * 0x01010101)) == 0xDEADBEEF & (c * 0x01010101), c\n"
        "for rep in range(2):\n"
        "    for c in (3, 5, 7):\n"
        "        out = ctx.evaluate_fused([('mul', 32, 0, ('clear', c), None), ('add', 32, 1, 0, None)], [32], [[a]], [2], [32])\n"
        "        assert ctx.decrypt(out[0][0]) == (0xDEADBEEF * (c + 1)) & m, c\n"
        "outs = ctx.evaluate_batch('rem', 32, [a] * 6, [3, 5, 7, 11, 13, 3])\n"
        "assert [ctx.decrypt(o) for o in outs] == [0xDEADBEEF % d for d in (3, 5, 7, 11, 13, 3)]\n"
        "ctx.close()\n"
        "print('evictions ok')\n")
    env = dict(os.environ, LFHE_PROGRAM_CACHE="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "evictions ok" in r.stdout, r.stdout + r.stderr
