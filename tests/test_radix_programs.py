"""Host logic of the integer operators, no GPU: the bootstrap schedule the library builds for each
operator is interpreted on clear blocks and must reproduce every known answer of the reference's
operator suite (tests/golden/fhevm_operator_kats.json, extracted by tools/extract_kats.py), while
never leaving the 4-bit block or the noise budget; a few schedules are also run on real ciphertexts
with the CPU oracle at small parameters."""
import json
import os
import random

import pytest

from fhe_amd.native import program, OPS, LfheError
from oracle import radix
from oracle.oracle import Oracle, Params, Rng, seed_bytes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "fhevm_operator_kats.json")))["kats"]
SELECT_KATS = json.load(open(os.path.join(GOLDEN, "fhevm_select_kats.json")))["kats"]
PROGRAM_OPS = {"add", "sub", "mul", "eq", "ne", "ge", "gt", "le", "lt", "min", "max", "and", "or", "xor", "not",
               "neg", "select"}
SHIFTS = {"shl", "shr", "rotl", "rotr"}
SPECIALISED = SHIFTS | {"div", "rem"}  # a clear right operand selects a schedule built for that value
_cache = {}


def prog(op, w, ln=1, rn=1, cn=1, scalar=None):
    key = (op, w, ln, rn, cn, scalar)
    if key not in _cache:
        _cache[key] = program(op, w, ln, rn, cn, scalar=scalar)
    return _cache[key]


def test_reference_vectors_follow_the_clear_semantics():
    """The restated clear semantics (wrapping mod 2^w, promotion to the wider type, shift amounts
    mod w) reproduce all 2394 reference vectors."""
    assert len(KATS) == 2394
    for k in KATS:
        op, w, ins, exp = radix.kat_case(k)
        got = radix.clear_op(op, w, ins[0], ins[1] if len(ins) > 1 else None)
        assert got == exp, k


def test_programs_reproduce_every_reference_vector():
    done = 0
    for k in KATS:
        op, w, ins, exp = radix.kat_case(k)
        if op in SPECIALISED:
            w = radix.type_width(k["types"][0])       # the amount / divisor does not widen the operand
            if k["types"][1].startswith("e"):         # encrypted amount: a euint8
                got = radix.run_plain(prog(op, w), ins[0], ins[1])
            else:
                got = radix.run_plain(prog(op, w, scalar=ins[1]), ins[0])
        else:
            assert op in PROGRAM_OPS
            got = radix.run_plain(prog(op, w), ins[0], ins[1] if len(ins) > 1 else None)
        assert got == exp, k
        done += 1
    assert done == 2394


def test_select_vectors():
    for k in SELECT_KATS:
        w = k["width"]
        got = radix.run_plain(prog("select", w), k["a"], k["b"], k["cond"])
        assert got == k["expected"], k


@pytest.mark.parametrize("w", [1, 8, 16, 32, 64, 128, 256])
def test_programs_on_random_and_edge_inputs(w):
    rnd = random.Random(w)
    m = (1 << w) - 1
    edge = [0, 1, m, m - 1, m >> 1, (m >> 1) + 1]
    pairs = [(a, b) for a in edge for b in edge] + [(rnd.randrange(m + 1), rnd.randrange(m + 1)) for _ in range(40)]
    ops = ["eq", "ne", "and", "or", "xor", "not", "select"] if w == 1 else sorted(PROGRAM_OPS)
    if w > 64:
        pairs = pairs[:12] + pairs[36:44]
    for op in ops:
        for noises in ((1, 1, 1), (2, 2, 2)):
            p = prog(op, w, *noises)
            for a, b in pairs:
                c = (a ^ b) & 1
                want = radix.clear_op(op, w, a, b, c)
                assert radix.run_plain(p, a, b, c) == want, (op, w, a, b)


@pytest.mark.parametrize("w", [8, 16, 32, 64, 128, 256])
def test_shifts_and_rotations_every_amount(w):
    rnd = random.Random(100 + w)
    m = (1 << w) - 1
    vals = [m, 1, 1 << (w - 1), rnd.randrange(m + 1), rnd.randrange(m + 1)]
    amounts = list(range(w)) if w <= 32 else [0, 1, 2, 3, w // 2 - 1, w // 2, w - 2, w - 1] + [rnd.randrange(w) for _ in range(6)]
    for op in sorted(SHIFTS):
        for noise in (1, 2):
            enc = prog(op, w, noise, noise)
            assert enc["rhs_width"] == 8
            for t in amounts:
                clear = prog(op, w, noise, scalar=t)
                assert clear["pbs"] <= w // 2 * 2 and (t % 2 or clear["pbs"] == 0)
                for a in vals[:3 if w > 64 else 5]:
                    want = radix.clear_op(op, w, a, t)
                    assert radix.run_plain(clear, a) == want, (op, w, a, t)
                    # the encrypted amount is a full euint8: bits above log2(w) are ignored
                    t8 = (t + w * rnd.randrange(256 // w)) & 0xFF if w < 256 else t
                    assert radix.run_plain(enc, a, t8) == want, (op, w, a, t8)
    # a clear amount beyond the width wraps like the reference's (amount mod width)
    assert radix.run_plain(prog("shl", w, scalar=w + 3), 5) == radix.clear_op("shl", w, 5, 3)


@pytest.mark.parametrize("w", [8, 16, 32, 64, 128, 256])
def test_division_by_clear_divisors(w):
    rnd = random.Random(200 + w)
    m = (1 << w) - 1
    divisors = [1, 2, 3, 5, 7, 10, 1 << (w // 2), (1 << (w // 2)) + 1, 1 << (w - 1), (1 << (w - 1)) + 1, m, m - 1]
    divisors += [rnd.randrange(1, m + 1) >> rnd.randrange(w) or 1 for _ in range(6 if w <= 64 else 2)]
    if w > 64:
            divisors = divisors[:3] + divisors[6:]
    for d in divisors:
        pq, pr = prog("div", w, scalar=d), prog("rem", w, scalar=d)
        vals = [0, 1, m, d, d - 1, (d + 1) & m, (m // d) * d, ((m // d) * d - 1) & m] + [rnd.randrange(m + 1) for _ in range(12 if w <= 64 else 3)]
        for a in vals:
            assert radix.run_plain(pq, a) == a // d, (w, a, d)
            assert radix.run_plain(pr, a) == a % d, (w, a, d)
    q2 = prog("div", w, 2, scalar=3)   # noise-2 input (a select result)
    assert radix.run_plain(q2, m) == m // 3
    with pytest.raises(LfheError):
        program("div", w, scalar=0)       # division by zero
    with pytest.raises(LfheError):
        program("div", w)                 # encrypted divisor: not in the reference's operator set


def test_fused_schedules():
    """Several operators as one schedule: the ERC20 transfer (le, select with a clear zero, sub, add)
    and a chain with a comparison result feeding a select, on clear blocks."""
    from fhe_amd.native import fused_program
    from fhe_amd.workloads import transfer_steps
    rnd = random.Random(77)
    for w in (8, 64):
        for noise in (None, [2, 2, 2]):
            p = fused_program(transfer_steps(w), [w, w, w], [5, 6], noise)
            assert p["out_widths"] == [w, w] and len(p["ins"]) == 3
            m = (1 << w) - 1
            cases = [(0, 0, 0), (m, m, m), (5, m, 5), (5, 0, 6)] + [tuple(rnd.randrange(m + 1) for _ in range(3)) for _ in range(30)]
            for frm, to, amt in cases:
                v = amt if amt <= frm else 0
                assert radix.run_plain_fused(p, [frm, to, amt]) == [frm - v, (to + v) & m], (w, frm, to, amt)
    # the four operators separately cost the same bootstraps but more layers
    p64 = fused_program(transfer_steps(64), [64] * 3, [5, 6])
    sep = [prog(o, 64) for o in ("le", "select", "sub", "add")]
    assert len(p64["layers"]) <= sum(len(q["layers"]) for q in sep) - 5
    assert p64["pbs"] <= sum(q["pbs"] for q in sep)
    # max(a, b) * 3 + (a == b), outputs of two different widths
    steps = [("max", 16, 0, 1, None), ("mul", 16, 2, ("clear", 3), None), ("eq", 16, 0, 1, None),
             ("shl", 16, 3, ("clear", 5), None)]
    p = fused_program(steps, [16, 16], [3, 4, 5])
    assert p["out_widths"] == [16, 1, 16]
    for a, b in [(7, 7), (65535, 1), (123, 45678)] + [(rnd.randrange(1 << 16), rnd.randrange(1 << 16)) for _ in range(20)]:
        t = (max(a, b) * 3) & 0xFFFF
        assert radix.run_plain_fused(p, [a, b]) == [t, int(a == b), (t << 5) & 0xFFFF]
    with pytest.raises(LfheError):
        fused_program([("add", 16, 0, 1, None)], [16, 8], [2])          # width mismatch
    with pytest.raises(LfheError):
        fused_program([("add", 16, 0, 5, None)], [16, 16], [2])         # dangling reference


@pytest.mark.parametrize("w", [8, 16, 32, 64, 128, 256])
def test_carry_and_sign_networks_on_chains_of_every_length(w):
    """The carry network (a resolved prefix that grows by groups, states composed four at a time) and
    the comparison tournament against operands built to stress them: a carry or borrow born at block
    i that runs through a stretch of `propagate` blocks and dies at block j, for every (i, j) of a
    sample, operands that differ in exactly one block, and random ones -- at every width, so that the
    chains cross every group boundary of every level."""
    rnd = random.Random(300 + w)
    m = (1 << w) - 1
    nb = w // 2
    pairs = []
    cuts = sorted(set(list(range(min(nb, 6))) + [nb - 1, nb - 2, nb // 2] + [rnd.randrange(nb) for _ in range(6)]))
    for i in cuts:
        for j in cuts:
            if j < i:
                continue
            # blocks i .. j: a + b = 3 per block (propagate), a generate right below at block i - 1 or a
            # carry born in block i itself; the rest random without carries (a | b disjoint)
            a = b = 0
            for k in range(nb):
                if i <= k <= j:
                    x = rnd.randrange(4)
                    a |= x << (2 * k); b |= (3 - x) << (2 * k)
                elif k == i - 1:
                    a |= 3 << (2 * k); b |= (1 + rnd.randrange(3)) << (2 * k)
                else:
                    x = rnd.randrange(4)
                    y = rnd.randrange(4 - x)
                    a |= x << (2 * k); b |= y << (2 * k)
            pairs += [(a, b), (a, (b + 1) & m), ((a + 1) & m, b)]
    pairs += [(0, 0), (m, m), (m, 1), (1, m), (0, 1), (1 << (w - 1), 1 << (w - 1))]
    pairs += [(rnd.randrange(m + 1), rnd.randrange(m + 1)) for _ in range(40)]
    for k in cuts:        # equal but for one block (and equal): the tournament has to find that block
        a = rnd.randrange(m + 1)
        pairs += [(a, a ^ (1 << (2 * k))), (a, a ^ (2 << (2 * k))), (a, a)]
    padd, psub, pneg = prog("add", w), prog("sub", w), prog("neg", w)
    cmps = {op: prog(op, w) for op in ("lt", "le", "gt", "ge", "min", "max", "eq", "ne")}
    for a, b in pairs:
        assert radix.run_plain(padd, a, b) == (a + b) & m, (w, a, b)
        assert radix.run_plain(psub, a, b) == (a - b) & m, (w, a, b)
        assert radix.run_plain(psub, b, a) == (b - a) & m, (w, a, b)
        assert radix.run_plain(pneg, a) == (-a) & m, (w, a)
        for op, p in cmps.items():
            assert radix.run_plain(p, a, b) == radix.clear_op(op, w, a, b), (op, w, a, b)
    if w <= 64:
        pmul = prog("mul", w)
        for a, b in pairs[::3]:
            assert radix.run_plain(pmul, a, b) == (a * b) & m, (w, a, b)


def test_product_with_a_clear_factor():
    """The schedule specialised on a clear factor (its digits pick the partial products) against every
    reference vector of a product with a clear right operand, against edge factors and random ones;
    it costs far fewer bootstraps than the general product."""
    done = 0
    for k in KATS:
        op, w, ins, exp = radix.kat_case(k)
        if op == "mul" and not k["types"][1].startswith("e"):
            assert radix.run_plain(prog("mul", w, scalar=ins[1]), ins[0]) == exp, k
            done += 1
    assert done >= 20
    rnd = random.Random(17)
    for w in (8, 16, 32, 64, 128, 256):
        m = (1 << w) - 1
        for c in [0, 1, 2, 3, 4, 16, m, m - 1, 1 << (w - 1), 1 << (w - 2)] + [rnd.randrange(m + 1) >> rnd.randrange(w) for _ in range(6)]:
            p = prog("mul", w, scalar=c)
            assert not p["rhs"]
            for a in [0, 1, m, 1 << (w - 1)] + [rnd.randrange(m + 1) for _ in range(6)]:
                assert radix.run_plain(p, a) == (a * c) & m, (w, a, c)
    assert prog("mul", 64, scalar=3)["pbs"] < 250 and prog("mul", 64, scalar=(1 << 64) - 1)["pbs"] < 1000
    assert prog("mul", 64, scalar=1)["pbs"] == 0 and prog("mul", 64, scalar=1 << 20)["pbs"] == 0
    q = prog("mul", 64, 2, scalar=5)      # a noise-2 input (a select result) is cleaned first
    assert radix.run_plain(q, 0xFEDCBA9876543210) == (5 * 0xFEDCBA9876543210) & ((1 << 64) - 1)


def test_bitwise_operators_with_a_clear_operand():
    """and / or / xor specialised on a clear operand: a table of the block alone, no bootstrap where
    the operand's digit is 0 or 3 (a byte mask costs none) -- reference vectors with a clear right
    operand, masks and random operands at every width, ebool included."""
    done = 0
    for k in KATS:
        op, w, ins, exp = radix.kat_case(k)
        if op in ("and", "or", "xor") and not k["types"][1].startswith("e"):
            assert radix.run_plain(prog(op, w, scalar=ins[1]), ins[0]) == exp, k
            done += 1
    assert done >= 30
    rnd = random.Random(23)
    for w in (1, 8, 16, 64, 256):
        m = (1 << w) - 1
        for c in [0, m, 1, m >> 1, 0xFF & m, 0xAA & m, 0x55 & m] + [rnd.randrange(m + 1) for _ in range(4)]:
            for op, f in (("and", lambda a, c: a & c), ("or", lambda a, c: a | c), ("xor", lambda a, c: a ^ c)):
                p = prog(op, w, scalar=c)
                assert not p["rhs"]
                for a in [0, m] + [rnd.randrange(m + 1) for _ in range(4)]:
                    assert radix.run_plain(p, a) == f(a, c), (op, w, a, c)
    assert prog("and", 64, scalar=0xFFFFFFFF)["pbs"] == 0 and prog("xor", 64, scalar=(1 << 64) - 1)["pbs"] == 0
    assert prog("or", 64, scalar=0x5555555555555555)["pbs"] == 32
    q = prog("and", 64, 2, scalar=0x0F0F)      # a noise-2 input passes through where it is kept
    assert radix.run_plain(q, 0xFEDCBA9876543210) == 0xFEDCBA9876543210 & 0x0F0F


def test_operands_at_noise_level_two_need_no_clean_up_for_comparisons_and_selects():
    """Results of a select leave at noise level 2.  Comparisons and equality read a - b + 3 (nothing is
    scaled), a select takes two bootstraps per block with the condition ready-scaled from the table
    that made it: the round of an argmax tournament on such operands costs 71 bootstraps, not 126."""
    from fhe_amd.native import fused_program
    rnd = random.Random(31)
    for w in (8, 64):
        m = (1 << w) - 1
        vals = [(0, 0), (m, m), (m, 0), (5, 6)] + [(rnd.randrange(m + 1), rnd.randrange(m + 1)) for _ in range(20)]
        for op in ("lt", "le", "gt", "ge", "eq", "ne", "min", "max"):
            p = prog(op, w, 2, 2)
            assert p["pbs"] <= prog(op, w)["pbs"] + w // 2 + 2, op          # no clean-up of 2 x w/2 blocks
            for a, b in vals:
                assert radix.run_plain(p, a, b) == radix.clear_op(op, w, a, b), (op, w, a, b)
        p = prog("select", w, 2, 2, 2)
        assert p["out_noise"] == 2 and p["pbs"] == 2 * (w // 2) + 1
        for a, b in vals:
            for c in (0, 1):
                assert radix.run_plain(p, a, b, c) == (a if c else b), (w, a, b, c)
    steps = [("gt", 32, 0, 1, None), ("select", 32, 0, 1, 4), ("select", 16, 2, 3, 4)]
    fresh = fused_program(steps, [32, 32, 16, 16], [5, 6])
    later = fused_program(steps, [32, 32, 16, 16], [5, 6], [2, 2, 2, 2])
    assert fresh["pbs"] == 46 and later["pbs"] <= 72 and len(later["layers"]) <= len(fresh["layers"])
    for _ in range(30):
        a, b = rnd.randrange(1 << 32), rnd.randrange(1 << 32)
        ia, ib = rnd.randrange(1 << 16), rnd.randrange(1 << 16)
        want = [a, ia] if a > b else [b, ib]
        assert radix.run_plain_fused(fresh, [a, b, ia, ib]) == want
        assert radix.run_plain_fused(later, [a, b, ia, ib]) == want


def test_random_chains_of_operators_as_one_schedule():
    """Random chains (results of selects feeding products, shifts, divisions, comparisons feeding
    selects, clear operands) built as ONE fused schedule and interpreted on clear blocks against the
    clear model: noise-level handling, pruning, layer balancing and the specialised forms meet here."""
    from fhe_amd.native import fused_program
    rnd = random.Random(91)
    binary = ["add", "sub", "mul", "and", "or", "xor", "min", "max"]
    compare = ["lt", "le", "gt", "ge", "eq", "ne"]
    for trial in range(60):
        w = rnd.choice([8, 16, 32, 64])
        m = (1 << w) - 1
        n_in = rnd.randrange(2, 4)
        widths = [w] * n_in
        steps, vw = [], list(widths)          # vw: width of every value so far
        for _ in range(rnd.randrange(3, 7)):
            ints = [i for i, x in enumerate(vw) if x == w]
            bools = [i for i, x in enumerate(vw) if x == 1]
            kind = rnd.random()
            if kind < 0.15 and bools:
                steps.append(("select", w, rnd.choice(ints), rnd.choice(ints), rnd.choice(bools))); vw.append(w)
            elif kind < 0.35:
                steps.append((rnd.choice(compare), w, rnd.choice(ints), rnd.choice(ints), None)); vw.append(1)
            elif kind < 0.5:
                op = rnd.choice(["shl", "shr", "rotl", "rotr", "div", "rem", "mul", "and"])
                c = rnd.randrange(1, m + 1) if op in ("div", "rem", "mul", "and") else rnd.randrange(w)
                steps.append((op, w, rnd.choice(ints), ("clear", c), None)); vw.append(w)
            elif kind < 0.6:
                steps.append((rnd.choice(["neg", "not"]), w, rnd.choice(ints), None, None)); vw.append(w)
            else:
                steps.append((rnd.choice(binary), w, rnd.choice(ints), rnd.choice(ints), None)); vw.append(w)
        outs = [i for i in range(n_in, len(vw))][-2:]
        p = fused_program(steps, widths, outs)
        for _ in range(6):
            vals = [rnd.choice([0, 1, m, rnd.randrange(m + 1)]) for _ in range(n_in)]
            model = list(vals)
            for op, _, a, b, c in steps:
                if op == "select":
                    model.append(model[a] if model[c] else model[b])
                elif op in ("neg", "not"):
                    model.append(radix.clear_op(op, w, model[a]))
                else:
                    model.append(radix.clear_op(op, w, model[a], b[1] if isinstance(b, tuple) else model[b]))
            assert radix.run_plain_fused(p, vals) == [model[o] for o in outs], (trial, steps, vals)


def test_layer_balancing_moves_bootstraps_and_nothing_else():
    """Balanced and as-built program of a euint64 multiplication (LFHE_LAYER_CAP=0 in a process of its
    own: the setting is read once): the same bootstraps on the same blocks through the same tables,
    the same depth, no layer after the first above one ciphertext per compute unit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys; sys.path.insert(0, %r); from fhe_amd.native import program; "
            "print(json.dumps(program('mul', 64)))" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LFHE_LAYER_CAP="0"), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    built, balanced = json.loads(r.stdout.strip().splitlines()[-1]), prog("mul", 64)
    ops = lambda p: sorted(tuple(q) for l in p["layers"] for q in l["pbs"])
    assert ops(built) == ops(balanced) and built["luts"] == balanced["luts"]
    assert len(built["layers"]) == len(balanced["layers"]) == 9
    assert [len(l["pbs"]) for l in built["layers"]][:3] == [1024, 387, 140]
    assert all(len(l["pbs"]) <= 256 for l in balanced["layers"][1:])
    rnd = random.Random(7)
    for _ in range(20):
        a, b = rnd.randrange(1 << 64), rnd.randrange(1 << 64)
        assert radix.run_plain(built, a, b) == radix.run_plain(balanced, a, b) == (a * b) & ((1 << 64) - 1)


def test_schedule_sizes_are_reported():
    """Bootstraps per operation (the scheduler's own count; the reference publishes none)."""
    want_at_most = {("add", 64): (150, 4), ("sub", 64): (150, 4), ("mul", 64): (1900, 9), ("lt", 64): (45, 4),
                    ("eq", 64): (45, 4), ("select", 64): (64, 2), ("add", 256): (700, 5), ("lt", 256): (175, 5)}
    for (op, w), (cap, depth) in want_at_most.items():
        p = prog(op, w)
        assert p["pbs"] == sum(len(l["pbs"]) for l in p["layers"]) <= cap, (op, w, p["pbs"])
        assert len(p["layers"]) <= depth, (op, w, len(p["layers"]))


def test_unsupported_operator_is_an_error():
    with pytest.raises(LfheError):
        program("cast", 8)


def test_programs_on_real_ciphertexts_small_parameters():
    """Full cryptography on the CPU: the same schedules over LWE blocks with the oracle's PBS."""
    orc = Oracle(Params(lwe_dim=24, poly_size=512, lwe_noise_log2=36, glwe_noise_log2=12))
    keys = orc.gen_keys(seed_bytes(41))
    pb = orc.prepare_bsk(keys[2])
    r = Rng(seed_bytes(42))
    cases = [("add", 8, 200, 100, 0), ("sub", 8, 3, 250, 0), ("mul", 8, 23, 45, 0), ("lt", 8, 77, 78, 0),
             ("ge", 8, 78, 78, 0), ("eq", 8, 5, 5, 0), ("select", 8, 11, 22, 1), ("max", 8, 9, 200, 0),
             ("xor", 8, 0xAA, 0x0F, 0), ("neg", 8, 1, 0, 0)]
    for op, w, a, b, c in cases:
        got = radix.run_oracle(prog(op, w), orc, keys, pb, r, a, b, c)
        assert got == radix.clear_op(op, w, a, b, c), (op, a, b)
    # 64-bit schedules compose four states per bootstrap through the padding bit (tables flagged in
    # "lut_wraps"): carries that run the whole length, a borrow, signs decided by the lowest block
    m64 = (1 << 64) - 1
    wide = [("add", m64, 1), ("add", 0x0123456789ABCDEF, 0xFEDCBA9876543210), ("sub", 0, 1), ("sub", 1 << 63, 1),
            ("lt", 1 << 62, (1 << 62) + 1), ("ge", m64, m64), ("gt", (1 << 63) + 5, (1 << 63) + 4)]
    for op, a, b in wide:
        p = prog(op, 64)
        assert any(p["lut_wraps"]), op
        assert radix.run_oracle(p, orc, keys, pb, r, a, b) == radix.clear_op(op, 64, a, b), (op, a, b)
    for op, a, t in [("shl", 0xB5, 3), ("rotr", 0xB5, 5), ("shr", 0xB5, 7)]:   # encrypted amounts
        assert radix.run_oracle(prog(op, 8), orc, keys, pb, r, a, t) == radix.clear_op(op, 8, a, t), (op, a, t)
    for op, a, d in [("rotl", 0xB5, 3), ("div", 200, 7), ("rem", 200, 7), ("div", 255, 16)]:  # clear operand
        assert radix.run_oracle(prog(op, 8, scalar=d), orc, keys, pb, r, a) == radix.clear_op(op, 8, a, d), (op, a, d)
