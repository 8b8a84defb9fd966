"""Interpreters for operator programs (the bootstrap schedules the library builds for each integer
operator), used by the tests as checkers.  TEST INFRASTRUCTURE ONLY.

A program is the JSON of lfhe_program_dump: layers of linear block combinations followed by
lookup-table bootstraps.  Two executors:
  * run_plain  - blocks are clear integers; checks every intermediate stays inside the 4-bit block
                 and yields the clear result, to be compared with the reference's known answers;
  * run_oracle - blocks are LWE ciphertexts and every bootstrap is the CPU oracle's PBS.
Plus the clear semantics of each operator (wrapping arithmetic mod 2^w, unsigned comparisons), as
pinned by the reference's operator tests (tests/fhevm-suite/e2e/test/fhevmOperations/)."""
from __future__ import annotations

import numpy as np


def digits(value: int, width: int) -> list:
    if width == 1:
        return [value & 1]
    return [(value >> (2 * i)) & 3 for i in range(width // 2)]


def from_digits(d: list, width: int) -> int:
    if width == 1:
        return d[0] & 1
    return sum((int(v) & 3) << (2 * i) for i, v in enumerate(d)) & ((1 << width) - 1)


def _run_layers_plain(prog: dict, blk: dict) -> None:
    """Layers on clear blocks.  A table flagged in "lut_wraps" takes inputs up to 31: a blind rotation
    returns minus the entry of x - 16 for an input x with the padding bit set, and the entries of such
    a table are values modulo 32 (t >= 16 stands for t - 32), so its output may be negative until the
    next linear operation adds its constant."""
    wraps = prog.get("lut_wraps") or [0] * len(prog["luts"])
    for layer in prog["layers"]:
        # the linear operations of a layer run side by side on the GPU: none may read another's output
        written = {op["out"] for op in layer["lin"]}
        assert not any(s in written for op in layer["lin"] for _, s in op["terms"]), "linear operations of one layer depend on each other"
        for op in layer["lin"]:
            v = op["cst"] + sum(c * blk[s] for c, s in op["terms"])
            assert 0 <= v <= 31, f"block overflow {v}"
            blk[op["out"]] = v
        for i, o, l in layer["pbs"]:
            x, tab = blk[i], prog["luts"][l]
            if wraps[l]:
                assert 0 <= x <= 31, f"bootstrap input {x}"
                t = tab[x & 15] - 32 if tab[x & 15] >= 16 else tab[x & 15]
                blk[o] = -t if x >= 16 else t
            else:
                assert 0 <= x <= 15, f"bootstrap input {x}"
                blk[o] = tab[x]


def run_plain(prog: dict, lhs: int, rhs: int | None = None, cond: int | None = None) -> int:
    w = prog["width"]
    blk = {}
    for b, v in zip(prog["lhs"], digits(lhs, w)):
        blk[b] = v
    if prog["rhs"]:
        rw = prog.get("rhs_width") or w
        for b, v in zip(prog["rhs"], digits(rhs & ((1 << rw) - 1), rw)):
            blk[b] = v
    if prog["cond"]:
        blk[prog["cond"][0]] = cond & 1
    _run_layers_plain(prog, blk)
    out = [blk[b] for b in prog["out"]]
    assert all(0 <= v <= 3 for v in out), "output not carry-free"
    return from_digits(out, prog["out_width"])


def run_plain_fused(prog: dict, inputs: list) -> list:
    """A fused program (several inputs and outputs) on clear blocks: one clear value per input,
    returns one per output."""
    blk = {}
    for ids, w, v in zip(prog["ins"], prog["in_widths"], inputs):
        for b, d in zip(ids, digits(v, w)):
            blk[b] = d
    _run_layers_plain(prog, blk)
    out = []
    for ids, w in zip(prog["outs"], prog["out_widths"]):
        d = [blk[b] for b in ids]
        assert all(0 <= v <= 3 for v in d), "output not carry-free"
        out.append(from_digits(d, w))
    return out


def run_oracle(prog: dict, orc, keys, pbsk, rng, lhs: int, rhs: int | None = None, cond: int | None = None) -> int:
    """Same program on real ciphertexts with the CPU oracle (use small parameters)."""
    lwe_sk, glwe_sk, bsk, ksk = keys
    w = prog["width"]
    delta = orc.p.delta
    luts = [orc.make_lut(lambda x, t=t: t[x]) for t in prog["luts"]]
    blk = {}
    for b, v in zip(prog["lhs"], digits(lhs, w)):
        blk[b] = orc.encrypt_block(glwe_sk, v, rng)
    if prog["rhs"]:
        rw = prog.get("rhs_width") or w
        for b, v in zip(prog["rhs"], digits(rhs & ((1 << rw) - 1), rw)):
            blk[b] = orc.encrypt_block(glwe_sk, v, rng)
    if prog["cond"]:
        blk[prog["cond"][0]] = orc.encrypt_block(glwe_sk, cond & 1, rng)
    for layer in prog["layers"]:
        for op in layer["lin"]:
            acc = np.zeros(orc.big + 1, dtype=np.uint64)
            acc[-1] = np.uint64((op["cst"] * delta) % (1 << 64))
            for c, s in op["terms"]:
                acc = acc + np.uint64(c % (1 << 64)) * blk[s]
            blk[op["out"]] = acc
        for i, o, l in layer["pbs"]:
            blk[o] = orc.pbs(ksk, pbsk, blk[i], luts[l])
    return from_digits([orc.decrypt_block(glwe_sk, blk[b]) for b in prog["out"]], prog["out_width"])


# ---- clear semantics -------------------------------------------------------------------------------
def clear_op(op: str, width: int, a: int, b: int | None = None, cond: int | None = None) -> int:
    m = (1 << width) - 1
    if op == "add": return (a + b) & m
    if op == "sub": return (a - b) & m
    if op == "mul": return (a * b) & m
    if op == "and": return a & b
    if op == "or": return a | b
    if op == "xor": return a ^ b
    if op == "not": return ~a & m
    if op == "neg": return (-a) & m
    if op == "eq": return int(a == b)
    if op == "ne": return int(a != b)
    if op == "ge": return int(a >= b)
    if op == "gt": return int(a > b)
    if op == "le": return int(a <= b)
    if op == "lt": return int(a < b)
    if op == "min": return min(a, b)
    if op == "max": return max(a, b)
    if op == "select": return a if cond else b
    if op == "shl": return (a << (b % width)) & m
    if op == "shr": return a >> (b % width)
    if op == "rotl": s = b % width; return ((a << s) | (a >> (width - s))) & m if s else a
    if op == "rotr": s = b % width; return ((a >> s) | (a << (width - s))) & m if s else a
    if op == "div": return a // b
    if op == "rem": return a % b
    raise ValueError(op)


def type_width(t: str) -> int:
    if t == "ebool":
        return 1
    return int(t.replace("euint", "").replace("uint", ""))


def kat_case(k: dict):
    """(op, working width, clear inputs, expected clear result) of one reference vector.
    Mixed widths are promoted to the wider type (the narrower operand is zero-extended)."""
    widths = [type_width(t) for t in k["types"]]
    w = max(widths)
    ins = [int(v) for v in k["inputs"]]
    exp = k["expected"]
    exp = {"true": 1, "false": 0}.get(exp, exp)
    return k["op"], w, ins, int(exp)
