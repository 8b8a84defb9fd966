#!/usr/bin/env python3
"""Worst-case magnitude bounds (in units of p) of the lazily reduced transforms of pbs_device.h.

Every value is an exact integer carried in an f64 and must stay below 2^53 = 12.79 p.
  * modular product r = b*w - rint(fl(fl(b*w) * fl(1/p))) * p with |w| <= p/2: the quotient estimate
    carries three roundings, so |r| <= (0.5 + C |b|/p) p with C = (p/2) * 3 * 2^-53 = 0.1172;
  * Cooley-Tukey butterfly (a, b) -> (a + t, a - t), t = b*w: both outputs <= |a| + P(|b|);
  * Gentleman-Sande butterfly (a, b) -> (a + b, (a - b)*w): outputs |a| + |b| and P(|a| + |b|);
  * a normalisation brings a value to 0.5 p (+ 1e-9);
  * an LDS exchange mixes registers between threads: afterwards every register carries the
    largest bound any register had.
The script prints the bound after every stage for a normalisation schedule and checks the limit.
Run with --search to look for the cheapest per-register schedule."""
import itertools
import sys

P_ = 703687441727489.0
LIMIT = 2.0 ** 53 / P_ - 0.05
C = (P_ / 2) * 3 * 2.0 ** -53
NORM = 0.5 + 1e-9


def prod(b):
    return 0.5 + C * b


def forward(start=2.0 ** 22 / P_):
    """11 Cooley-Tukey stages from digits |d| <= 2^22, no normalisation."""
    b = start
    out = []
    for _ in range(11):
        b = b + prod(b)
        out.append(b)
    return out


PASS4 = [[(0, 1), (2, 3), (4, 5), (6, 7)], [(0, 2), (1, 3), (4, 6), (5, 7)]]
PASS8 = PASS4 + [[(0, 4), (1, 5), (2, 6), (3, 7)]]


def inverse(schedule, start, verbose=False):
    """schedule: dict (pass index 0..3, stage index within pass) -> set of registers normalised
    AFTER that stage.  Returns (worst bound seen, final bounds, number of normalisations)."""
    x = [start] * 8
    worst, norms = start, 0
    for pi, stages in enumerate([PASS4, PASS8, PASS8, PASS8]):
        for si, pairs in enumerate(stages):
            for a, b in pairs:
                s = x[a] + x[b]
                worst = max(worst, s)
                x[a], x[b] = s, prod(s)
            for r in schedule.get((pi, si), ()):
                x[r] = NORM
                norms += 1
            if verbose:
                print(f"  pass {4 - pi} stage {si + 1}: " + " ".join(f"{v:6.2f}" for v in x))
        if pi < 3:
            x = [max(x)] * 8   # exchange
    return worst, x, norms


def schedule_from_header(path, grouped=False):
    """The GS_NORM_PASS* masks of pbs_device.h as a schedule (grouped: the variant the grouped blind
    rotation compiles in)."""
    import re
    text = open(path).read()
    sched = {}
    names = ("GS_NORM_PASS4", "GS_NORM_PASS3", "GS_NORM_PASS2_GROUPED" if grouped else "GS_NORM_PASS2", "GS_NORM_PASS1")
    for pi, name in enumerate(names):
        m = re.search(name + r"\[\d\] = \{([^}]*)\}", text)
        for si, tok in enumerate(m.group(1).split(",")):
            mask = int(tok.strip(), 0)
            if mask:
                sched[(pi, si)] = {r for r in range(8) if mask >> r & 1}
    return sched


def prod2(b, w):
    """General modular product, neither operand confined to p/2: the quotient estimate must stay
    below 2^53 (b w < 2^53 / p) and carries the same three roundings."""
    assert b * w < LIMIT, (b, w)
    return 0.5 + 2 * C * b * w


def grouped_pointwise(monomials=3):
    """Grouped blind rotation: bound of the transformed digits after the last forward pass when only
    registers 0 and 4 are normalised before it (ct_radix4x2_bounded), bound of the step's GGSW entry
    (the gadget row plus `monomials` products of operands <= p/2; the two-factor form nests two of
    them and stays below this), and bound of the point-wise sum of two products."""
    f = forward()
    a = NORM                      # the normalised register: the only value that reaches an output unmultiplied
    d = a + prod(f[8]) + prod(f[8] + prod(f[8]))
    B = 0.5 + monomials * prod(0.5)
    return d, B, 2 * prod2(B, d)


def check_header(path):
    """(worst bound, final bound, normalisations per polynomial) of the schedule compiled into the kernels."""
    f = forward()
    assert f[-1] < LIMIT and forward(0.5)[-1] < LIMIT        # digits, and canonical inputs (key preparation)
    w, x, n = inverse(schedule_from_header(path), 2 * prod(f[-1]))
    # grouped blind rotation: the step's GGSW entry (wide) multiplies transformed digits of which
    # only two inputs of the last pass were normalised; the schedule's first stage normalises every
    # sum, so it takes any point-wise bound whose double stays below the limit
    d, B, pw = grouped_pointwise()
    wg, xg, ng = inverse(schedule_from_header(path, grouped=True), pw)
    # its outputs are canonicalised (exact below 3 p) or lifted, so 2.5 p is enough there
    assert d < 3.4 and B < 2.2 and 2 * pw < LIMIT and wg < LIMIT and max(xg) <= 2.5 and ng == n - 1
    # the one-bit kernels' schedule takes the grouped point-wise bound as well (the 256-thread grouped kernel uses it)
    w1, x1, _ = inverse(schedule_from_header(path), pw)
    assert w1 < LIMIT and max(x1) <= 2.0
    # factored key image: entry d (m1 + a)(m2 + b) + c with m <= prod(0.5) and a, b, d, c <= p/2: a smaller
    # entry than the plain form's, so everything above covers it.  (It would even take the transformed
    # digits un-normalised -- 2 * 6.2 p at the first inverse stage -- for 0.15 % of the time: not taken.)
    m = prod(0.5)
    Bf = prod(prod2(m + NORM, m + NORM)) + NORM
    assert Bf < 1.1 and Bf < B and 2 * prod2(Bf, f[-1]) * 2 < LIMIT
    # three key bits per step: B = A0 + m3 A1 with both halves factored; A1 = d1 (m1 + a1)(m2 + b1) + c1 meets
    # m3 un-normalised, A0 = d0 (m1 + a0)(m2 + b0) + c0 is added as it is -- still below the plain pairs' entry
    half = prod2(m + NORM, m + NORM)
    B3 = (prod(half) + NORM) + prod2(prod(half) + NORM, m)
    assert B3 < 1.8 and B3 < B
    return max(w, wg), max(x), n


def main():
    f = forward()
    print("forward, bound after each stage:", " ".join(f"{v:.2f}" for v in f))
    assert f[-1] < LIMIT
    pw = 2 * prod(f[-1])
    print(f"point-wise sum of two products: {pw:.2f} p   (limit {LIMIT:.2f} p)")
    import os
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe_amd", "csrc", "pbs_device.h")
    allr = set(range(8))
    for name, sch in (("three full normalisations after stages 2, 6, 10", {(0, 1): allr, (2, 0): allr, (3, 1): allr}),
                      ("schedule in pbs_device.h", schedule_from_header(hdr))):
        w, x, n = inverse(sch, pw, verbose=True)
        print(f"{name}: worst {w:.2f} p, final {max(x):.2f} p, {n} normalisations ->", "OK" if w < LIMIT else "OVER")
    if "--search" in sys.argv:
        best = None
        # per stage: normalise registers whose bound exceeds a threshold (one threshold per pass end,
        # plus optional mid-pass thresholds)
        grid = [1.25, 2.0, 3.3, 99.0]
        slots = [(0, 0), (0, 1), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2), (3, 0), (3, 1), (3, 2)]
        for ths in itertools.product(grid, repeat=len(slots)):
            x = [pw] * 8
            worst, norms, ok, sched = pw, 0, True, {}
            for pi, stages in enumerate([PASS4, PASS8, PASS8, PASS8]):
                for si, pairs in enumerate(stages):
                    for a, b in pairs:
                        s = x[a] + x[b]
                        worst = max(worst, s)
                        x[a], x[b] = s, prod(s)
                    if worst >= LIMIT:
                        ok = False
                        break
                    if (pi, si) in slots:
                        th = ths[slots.index((pi, si))]
                        regs = {r for r in range(8) if x[r] > th}
                        for r in regs:
                            x[r] = NORM
                        norms += len(regs)
                        if regs:
                            sched[(pi, si)] = regs
                if not ok:
                    break
                if pi < 3:
                    x = [max(x)] * 8
            if ok and max(x) <= 2.0 and (best is None or norms < best[0]):
                best = (norms, sched, worst, max(x))
        print("cheapest schedule found:", best)


if __name__ == "__main__":
    main()
