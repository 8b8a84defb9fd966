#!/usr/bin/env python3
"""Failure probability of a bootstrap per bootstrapping-key flavour, computed from measured noise.

The parameter set (PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128, sdk/relayer/src/tfhe.ts:14-19)
promises log2 p_fail <= -128; its ClientKey fixture (sdk/relayer/src/test/keys/privateKey.bin) stores
log2_p_fail = -129.15 and the constants of the modulus-switch noise reduction that enforces it:
bound 2^58 (half a message step), r_sigma 13.18, input variance 2.63e-7 (torus^2).

Model.  A bootstrap decodes  x = e_in + e_ms  where
  e_in  is the noise of the key-switched input at the set's max_noise_level = 5: 25 V_br + V_ks,
        Gaussian by the central limit theorem (thousands of independent terms); MEASURED on the GPU by
        tools/noise_probe.py as `level5_input_noise` (5 x a bootstrap output, key-switched, decrypted
        under the small key);
  e_ms  is the rounding error of the switch to Z_2N under the binary key: for the candidate the noise
        reduction picked, mean m = e_b - sum e_i / 2 and variance v = sum e_i^2 / 4 are public
        functions of the ciphertext (fhe_amd/csrc/ms_noise.h); the probe records (m, sqrt v) of the
        chosen candidate for every sample.
It fails when |x| > 2^58.  Per ciphertext  p = Q((2^58 - m) / s) + Q((2^58 + m) / s),  s^2 = V_in + v;
p_fail of a key flavour is the mean of p over the recorded candidates.  The noise reduction accepts
a candidate when |m| + r_sigma sqrt(V_pub + v) <= 2^58 with V_pub the set's published input
variance: a flavour whose measured V_in is at most V_pub inherits the set's own guarantee, one
above it is priced by the formula.

usage: pfail.py [profiles/r03/noise_probe.json]   (the .npz of candidates sits next to it)
"""
from __future__ import annotations

import json
import math
import os
import sys

import numpy as np
from scipy.special import log_ndtr, logsumexp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEFAULT_PROBE = os.path.join(ROOT, "profiles", "r03", "noise_probe.json")
LN2 = math.log(2.0)


def published():
    from fhe_amd.tfhers_wire import parse_client_key
    prm = parse_client_key(open(os.path.join(ROOT, "tests", "golden", "privateKey.bin"), "rb").read())["params"]
    return {"log2_p_fail": prm["log2_p_fail"], "r_sigma": prm["ms_r_sigma_factor"], "bound": prm["ms_bound"],
            "input_variance_log2": math.log2(prm["ms_input_variance"]) + 128, "max_noise_level": prm["max_noise_level"],
            "lwe_dim": prm["lwe_dim"], "poly_size": prm["poly_size"], "ks_base_log": prm["ks_base_log"],
            "ks_level": prm["ks_level"], "lwe_noise_log2": prm["lwe_noise_log2"]}


def ks_variance_log2(pub) -> float:
    """Formula value of the key-switch noise (cross-check of the measured one): N l E[d^2] Var(TUniform(b))
    for the key noise + N/2 times the variance of what the decomposition rounds away."""
    B, l, N, b = 1 << pub["ks_base_log"], pub["ks_level"], pub["poly_size"], pub["lwe_noise_log2"]
    var_key = (2.0 ** (2 * b + 1) + 1) / 6
    d2 = (B * B + 2) / 12
    cut = 2.0 ** (64 - pub["ks_base_log"] * l)          # weight of the last digit kept
    return math.log2(N * l * d2 * var_key + N / 2 * cut * cut / 12)


def blind_rotation_variance_log2(bits: int, n: int = 918, N: int = 2048, base_log: int = 23, glwe_noise_log2: int = 17) -> float:
    """Formula value of a bootstrap's output variance (2^64 torus scale) for `bits` key bits per step,
    one decomposition level, k = 1.  Per external product: the key noise of every GGSW in the step
    ((k+1) N digits of variance (B^2 + 2) / 12 times Var TUniform) and what the decomposition rounds
    away ((q / B)^2 / 12 per coefficient, seen through the key: 1 + N / 2).  One bit per step (the CMux
    acc + G [.] ((X^a - 1) acc)): n products of one GGSW, the rounding counted where the key bit is 1
    (half of them).  Grouped, accumulator replaced by the product: ceil(n / bits) products of 2^bits
    GGSWs, the rounding in every one."""
    var_key = (2.0 ** (2 * glwe_noise_log2 + 1) + 1) / 6
    B = 2.0 ** base_log
    key = 2 * N * (B * B + 2) / 12 * var_key
    rounding = (2.0 ** (64 - base_log)) ** 2 / 12 * (1 + N / 2)
    if bits == 1:
        return math.log2(n * (key + rounding / 2))
    steps = -(-n // bits)
    return math.log2(steps * ((1 << bits) * key + rounding))


def log2_pfail(v_in_log2: float, mean58: np.ndarray, sd58: np.ndarray, in_mean58: float = 0.0) -> float:
    """mean and sd of the rounding error in units of 2^58; v_in (variance about the mean) on the 2^64
    torus scale; in_mean58: mean of the input noise in units of 2^58 (for one key the key-switch noise
    has a small fixed offset, minus half the sum of the key's noise terms, because its digits lie in
    [-8, 7])."""
    s = np.sqrt(2.0 ** (v_in_log2 - 116) + sd58.astype(np.float64) ** 2)
    m = mean58.astype(np.float64) + in_mean58
    lp = np.logaddexp(log_ndtr(-(1 - m) / s), log_ndtr(-(1 + m) / s))
    return float((logsumexp(lp) - math.log(len(lp))) / LN2)


def analyse(probe_path: str = DEFAULT_PROBE) -> dict:
    pub = published()
    probe = json.load(open(probe_path))
    cand = np.load(os.path.splitext(probe_path)[0] + "_candidates.npz")
    out = {"published": pub, "ks_variance_log2_formula": ks_variance_log2(pub), "keys": {}}
    ref = probe["q64"]["output_noise_log2_std"]
    for name, rec in probe.items():
        v_br = 2 * rec["output_noise_log2_std"]
        v_in = 2 * rec["level5_input_noise_log2_std"]
        v_ks = math.log2(max(2.0 ** (2 * rec["ks_noise_log2_std"]) - 2.0 ** v_br, 1.0))
        mean, sd = cand[name + "_mean"], cand[name + "_sd"]
        mu = rec["level5_input_noise_mean_over_std"] * 2.0 ** (rec["level5_input_noise_log2_std"] - 58)
        out["keys"][name] = {
            "samples": rec["bootstraps"],
            "v_br_log2": v_br, "v_br_ratio_to_q64": 2.0 ** (v_br - 2 * ref),
            "v_br_log2_formula": blind_rotation_variance_log2(3 if name.endswith("3") else 2 if "grouped" in name else 1),
            "v_ks_log2_measured": v_ks,
            "v_in_level5_log2": v_in,
            "v_in_over_published": 2.0 ** (v_in - pub["input_variance_log2"]),
            "input_mean_over_bound": mu,
            "log2_p_fail": log2_pfail(v_in, mean, sd, mu),
            # the same candidates priced with the set's own input variance: what the published constants give
            "log2_p_fail_at_published_variance": log2_pfail(pub["input_variance_log2"], mean, sd),
            # a candidate sitting exactly on the acceptance bound with the typical rounding variance
            "log2_p_fail_worst_accepted": float(
                (log_ndtr(-pub["r_sigma"] * math.sqrt((2.0 ** (pub["input_variance_log2"] - 116) + np.median(sd) ** 2) /
                                                     (2.0 ** (v_in - 116) + np.median(sd) ** 2))) + LN2) / LN2),
            "decision_noise_log2_std": rec["decision_noise_log2_std"],
            "decision_noise_log2_max": rec["decision_noise_log2_max"],
        }
    return out


def main():
    res = analyse(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_PROBE)
    pub = res["published"]
    print(f"published: log2 p_fail {pub['log2_p_fail']:.2f}, input variance 2^{pub['input_variance_log2']:.3f}, "
          f"r_sigma {pub['r_sigma']:.4f}; key-switch variance by formula 2^{res['ks_variance_log2_formula']:.3f}")
    print(f"{'key':12s} {'V_br':>8s} {'/q64':>6s} {'V_ks':>8s} {'V_in(5)':>8s} {'/publ.':>7s} {'log2 p_fail':>12s} "
          f"{'at publ. V':>11s} {'worst acc.':>11s}")
    for name, k in res["keys"].items():
        print(f"{name:12s} {k['v_br_log2']:8.3f} {k['v_br_ratio_to_q64']:6.3f} {k['v_ks_log2_measured']:8.3f} "
              f"{k['v_in_level5_log2']:8.3f} {k['v_in_over_published']:7.4f} {k['log2_p_fail']:12.2f} "
              f"{k['log2_p_fail_at_published_variance']:11.2f} {k['log2_p_fail_worst_accepted']:11.2f}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
