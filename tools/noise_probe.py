#!/usr/bin/env python3
"""Measured noise of the bootstrap path for each bootstrapping-key flavour, on the GPU, with the
library's own keys:

  output_noise       spread of  phase - delta * m  of a bootstrap's output (identity table)
  ks_noise           the same after a key switch of that output (small key): V_br + V_ks
  level5_input_noise the same for 5 x output (the parameter set's max_noise_level) after the key
                     switch: 25 V_br + V_ks -- this is what the modulus-switch noise reduction calls
                     its input variance
  decision_noise     what the blind rotation decodes: the level-5 ciphertext after noise reduction
                     and rounding to Z_2N, against half a message step (2^58)
  candidates         per sample, mean and standard deviation of the rounding error of the candidate
                     the noise reduction chose (public quantities of the ciphertext), as float32 in
                     units of 2^58 -- tools/pfail.py integrates the failure probability over them

usage: noise_probe.py [samples per key = 131072] [out prefix = gpurun_out/noise_probe]
writes <prefix>.json and <prefix>_candidates.npz
"""
from __future__ import annotations

import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FLAVOURS = (("q64", 1), ("prime", 2), ("grouped", 3), ("q64_grouped", 4), ("grouped3", 5), ("q64_grouped3", 6))
CHUNK = 16384
KEEP = 32768      # candidate records kept per key


def phase_err(cts, sk, msgs, dim):
    with np.errstate(over="ignore"):
        phase = cts[:, dim] - (cts[:, :dim] * sk[None, :]).sum(axis=1, dtype=np.uint64)
        return (phase - (msgs.astype(np.uint64) << np.uint64(59))).astype(np.int64).astype(np.float64)


def ms_round_error(a):
    """rounded - exact of the switch to Z_4096, signed, units of 2^-64 (fhe_amd/csrc/ms_noise.h)."""
    with np.errstate(over="ignore"):
        t = (a >> np.uint64(51)) + np.uint64(1) >> np.uint64(1)
        return ((t << np.uint64(52)) - a).astype(np.int64)


def main():
    from fhe_amd.native import Context, KEY_CLIENT
    N, n = 2048, 918
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    prefix = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", "noise_probe")
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    res, cand = {}, {}
    only = os.environ.get("NOISE_PROBE_ONLY")      # a comma-separated subset of the flavours, merged into an existing file
    if only and os.path.exists(prefix + ".json"):
        res = json.load(open(prefix + ".json"))
        cand = dict(np.load(prefix + "_candidates.npz"))
    for name, pid in FLAVOURS:
        if only and name not in only.split(","):
            continue
        ctx = Context(seed=bytes((13 * i + 1) & 0xFF for i in range(32)), params_id=pid)
        ctx.keygen()
        blob = ctx.export_keys(KEY_CLIENT)
        keys = np.frombuffer(blob[-(n + N) * 8:], dtype=np.uint64)
        lwe_sk, sk = keys[:n].copy(), keys[n:].copy()
        lut = ctx.make_lut(lambda x: x)[None]
        rng = np.random.default_rng(5)
        acc = {k: [] for k in ("in", "out", "ks", "l5", "dec")}
        means, sds, chosen_all = [], [], []
        done = 0
        while done < B:
            b = min(CHUNK, B - done)
            vals = rng.integers(0, 256, size=(b + 3) // 4)
            blocks = np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1)
                                     for v in vals])[:b]
            msgs = np.array([(int(vals[i // 4]) >> (2 * (i % 4))) & 3 for i in range(b)], dtype=np.uint64)
            out = ctx.pbs_batch(blocks, lut)
            acc["in"].append(phase_err(blocks, sk, msgs, N))
            acc["out"].append(phase_err(out, sk, msgs, N))
            acc["ks"].append(phase_err(ctx.keyswitch(out), lwe_sk, msgs, n))
            with np.errstate(over="ignore"):
                five = out * np.uint64(5)
            small5 = ctx.keyswitch(five)
            acc["l5"].append(phase_err(small5, lwe_sk, 5 * msgs, n))
            red, chosen = ctx.ms_noise_reduce(small5)
            e = ms_round_error(red).astype(np.float64)
            means.append((e[:, n] - e[:, :n].sum(axis=1) / 2) / 2.0 ** 58)
            sds.append(np.sqrt((e[:, :n] ** 2).sum(axis=1) / 4) / 2.0 ** 58)
            chosen_all.append(chosen)
            with np.errstate(over="ignore"):
                rounded = red + ms_round_error(red).astype(np.uint64)
            acc["dec"].append(phase_err(rounded, lwe_sk, 5 * msgs, n))
            done += b
            print(f"{name}: {done}/{B}", file=sys.stderr, flush=True)
        a = {k: np.concatenate(v) for k, v in acc.items()}
        chosen = np.concatenate(chosen_all)
        res[name] = {
            "bootstraps": B,
            "input_noise_log2_std": math.log2(a["in"].std()),
            "output_noise_log2_std": math.log2(a["out"].std()),
            "output_noise_log2_max": math.log2(np.abs(a["out"]).max()),
            "ks_noise_log2_std": math.log2(a["ks"].std()),
            "level5_input_noise_log2_std": math.log2(a["l5"].std()),
            "level5_input_noise_mean_over_std": float(a["l5"].mean() / a["l5"].std()),
            "decision_noise_log2_std": math.log2(a["dec"].std()),
            "decision_noise_log2_max": math.log2(np.abs(a["dec"]).max()),
            "ms_candidate_none_frac": float((chosen < 0).mean()),
            "ms_candidate_index_mean": float(np.where(chosen < 0, 0, chosen + 1).mean()),
            "ms_candidate_index_max": int(chosen.max()),
        }
        cand[name + "_mean"] = np.concatenate(means)[:KEEP].astype(np.float32)
        cand[name + "_sd"] = np.concatenate(sds)[:KEEP].astype(np.float32)
        ctx.close()
    with open(prefix + ".json", "w") as f:
        json.dump(res, f, indent=1)
    np.savez_compressed(prefix + "_candidates.npz", **cand)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
