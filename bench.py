#!/usr/bin/env python3
"""Batched programmable-bootstrap benchmark (BASELINE.json config 2).

One step = one pass of the hot path (key switch -> modulus switch + blind rotation + sample
extraction) over a batch of 4096 independent ciphertext blocks per GPU, inputs and keys resident
in HBM.  With N GPUs every rank bootstraps its own 4096 blocks (weak scaling); the server keys are
generated on rank 0 and replicated with one RCCL broadcast of their device images, there is no
collective on the data path.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
N = 2048
# algorithmic work of one bootstrap at n=918, k=1, N=2048, l=1 (single-modulus count):
# 3672 length-2048 transforms * 11264 butterflies + 918*4*2048 point-wise multiply-accumulates
MODMUL_PER_PBS = 3672 * 11264 + 918 * 4 * 2048
# with the key bits in pairs: 459 steps of 4 transforms, and per transform-domain point 12 products
# for the step's GGSW (3 key rows x 4 polynomials) + 4 for the external product itself
MODMUL_PER_PBS_GROUPED = 459 * (4 * 11264 + 16 * 2048)
# with the key image in factored form the entry sums cost 2 products each instead of 3: 8 + 4 per point
MODMUL_PER_PBS_GROUPED_FACTORED = 459 * (4 * 11264 + 12 * 2048)
# register-resident calibration on MI355X (profiles/r01/valu_rates_8wg_per_cu.txt): 4748 G/s exact
# f64 butterflies (modular multiply + add + sub) per prime.  Over q = 2^64 a logical modular product
# costs one butterfly in each of two primes; over the prime ring modulus it costs one.
PEAK_GBFLY_PER_PRIME = 4748.0
# memory-side traffic of one 4096-ciphertext blind-rotation launch: read from the tracked counter summary
# (tools/pmc_passes.sh + tools/pmc_summary.py: separate rocprofv3 --pmc passes, (2 * FETCH_SIZE +
# WRITE_SIZE) * 1024, Infinity-Cache hits included).  The summary is stamped with a hash of the kernel
# sources; a build from other sources reports null instead of a stale number.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03", "traffic.json")


def measured_counters(flavour: str, batch: int) -> dict:
    """The tracked counter summary of this build's blind-rotation kernel ({} when there is none or it
    was taken on other kernel sources): bytes_per_launch_4096, clock_ghz, issue_frac, source, ..."""
    if batch != BATCH or not os.path.exists(TRAFFIC_FILE):
        return {}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import kernel_source_hash
        rec = json.load(open(TRAFFIC_FILE)).get(flavour)
        if not rec or rec.get("kernel_sources") != kernel_source_hash():
            return {}
        return rec
    except (OSError, ValueError, KeyError, ImportError):
        return {}


# algorithmic bytes of one batch launch: both keys once + one ciphertext in and out per bootstrap
BSK_BYTES, KSK_BYTES, CT_BYTES = 60162048, 60227584, 16392
# device image of the bootstrapping key (f64 residues in the transform domain)
KEY_IMAGE_BYTES = {"q64": 918 * 131072, "q64_grouped": 1836 * 131072, "q64_grouped3": 2448 * 131072, "prime": 918 * 65536,
                   "grouped": 1836 * 65536, "grouped3": 2448 * 65536}
CTS_PER_WORKGROUP = {"q64": 1, "q64_grouped": 2, "q64_grouped3": 2, "prime": 1, "grouped": 2, "grouped3": 2}
# three key bits per step: 306 steps of 4 transforms, per point 20 products for the step's GGSW entries (five each,
# factored) + 4 for the external product
MODMUL_PER_PBS_GROUPED3 = 306 * (4 * 11264 + 24 * 2048)


# bootstrapping-key flavour -> (parameter set id, label, kernel, VALU peak in butterflies/s per GPU)
FLAVOURS = {
    "q64": (1, "2^64", "k_blind_rotate"),
    "q64_grouped": (4, "2^64, key bits in pairs", "k_blind_rotate_g64_pair<true, 2>"),
    "q64_grouped3": (6, "2^64, key bits in triples", "k_blind_rotate_g64_pair<true, 2, false, 3>"),
    "prime": (2, "703687441727489 (NTT prime)", "k_blind_rotate_q"),
    "grouped": (3, "703687441727489 (NTT prime), key bits in pairs", "k_blind_rotate_q2_pair<false, true>"),
    "grouped3": (5, "703687441727489 (NTT prime), key bits in triples", "k_blind_rotate_q3<2>"),
}


def cpu_baseline(seconds_budget: float = 12.0, flavour: str = "prime"):
    """The CPU oracle (own port; the reference engine is not available offline) timed on the host
    cores of this box on a bounded sample of the same workload."""
    from oracle.oracle import Oracle, Rng, seed_bytes
    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    orc = Oracle()
    lwe_sk, glwe_sk, bsk, ksk = orc.gen_keys(seed_bytes(9))
    orc.set_ms(orc.gen_ms_zeros(lwe_sk, seed_bytes(9), orc.ms_params().zeros_count))   # same pipeline as the GPU's
    if flavour == "grouped3":
        pb = orc.prepare_bsk_qp_grouped(orc.gen_bsk_qp_grouped(lwe_sk, glwe_sk, seed_bytes(9), 3), 3)
    elif flavour == "grouped":
        pb = orc.prepare_bsk_qp_grouped(orc.gen_bsk_qp_grouped(lwe_sk, glwe_sk, seed_bytes(9)))
    elif flavour == "q64_grouped":
        pb = orc.prepare_bsk_grouped(orc.gen_bsk_grouped(lwe_sk, glwe_sk, seed_bytes(9)))
    elif flavour == "q64_grouped3":
        pb = orc.prepare_bsk_grouped(orc.gen_bsk_grouped(lwe_sk, glwe_sk, seed_bytes(9), 3), 3)
    elif flavour == "prime":
        pb = orc.prepare_bsk_qp(orc.gen_bsk_qp(lwe_sk, glwe_sk, seed_bytes(9)))
    else:
        pb = orc.prepare_bsk(bsk)
    pbs_batch = orc.pbs_batch if flavour.startswith("q64") else orc.pbs_batch_qp
    r = Rng(seed_bytes(10))
    lut = orc.make_lut(lambda x: x)
    probe = np.stack([orc.encrypt_block(glwe_sk, int(r.u64() % 16), r) for _ in range(threads)])
    t0 = time.perf_counter()
    pbs_batch(ksk, pb, probe, lut[None], None, nthreads=threads)
    per_round = time.perf_counter() - t0
    rounds = max(1, min(96, int(seconds_budget / max(per_round, 1e-3))))      # about twelve seconds of all cores
    cts = np.stack([orc.encrypt_block(glwe_sk, int(r.u64() % 16), r) for _ in range(threads * rounds)])
    t0 = time.perf_counter()
    out = pbs_batch(ksk, pb, cts, lut[None], None, nthreads=threads)
    dt = time.perf_counter() - t0
    assert all(orc.decrypt_block(glwe_sk, out[i]) == orc.decrypt_block(glwe_sk, cts[i]) for i in range(0, len(cts), 5))
    t0 = time.perf_counter()
    pbs_batch(ksk, pb, cts[:3], lut[None], None, nthreads=1)
    one_core = 3 / (time.perf_counter() - t0)
    return {"value": len(cts) / dt, "unit": "PBS/s", "cores": threads, "kind": "port", "value_1core": one_core,
            "sample": f"{len(cts)} bootstraps of the same workload (n=918, N=2048) on {threads} threads, 3 on one; "
                      "own CPU restatement (oracle C port with OpenMP): the reference engine is unavailable offline"}


class GpuBackend:
    """What the benchmark needs from the machine: the device of a rank, a context with replicated
    keys, a device-wide wait.  The HIP library is the only compute path of this default; the CPU
    tests drive the same control flow (ranks, shards, collectives, verification, exit status) with a
    plaintext stand-in from tests/ injected in its place."""
    collective = "nccl"

    def device(self, local: int):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("no GPU: the HIP path is the only path")
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def context(self, dist, rank, world, dev, seed, params_id):
        from fhe_amd.sharding import replicated_context
        return replicated_context(dist, rank, world, dev, seed, params_id)

    def synchronize(self):
        import torch
        torch.cuda.synchronize()


# first reference vector of the euint64 multiplication (tests/fhevm-suite/e2e/test/fhevmOperations/
# fhevmOperations6.ts:647-660)
MUL_KAT = (4293652672, 4293367238, 18434227713315959936)


def mul_latency(ctx, reps: int = 3):
    """One euint64 multiplication, operands and result in host memory (the shape of POST /evaluate):
    best of `reps` after a warm-up that builds and uploads the schedule.  Returns (ms, bootstraps, ok)."""
    a, b = ctx.encrypt(MUL_KAT[0], 64), ctx.encrypt(MUL_KAT[1], 64)
    ctx.evaluate("mul", 64, a, b)
    best, ok = None, True
    n0 = ctx.pbs_count()
    for _ in range(reps):
        t0 = time.perf_counter()
        out = ctx.evaluate("mul", 64, a, b)
        dt = time.perf_counter() - t0
        ok = ok and ctx.decrypt(out) == MUL_KAT[2]
        best = dt if best is None else min(best, dt)
    return best * 1e3, (ctx.pbs_count() - n0) // reps, ok


def main(argv=None, backend=None):
    backend = backend or GpuBackend()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bsk-modulus", choices=list(FLAVOURS), default="grouped3",
                    help="bootstrapping key of the timed leg: over 2^64 (two-prime transform + CRT lift), over the NTT "
                         "prime (one transform per polynomial), or over the prime with the key bits taken two per "
                         "blind-rotation step, or three (the default)")
    ap.add_argument("--no-alt", action="store_true", help="skip the legs with the other bootstrapping keys")
    ap.add_argument("--no-operators", action="store_true", help="skip the euint64 multiplication latency probe")
    args = ap.parse_args(argv)

    import torch
    from fhe_amd.sharding import max_over_ranks

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = backend.device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend.collective == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend.collective, rank=rank, world_size=world)

    seed = bytes((7 * i + 3) & 0xFF for i in range(32))

    def make_context(flavour: str):
        """Context with keys on every rank: generated on rank 0, device images broadcast once."""
        return backend.context(dist, rank, world, dev, seed, FLAVOURS[flavour][0])

    B = args.batch
    # 16 distinct tables, table j on block i when i % 16 == j (table 0 is the identity)
    f = [(lambda x, j=j: ((2 * j + 1) * x + j) % 16) for j in range(16)]
    idx = (np.arange(B) % 16).astype(np.uint32)

    def run_leg(flavour: str, steps: int, warmup: int, operators: bool = False):
        """K timed passes over this rank's B blocks; returns (elapsed s, ks ms, br ms, verified, mul probe)."""
        ctx, keep = make_context(flavour)
        # synthetic inputs: batch blocks = batch/4 euint8 values, fresh encryptions, distinct per rank
        rng = np.random.default_rng(1234 + rank)
        vals = rng.integers(0, 256, size=(B + 3) // 4)
        blocks = np.concatenate([np.frombuffer(ctx.encrypt(int(v), 8)[16:], dtype=np.uint64).reshape(4, N + 1)
                                 for v in vals])[:B]
        msgs = np.array([(int(vals[i // 4]) >> (2 * (i % 4))) & 3 for i in range(B)])
        luts = np.stack([ctx.make_lut(g) for g in f])
        d_in = torch.from_numpy(blocks.view(np.int64)).to(dev)
        d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
        d_idx = torch.from_numpy(idx.view(np.int32)).to(dev)
        d_out = torch.zeros_like(d_in)
        backend.synchronize()

        def step(iters):
            return ctx.pbs_batch_device_timed(B, d_in.data_ptr(), d_luts.data_ptr(), d_idx.data_ptr(),
                                              d_out.data_ptr(), iters)

        if warmup > 0:
            step(warmup)
        ctx.sync()
        backend.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        ks_ms, br_ms = step(steps)   # K back-to-back passes, each kernel bracketed by HIP events
        ctx.sync()
        backend.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            elapsed = max_over_ranks(dist, elapsed, dev)
        # correctness of what was timed: every sampled output decrypts to f(m)
        out = d_out.cpu().numpy().view(np.uint64)
        sample = list(range(0, B, max(1, B // 128)))
        got = ctx.decrypt_blocks(out[sample])
        ok = all(g == f[idx[i]](int(msgs[i])) % 16 for g, i in zip(got, sample))
        if dist is not None:
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        # outside the timed region: the latency of one euint64 multiplication on this rank's GPU
        # (rank 0 reports; no collective involved)
        probe = mul_latency(ctx) if operators and rank == 0 else None
        if probe is not None:
            ok = ok and probe[2]
        factored = bool(ctx.bsk_factored()) if hasattr(ctx, "bsk_factored") else False
        ctx.close()
        del keep
        return elapsed, ks_ms, br_ms, ok, probe, factored

    flavour = args.bsk_modulus
    elapsed, ks_ms, br_ms, ok, probe, factored = run_leg(flavour, args.steps, args.warmup, operators=not args.no_operators)
    alts = []
    if not args.no_alt:
        a_steps = max(1, min(args.steps, 3))
        for other in FLAVOURS:
            if other == flavour:
                continue
            # the fastest key with every modulus as published also reports the latency of one multiplication
            a_el, a_ks, a_br, a_ok, a_probe, a_fact = run_leg(other, a_steps, 1,
                                                              operators=other == "q64_grouped3" and not args.no_operators)
            ok = ok and a_ok
            alts.append({"flavour": other, "bsk_modulus": FLAVOURS[other][1], "kernel": FLAVOURS[other][2],
                         "value": B * a_steps * world / a_el, "unit": "PBS/s", "steps": a_steps,
                         "kernel_ms": {"keyswitch_and_ms_noise_reduction": a_ks, "blind_rotate": a_br},
                         "key_image": "factored" if a_fact else "as generated", "verified": a_ok})
            if a_probe is not None:
                alts[-1]["euint64_mul_latency_ms"] = a_probe[0]

    if rank == 0:
        total = B * args.steps * world
        value = total / elapsed
        br_s = br_ms * 1e-3
        # priced with the products the kernel really executes (12 per point with the factored key image, 16 without)
        modmul = MODMUL_PER_PBS
        if flavour in ("grouped3", "q64_grouped3"):
            modmul = MODMUL_PER_PBS_GROUPED3
        elif flavour in ("grouped", "q64_grouped"):
            modmul = MODMUL_PER_PBS_GROUPED_FACTORED if factored else MODMUL_PER_PBS_GROUPED
        achieved = B * modmul / br_s / 1e9
        peak = PEAK_GBFLY_PER_PRIME / 2 if flavour.startswith("q64") else PEAK_GBFLY_PER_PRIME
        algo_bytes = BSK_BYTES + KSK_BYTES + 2 * CT_BYTES * B
        pmc = measured_counters(flavour, B)
        traffic, traffic_source = pmc.get("bytes_per_launch_4096"), pmc.get("source")
        classic = B * MODMUL_PER_PBS / br_s / 1e9 / peak
        unit_note = (f"achieved = {B} bootstraps x modmul_per_pbs / blind-rotation launch time; modmul_per_pbs counts the modular "
                     "products this kernel's algorithm executes")
        if flavour in ("grouped3", "q64_grouped3"):
            unit_note += (f": {MODMUL_PER_PBS_GROUPED3} = 306 steps x (4 transforms x 11264 butterflies + 24 products x 2048 points) "
                          f"with the key bits in triples, where the classic one-bit-per-step count ({MODMUL_PER_PBS}) would read "
                          f"frac = {classic:.3f}")
        elif flavour in ("grouped", "q64_grouped"):
            unit_note += (f": {MODMUL_PER_PBS_GROUPED_FACTORED if factored else MODMUL_PER_PBS_GROUPED} = 459 steps x (4 transforms x "
                          f"11264 butterflies + {12 if factored else 16} products x 2048 points) with the key bits in pairs, where the "
                          f"classic one-bit-per-step count (918 x 4 transforms + 918 x 4 x 2048 products = {MODMUL_PER_PBS}) would "
                          f"read frac = {classic:.3f}")
        unit_note += (f"; peak = {PEAK_GBFLY_PER_PRIME:.0f} G exact f64 butterflies/s per prime, calibrated register-resident at "
                      "2.39 GHz (profiles/r01/valu_rates_8wg_per_cu.txt)" + (", halved: two primes per product over 2^64"
                                                                              if flavour.startswith("q64") else ""))
        res = {
            "metric": "programmable bootstraps/sec (node)",
            "value": value,
            "unit": "PBS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"batched PBS microbench: {B} independent euint8 blocks per GPU, "
                                   "PARAM_MESSAGE_2_CARRY_2_KS_PBS_TUNIFORM_2M128 (n=918, N=2048), 16 distinct LUTs",
                       "bsk_modulus": FLAVOURS[flavour][1], "key_image": "factored" if factored else "as generated",
                       "batch_per_gpu": B, "parallelism": f"batch-shard x{world}, keys broadcast once"},
            "verified": ok,
            "kernel_ms": {"keyswitch_and_ms_noise_reduction": ks_ms, "blind_rotate": br_ms},
            "roofline": {"bound": "valu", "kernel": FLAVOURS[flavour][2], "achieved": achieved, "peak": peak,
                         "unit": "Gmodmul/s", "frac": achieved / peak, "modmul_per_pbs": modmul,
                         "unit_note": unit_note, "frac_with_classic_unit": classic,
                         # from the tracked rocprofv3 --pmc summary of this build (null: none for these sources)
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_ratio": traffic / algo_bytes if traffic else None,
                         # what leaves the L2s is the key image, once per XCD and per round of co-resident
                         # workgroups at best (the image does not fit the 4 MB L2 of an XCD, which holds about
                         # eight steps of it, so workgroups that drift further apart fetch a step again; it does
                         # fit the 256 MB memory-side cache, so little of this reaches HBM)
                         "traffic_note": (f"{traffic / KEY_IMAGE_BYTES[flavour]:.1f} sweeps of the {KEY_IMAGE_BYTES[flavour] >> 20} MiB "
                                          f"key image (8 XCDs x {B // CTS_PER_WORKGROUP[flavour] // 512:d} rounds of 2 workgroups per CU "
                                          f"= {8 * (B // CTS_PER_WORKGROUP[flavour] // 512):d} if every round shared each fetch); "
                                          f"{traffic / br_s / 1e9:.0f} GB/s; traffic_ratio is against keys once + ciphertexts in and out")
                                         if traffic and flavour in ("grouped", "grouped3") else None,
                         "issue_frac": pmc.get("issue_frac"), "clock_ghz": pmc.get("clock_ghz"),
                         "valu_wave_insts_per_launch": pmc.get("valu_wave_insts_per_launch"),
                         "hbm": {"achieved": algo_bytes / ((ks_ms + br_ms) * 1e-3) / 1e9, "peak": 8000.0,
                                 "unit": "GB/s", "algorithmic_bytes": algo_bytes},
                         # key words the workgroups request per launch (served by L2): every workgroup streams
                         # the whole device key image once, and the grouped kernels put two ciphertexts in a
                         # workgroup at this batch size
                         "key_stream": {"bytes": B // CTS_PER_WORKGROUP[flavour] * KEY_IMAGE_BYTES[flavour],
                                        "achieved": B // CTS_PER_WORKGROUP[flavour] * KEY_IMAGE_BYTES[flavour] / br_s / 1e9,
                                        "ciphertexts_per_workgroup": CTS_PER_WORKGROUP[flavour], "unit": "GB/s"}},
        }
        if probe is not None:
            # second half of the headline metric: one euint64 multiplication, host buffers in and out
            res["euint64_mul_latency_ms"] = probe[0]
            res["mul_pbs"] = probe[1]
        if alts:
            res["other_bootstrapping_keys"] = alts
            # the parameter set with every modulus as the reference names it (2^64): the one-bit key any
            # 2^64 engine can load, and this library's own key with the bits in pairs
            for name, field in (("q64", "value_q64_key"), ("q64_grouped", "value_q64_grouped_key"),
                                ("q64_grouped3", "value_q64_grouped3_key")):
                leg = [a for a in alts if a["flavour"] == name]
                if leg:
                    res[field] = leg[0]["value"]
        if flavour == "q64":
            res["value_q64_key"] = value
        if flavour == "q64_grouped":
            res["value_q64_grouped_key"] = value
        if flavour == "q64_grouped3":
            res["value_q64_grouped3_key"] = value
        # `value` is on this library's fastest server key (config.bsk_modulus says which); the same workload with
        # every modulus exactly as the parameter set is published:
        res["value_param_set_as_published"] = {
            "one_bit_key_any_2^64_engine_can_load": res.get("value_q64_key"),
            "own_key_with_the_key_bits_in_pairs": res.get("value_q64_grouped_key"),
            "own_key_with_the_key_bits_in_triples": res.get("value_q64_grouped3_key"), "unit": "PBS/s",
            "euint64_mul_latency_ms_with_the_key_bits_in_triples": next(
                (a.get("euint64_mul_latency_ms") for a in alts if a["flavour"] == "q64_grouped3"),
                res.get("euint64_mul_latency_ms") if flavour == "q64_grouped3" else None)}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(flavour=flavour)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("verification failed")


if __name__ == "__main__":
    main()
