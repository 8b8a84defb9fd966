#!/usr/bin/env python3
"""Sums the counters of tools/pmc_passes.sh per kernel and writes the tracked summary:
    python tools/pmc_summary.py gpurun_out/pmc_x profiles/r02/pmc_x.txt [profiles/r02/traffic.json flavour]
The blind-rotation kernel of the run is the one with the largest SQ_WAVE_CYCLES.  With the last two
arguments the memory-side traffic of that kernel's launch, (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950
tallies the 128-byte requests of wide loads at 64 bytes, so FETCH_SIZE is doubled), goes into the JSON
that bench.py reads, keyed by bootstrapping-key flavour and stamped with the hash of the kernel sources."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["kernels.hip", "pbs_device.h", "pbs_phases.h", "ks_mfma.hip", "ms_noise.h"]


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "fhe_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    per = {}      # kernel -> counter -> (sum, dispatches)
    span = {}     # kernel -> [sum of dispatch durations (ns), dispatches] of the pass that holds GRBM_GUI_ACTIVE
    for path in sorted(glob.glob(os.path.join(src, "*", "*counter_collection.csv")) +
                       glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv"))):
        seen = set()
        for row in csv.DictReader(open(path)):
            k, c = row["Kernel_Name"], row["Counter_Name"]
            d = per.setdefault(k, {}).setdefault(c, [0.0, 0])
            d[0] += float(row["Counter_Value"])
            if (k, c, row["Dispatch_Id"]) not in seen:
                seen.add((k, c, row["Dispatch_Id"]))
                d[1] += 1
                if c == "GRBM_GUI_ACTIVE" and os.sep + "sq" + os.sep in path:
                    sp = span.setdefault(k, [0, 0])
                    sp[0] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                    sp[1] += 1
    if not per:
        raise SystemExit(f"no counter CSVs under {src}")
    main_k = max(per, key=lambda k: per[k].get("SQ_WAVE_CYCLES", [0, 0])[0])
    lines = [f"rocprofv3 --pmc passes (tools/pmc_passes.sh, one run per pass) on `bench.py --steps 1 --warmup 0`,",
             f"kernel {main_k}; sums over all XCDs and over the dispatches listed (values per launch = sum / dispatches).",
             f"kernel sources {kernel_source_hash()}", ""]
    c = per[main_k]
    for name in sorted(c):
        s, n = c[name]
        lines.append(f"  {name:42s} {s:20.1f}   dispatches {n}   per launch {s / max(n, 1):18.1f}")
    # clock and issue rate of the pass that counted instructions (`sq`): busy cycles per XCD over the
    # dispatch's own duration, and vector-ALU wave-instructions x 4 cycles over the 1024 SIMDs' cycles
    extra = {}
    if main_k in span and "SQ_INSTS_VALU" in c and os.path.exists(os.path.join(src, "sq")):
        sq = {}
        for path in glob.glob(os.path.join(src, "sq", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if row["Kernel_Name"] == main_k:
                    sq[row["Counter_Name"]] = sq.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        ns, n = span[main_k]
        cycles = sq["GRBM_GUI_ACTIVE"] / 8 / n                 # per XCD and launch
        extra = {"clock_ghz": sq["GRBM_GUI_ACTIVE"] / 8 / ns, "cycles_per_launch": cycles,
                 "valu_wave_insts_per_launch": sq["SQ_INSTS_VALU"] / n,
                 "issue_frac": sq["SQ_INSTS_VALU"] / n * 4 / 1024 / cycles, "launch_ms_under_counters": ns / n / 1e6}
        lines += ["", f"  clock {extra['clock_ghz']:.3f} GHz ({cycles / 1e6:.1f} M cycles per launch of {ns / n / 1e6:.2f} ms); vector-ALU issue "
                      f"{extra['issue_frac']:.3f} of the cycles (SQ_INSTS_VALU x 4 / 1024 SIMDs / cycles)"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f = c["FETCH_SIZE"][0] / max(c["FETCH_SIZE"][1], 1)
        w = c["WRITE_SIZE"][0] / max(c["WRITE_SIZE"][1], 1)
        traffic = int((2 * f + w) * 1024)
        lines += ["", f"  memory-side bytes per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB = {traffic} B = {traffic / 1e9:.2f} GB"]
        if len(sys.argv) >= 5:
            tj, flavour = sys.argv[3], sys.argv[4]
            doc = json.load(open(tj)) if os.path.exists(tj) else {}
            doc[flavour] = {"bytes_per_launch_4096": traffic, "kernel": main_k, "kernel_sources": kernel_source_hash(),
                            "source": os.path.relpath(dst, ROOT), **extra}
            json.dump(doc, open(tj, "w"), indent=1, sort_keys=True)
    others = sorted(k for k in per if k != main_k)
    lines += ["", "other kernels of the run: " + ", ".join(o.split("(")[0] for o in others)]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
