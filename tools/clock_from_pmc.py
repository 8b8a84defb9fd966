#!/usr/bin/env python3
"""Effective shader clock of every kernel of a `rocprofv3 --pmc GRBM_GUI_ACTIVE` run: busy cycles
(summed over the eight XCDs, hence / 8) over the dispatch's own duration.
    python tools/clock_from_pmc.py gpurun_out/<dir> [min_ms]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
min_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e5
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not files:
    raise SystemExit(f"no counter CSV under {d}")
agg = collections.defaultdict(lambda: [0.0, 0, 0])
for r in csv.DictReader(open(files[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    k = r["Kernel_Name"].split("(")[0][-48:]
    a = agg[k]
    a[0] += float(r["Counter_Value"])
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[2] += 1
for k, (c, dt, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if dt / n >= min_ns:
        print(f"{k:50s} launches {n:3d}  {dt / n / 1e6:9.3f} ms each  {c / 8 / dt:6.3f} GHz")
