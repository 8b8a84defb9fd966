#!/bin/bash
# kernel times of one bench run, key-switch kernels and the rest: tools/ks_prof.sh OUTDIR
set -e
out=$1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out" -o k --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline --no-operators > "$out.json" 2> "$out.err"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:44].ljust(46), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
