#!/bin/bash
# Same-box A/B of library builds: tools/ab_bench.sh BATCH REPS lib1.so lib2.so ...
# Each build is a liblfhe.so made with other -D options (hipcc line of fhe_amd/build.py, output under ab/,
# which git ignores and gpurun uploads); the bench picks it up through LFHE_LIB.  Boxes differ by a few per
# cent, so only numbers from one call compare.
set -e
batch=$1; reps=$2; shift 2
for rep in $(seq "$reps"); do
  for lib in "$@"; do
    LFHE_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --steps 5 --warmup 1 --no-operators --batch "$batch" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', d['config']['batch_per_gpu'], round(d['value']), d['kernel_ms'], d['verified'])"
  done
done
