#!/bin/bash
# throughput of the default bench leg against the launch size
for b in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --steps 2 --warmup 1 --batch "$b" 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['config']['batch_per_gpu'], round(d['value']), d['kernel_ms'])" || exit 1
done
