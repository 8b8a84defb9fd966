#!/bin/bash
# Counter passes on one blind-rotation launch of 4096 ciphertexts: tools/pmc_passes.sh OUTDIR [bench flags]
# One rocprofv3 --pmc run per pass (the profiler does not split counters over passes: at most 8 SQ,
# 4 TCC (FETCH_SIZE takes 3, WRITE_SIZE 2), 4 TCP, 2 TA, 2 TD, 2 GRBM counters each), each under its own
# time limit, the script stops at the first failure.  tools/pmc_summary.py turns the CSVs into the tracked
# summary and into profiles/rNN/traffic.json, which bench.py reads.
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
run() {
  tag=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" -d "$out/$tag" -o "$tag" --output-format csv -- \
    python3 bench.py --steps 1 --warmup 0 --no-alt --no-cpu-baseline --no-operators $BENCH_FLAGS > "$out/$tag.log" 2>&1
  echo "pass $tag done"
}
BENCH_FLAGS="$*"
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE TCC_REQ_sum
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TD_LOAD_WAVEFRONT_sum
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
