#!/bin/bash
# scripts/profile.sh -- rocprofv3 over the roofline configuration (BASELINE.json configs[2]); the
# MI355X counterpart of the reference's run.sh / run_slurm.sh (VTune over ./nbody.x).
#   pass 1: --kernel-trace --stats            per-kernel time
#   pass 2..5: --pmc (separate passes)        HBM bytes (FETCH_SIZE, WRITE_SIZE), VALU issue, clock
# Counters are collected in their own runs (never combined with sys/hip/hsa traces).
# Usage (on the GPU box, from the repo root):  bash scripts/profile.sh [outdir] [n] [steps] [precision] [reference|tree]
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
N=${2:-262144}
STEPS=${3:-10}
PREC=${4:-32}
ORDER=${5:-reference}   # summation order profiled (an explicit order keeps bench.py from also timing the other one)
mkdir -p "$OUT"
BENCH="python3 bench.py --steps $STEPS --warmup 2 --n $N --precision $PREC --order $ORDER --cpu-baseline none"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1 || echo "stats pass failed" >&2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || echo "fetch pass failed" >&2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || echo "write pass failed" >&2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || echo "sq pass failed" >&2
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm" -- $BENCH > "$OUT/pmc_grbm.log" 2>&1 || echo "grbm pass failed" >&2
find "$OUT" -name "*.csv" | head -40
