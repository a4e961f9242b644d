#!/bin/bash
# scripts/profile_slice.sh [outdir] [n] [own] [before-variant: cxx|asm2] -- rocprofv3 over ONE rank's slice of a reference-order run (what a rank of a multi-GPU job
# launches per step), before (one body per lane, compiled loop with plain VALU ops: round 3's shape) and after (the two-j-records-per-
# packed-operation loop, round 4).  The counters that say why small slices are slow are occupancy, not traffic: waves resident and busy
# (SQ_WAVES, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES), VALU issue (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU), waiting (SQ_WAIT_INST_ANY), and the
# wall clock of the GPU (GRBM_GUI_ACTIVE).  Counters in their own passes, never with sys/hip/hsa traces; the program directly after `--`.
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof_slice}
N=${2:-262144}
OWN=${3:-65536}
BEFORE=${4:-cxx}   # round 3's shape for this slice: cxx (one body per lane, compiled loop) up to 65536 bodies, asm2 (two per lane, plain asm loop) above
mkdir -p "$OUT"
for V in $BEFORE auto; do
  RUN="python3 tools/slice_run.py $N $OWN $V 8"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${V}_stats" -- $RUN > "$OUT/${V}_stats.log" 2>&1 || echo "$V stats pass failed" >&2
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$OUT/${V}_pmc_sq" -- $RUN > "$OUT/${V}_pmc_sq.log" 2>&1 || echo "$V sq pass failed" >&2
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/${V}_pmc_grbm" -- $RUN > "$OUT/${V}_pmc_grbm.log" 2>&1 || echo "$V grbm pass failed" >&2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${V}_pmc_fetch" -- $RUN > "$OUT/${V}_pmc_fetch.log" 2>&1 || echo "$V fetch pass failed" >&2
done
find "$OUT" -name "*.csv" | head -60
