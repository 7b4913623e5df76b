#!/bin/bash
# Collect rocprofv3 PMC counters for one bench step, one counter group per pass (kernel-trace only).
# usage: scripts/pmc_passes.sh <outdir> [extra bench args]
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o "$name" -- \
    python "$ROOT/bench.py" --steps 1 --warmup 1 --no-roofline --no-alt --cpu-baseline-clips 0 --traffic off --no-parity --no-aux "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
EXTRA=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
