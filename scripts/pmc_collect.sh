#!/bin/bash
# Dev tool (GPU box): PMC counters of the production TRI3 kernel on T1M, one rocprofv3 pass per counter group
# (kernel-trace only, as the pool requires).  Usage: bash scripts/pmc_collect.sh <outdir>
set -e
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o p -- \
      python3 $GRAFT_REPO_ROOT/scripts/prof_run.py --reps 20 --tile 0 > "$OUT/p$i.log" 2>&1
done
echo done
