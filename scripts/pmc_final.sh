#!/bin/bash
# Dev tool (GPU box): SQ counters of the shipped TRI3 kernel on T1M (bench.py's replayed leg), one rocprofv3 pass per
# counter group (kernel-trace only, as the pool requires).  Usage: bash scripts/pmc_final.sh <outdir>
# Writes <outdir>/pmc_sq_summary.txt: mean per launch of every counter over the launches of tri3_energy_pair_kernel.
set -e
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_final}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o p -- \
      python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --only-regime replayed --steps 50 > "$OUT/p$i.log" 2>&1 || true
  python3 - "$OUT/p$i/p_counter_collection.csv" >> "$OUT/pmc_sq_summary.txt" <<'PY'
import csv, sys, collections
try:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        if "tri3_energy_pair_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "launches", len(v), "mean", sum(v) / len(v))
except Exception as e:
    print("failed", sys.argv[1], e)
PY
  rm -rf "$OUT/p$i"
done
cat "$OUT/pmc_sq_summary.txt"
echo done
