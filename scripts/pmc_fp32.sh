#!/bin/bash
# Dev tool (GPU box): SQ counters of the fp32-arithmetic kernel against the fp64-arithmetic float-row instance on an fp32 T1M
# model (scripts/fp32_kernel_timing.py), one rocprofv3 pass per counter group (kernel-trace only, as the pool requires).
# Writes <outdir>/pmc_fp32_summary.txt: mean per launch of every counter, per kernel.
set -e
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_fp32}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -o p -- \
      python3 $GRAFT_REPO_ROOT/scripts/fp32_kernel_timing.py > "$OUT/p$i.log" 2>&1 || true
  python3 - "$OUT/p$i/p_counter_collection.csv" >> "$OUT/pmc_fp32_summary.txt" <<'PY'
import csv, sys, collections
try:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        n = r["Kernel_Name"]
        if "tri3_energy_pair" in n:
            k = "fp32_arithmetic" if "pair_f32_kernel" in n else ("fp64_arith_float_rows" if "float" in n else "fp64_model")
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "launches", len(v), "mean", sum(v) / len(v))
except Exception as e:
    print("failed", sys.argv[1], e)
PY
  rm -rf "$OUT/p$i"
done
cat "$OUT/pmc_fp32_summary.txt"
echo done
