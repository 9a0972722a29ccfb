#!/bin/bash
# Dev tool (GPU box): rocprofv3 evidence for bench.py's roofline block -- the dominant kernel's average duration in each
# cache regime (kernel-trace + stats; hipGraph launches are traced) and the HBM-side traffic of the replayed regime
# (PMC passes: FETCH_SIZE and WRITE_SIZE cannot share a pass; kernel-trace only, as the pool requires).
# Usage: bash scripts/prof_regimes.sh <outdir>    (summaries land in <outdir>/*.csv; copy the ones to keep to profiles/)
set -e
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/r2_rp}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for r in replayed rewritten_inputs rotating_sets; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$r" -o p -- \
      python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --only-regime $r > "$OUT/$r.log" 2>&1
  cp "$OUT/$r/p_kernel_stats.csv" "$OUT/kernel_stats_$r.csv"
  rm -rf "$OUT/$r"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o p -- \
    python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-regimes --no-extra > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.log"
cp "$OUT/bench/p_kernel_stats.csv" "$OUT/kernel_stats_bench_py.csv"; rm -rf "$OUT/bench"
for c in FETCH_SIZE WRITE_SIZE; do
  for r in replayed rotating_sets; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${c}_$r" -o p -- \
        python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --only-regime $r --steps 50 > "$OUT/pmc_${c}_$r.log" 2>&1
    python3 - "$OUT/pmc_${c}_$r/p_counter_collection.csv" "$c" "$r" >> "$OUT/pmc_summary.txt" <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1]))
        if "tri3_energy_" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print(sys.argv[2], sys.argv[3], "launches", len(vals), "mean", sum(vals) / max(len(vals), 1))
PY
    rm -rf "$OUT/pmc_${c}_$r"
  done
done
cat "$OUT/pmc_summary.txt"
echo done
