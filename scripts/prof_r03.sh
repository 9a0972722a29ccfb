#!/bin/bash
# Dev tool (GPU box): rocprofv3 evidence for bench.py's roofline block and config.extra, round 3.
#   part "regimes": the dominant kernel's average duration in each cache regime of T1M (kernel-trace + stats) and of the
#                   bench command itself
#   part "traffic": FETCH_SIZE / WRITE_SIZE per launch (SEPARATE --pmc passes, kernel-trace only, as the pool requires)
#                   of T1M (replayed + rotating) and of every config.extra workload (Q1M, T2M, cfg5, cfg5auto, cfg5r, cfg5u)
# Usage: bash scripts/prof_r03.sh <outdir> [regimes|traffic|all] ; then python scripts/summarise_r03.py <outdir> profiles/r03
set -e
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/r3_rp}
PART=${2:-all}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
if [ "$PART" = regimes ] || [ "$PART" = all ]; then
  for r in replayed rewritten_inputs rotating_sets; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$r" -o p -- \
        python3 $B --no-cpu-baseline --only-regime $r > "$OUT/$r.log" 2>&1
    cp "$OUT/$r/p_kernel_stats.csv" "$OUT/kernel_stats_$r.csv"
    rm -rf "$OUT/$r"
    echo "regime $r done"
  done
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o p -- \
      python3 $B --no-cpu-baseline --no-regimes --no-extra > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.log"
  cp "$OUT/bench/p_kernel_stats.csv" "$OUT/kernel_stats_bench_py.csv"; rm -rf "$OUT/bench"
  echo "bench under rocprof done"
fi
pmc_one() {   # $1 counter, $2 tag, rest: bench args
  local c=$1 tag=$2; shift 2
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${c}_$tag" -o p -- \
      python3 $B --no-cpu-baseline --steps 50 "$@" > "$OUT/pmc_${c}_$tag.log" 2>&1
  python3 - "$OUT/pmc_${c}_$tag/p_counter_collection.csv" "$c" "$tag" >> "$OUT/pmc_summary.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if ("tri3_energy_" in r["Kernel_Name"] or "quad4_energy_" in r["Kernel_Name"]) and r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"].split("<")[0].split("(")[0]].append(float(r["Counter_Value"]))
k = max(acc, key=lambda n: len(acc[n]))
print(sys.argv[2], sys.argv[3], k, "launches", len(acc[k]), "mean", sum(acc[k]) / len(acc[k]))
PY
  rm -rf "$OUT/pmc_${c}_$tag"
  echo "pmc $c $tag done"
}
if [ "$PART" = traffic ] || [ "$PART" = all ]; then
  # XS: the workloads of this call (a gpurun call is limited to 20 minutes: split the list over calls; the summary file
  # is appended to)
  XS=${XS:-"T1M Q1M T2M cfg5 cfg5auto cfg5r cfg5u"}
  for c in FETCH_SIZE WRITE_SIZE; do
    for x in $XS; do
      if [ "$x" = T1M ]; then
        pmc_one $c T1M_replayed --only-regime replayed
        pmc_one $c T1M_rotating --only-regime rotating_sets
      else
        pmc_one $c $x --only-extra $x
      fi
    done
  done
  cat "$OUT/pmc_summary.txt"
fi
echo done
