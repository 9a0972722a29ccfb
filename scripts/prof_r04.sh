#!/bin/bash
# Dev tool (GPU box): rocprofv3 evidence for round 4 (kernel-trace + stats only; the counter traffic of the headline kernel is
# measured by bench.py itself since this round -- two --pmc child passes, roofline.traffic_kind).
#   regimes   the dominant kernel's average duration in each cache regime of T1M + the bench command itself
#   fp32      the fp32-arithmetic kernel against the fp64-arithmetic float-row instance (scripts/fp32_kernel_timing.py)
#   sharded   the owner-sharded steps on one rank (scripts/sharded_step_timing.py): one-launch step vs energy + put
# Usage: bash scripts/prof_r04.sh <outdir> ; then python scripts/summarise_r03.py <outdir> profiles/r04 --bench-json <bench.json>
set -e
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/r4_rp}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for r in replayed rewritten_inputs rotating_sets; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$r" -o p -- \
      python3 $B --no-cpu-baseline --only-regime $r > "$OUT/$r.log" 2>&1
  cp "$OUT/$r/p_kernel_stats.csv" "$OUT/kernel_stats_$r.csv"; rm -rf "$OUT/$r"
  echo "regime $r done"
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o p -- \
    python3 $B --no-cpu-baseline --no-regimes --no-extra --no-pmc > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench.log"
cp "$OUT/bench/p_kernel_stats.csv" "$OUT/kernel_stats_bench_py.csv"; rm -rf "$OUT/bench"
echo "bench under rocprof done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fp32" -o p -- \
    python3 $GRAFT_REPO_ROOT/scripts/fp32_kernel_timing.py > "$OUT/fp32_timing_under_rocprof.json" 2> "$OUT/fp32.log"
cp "$OUT/fp32/p_kernel_stats.csv" "$OUT/kernel_stats_fp32_model.csv"; rm -rf "$OUT/fp32"
echo "fp32 done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sh" -o p -- \
    python3 $GRAFT_REPO_ROOT/scripts/sharded_step_timing.py > "$OUT/sharded_step_under_rocprof.json" 2> "$OUT/sharded.log"
cp "$OUT/sh/p_kernel_stats.csv" "$OUT/kernel_stats_sharded_steps.csv"; rm -rf "$OUT/sh"
echo "sharded done"
