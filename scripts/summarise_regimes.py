#!/usr/bin/env python3
"""Turn the output directory of scripts/prof_regimes.sh into the two summaries bench.py reads:
profiles/r02/regimes_rocprof.json (per-regime rocprofv3 average of the dominant kernel) and
profiles/r02_hbm_traffic_T1M.json (FETCH_SIZE / WRITE_SIZE per launch, corrected as MI355X_MICROARCH.md prescribes).

    python scripts/summarise_regimes.py gpurun_out/r2_rp2 profiles/r02
"""
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
ALG = 12 * 1000000 + 64 * 501501 + 8
out = {"source": "rocprofv3 --kernel-trace --stats on `bench.py --no-cpu-baseline --only-regime <r>` (scripts/prof_regimes.sh), "
                 "MI355X, round 2 (final kernel: one-round prologue, paired slots); hipGraph launches traced",
       "workload": {"elements": 1000000, "nodes": 501501, "tiles": 1024}, "alg_bytes_per_launch": ALG, "regimes": {}}
for r in ("replayed", "rewritten_inputs", "rotating_sets"):
    rows = [x for x in csv.DictReader(open(os.path.join(src, f"kernel_stats_{r}.csv"))) if "tri3_energy_" in x["Name"]]
    k = max(rows, key=lambda x: float(x["TotalDurationNs"]))
    avg = float(k["AverageNs"]) * 1e-3
    out["regimes"][r] = {"kernel": k["Name"][:90], "calls": int(k["Calls"]), "avg_us": avg, "min_us": float(k["MinNs"]) * 1e-3,
                         "frac_of_8TBs": ALG / avg * 1e-6 / 8000.0}
    shutil.copy(os.path.join(src, f"kernel_stats_{r}.csv"), os.path.join(dst, f"kernel_stats_{r}.csv"))
for f in ("kernel_stats_bench_py.csv", "bench_under_rocprof.json", "pmc_summary.txt"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
json.dump(out, open(os.path.join(dst, "regimes_rocprof.json"), "w"), indent=1)
pm = {}
for line in open(os.path.join(src, "pmc_summary.txt")):
    p = line.split()
    pm[(p[0], p[1])] = float(p[-1])
fetch, write = pm[("FETCH_SIZE", "replayed")], pm[("WRITE_SIZE", "replayed")]
tr = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes on `bench.py --only-regime replayed|rotating_sets` "
                "(scripts/prof_regimes.sh), MI355X round 2, mean over the launches of tri3_energy_pair_kernel; identical in both regimes "
                "(the counters sit on the L2's fabric side: Infinity-Cache hits are counted)",
      "workload": out["workload"], "counters": {"FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
                                                 "FETCH_SIZE_KB_rotating": pm[("FETCH_SIZE", "rotating_sets")],
                                                 "WRITE_SIZE_KB_rotating": pm[("WRITE_SIZE", "rotating_sets")]},
      "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE exact",
      "read_bytes_corrected": 2 * fetch * 1024, "write_bytes": write * 1024,
      "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024}
json.dump(tr, open(os.path.join(os.path.dirname(dst.rstrip("/")), "r02_hbm_traffic_T1M.json"), "w"), indent=1)
print(json.dumps({r: round(v["avg_us"], 3) for r, v in out["regimes"].items()}), "traffic", tr["traffic_bytes_per_launch"])
