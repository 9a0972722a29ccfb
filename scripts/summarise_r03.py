#!/usr/bin/env python3
"""Turn the output directory of scripts/prof_r03.sh into the summaries bench.py reads:
profiles/r03/regimes_rocprof.json (per-regime rocprofv3 average of the dominant kernel on T1M) and profiles/r03/traffic.json
(FETCH_SIZE / WRITE_SIZE per launch of T1M and of every config.extra workload, corrected as MI355X_MICROARCH.md prescribes),
each keyed by the workload's shape "elements/nodes/tiles" (taken from `--bench-json`, a bench.py output of the same build).

    python scripts/summarise_r03.py gpurun_out/r3_rp profiles/r03 --bench-json gpurun_out/r3_bench.json
"""
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
bench = json.load(open(sys.argv[sys.argv.index("--bench-json") + 1])) if "--bench-json" in sys.argv else None
os.makedirs(dst, exist_ok=True)
shapes = {}
if bench is not None:
    c = bench["config"]
    shapes["T1M"] = (f"{c['elements']}/{c['nodes']}/{c['tiles']}", bench["roofline"]["alg_bytes_per_launch"])
    for e in c.get("extra", []):
        shapes[e["key"]] = (f"{e['elements']}/{e['nodes']}/{e['tiles']}", e["alg_bytes_per_launch"])
ALG = 12 * 1000000 + 64 * 501501 + 8
if os.path.exists(os.path.join(src, "kernel_stats_replayed.csv")):
    out = {"source": "rocprofv3 --kernel-trace --stats on `bench.py --no-cpu-baseline --only-regime <r>` (scripts/prof_r03.sh / prof_r04.sh), MI355X; "
                     "hipGraph launches traced",
           "shape": shapes.get("T1M", ("1000000/501501/1024",))[0], "alg_bytes_per_launch": ALG, "regimes": {}}
    for r in ("replayed", "rewritten_inputs", "rotating_sets"):
        rows = [x for x in csv.DictReader(open(os.path.join(src, f"kernel_stats_{r}.csv"))) if "tri3_energy_" in x["Name"]]
        # T1M's plan takes the sc1 (16) store instance; other instances of the kernel in the same run are other legs
        sc1 = [x for x in rows if x["Name"].split("(")[0].rstrip().endswith(", 16>")]
        k = max(sc1 or rows, key=lambda x: float(x["TotalDurationNs"]))
        avg = float(k["AverageNs"]) * 1e-3
        out["regimes"][r] = {"kernel": k["Name"][:90], "calls": int(k["Calls"]), "avg_us": avg, "min_us": float(k["MinNs"]) * 1e-3,
                             "frac_of_8TBs": ALG / (avg * 1e-6) / 1e9 / 8000.0}
        shutil.copy(os.path.join(src, f"kernel_stats_{r}.csv"), os.path.join(dst, f"kernel_stats_{r}.csv"))
    for f in ("kernel_stats_bench_py.csv", "bench_under_rocprof.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    json.dump(out, open(os.path.join(dst, "regimes_rocprof.json"), "w"), indent=1)
    print(json.dumps({r: round(v["avg_us"], 3) for r, v in out["regimes"].items()}))
if os.path.exists(os.path.join(src, "pmc_summary.txt")):
    shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, "pmc_summary.txt"))
    pm = {}
    for line in open(os.path.join(src, "pmc_summary.txt")):
        p = line.split()
        k = p.index("launches")                       # the kernel name in between may contain spaces ("void hfem::...")
        pm[(p[0], p[1])] = (float(p[-1]), " ".join(p[2:k]), int(p[k + 1]))
    tr = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes on `bench.py --only-regime ...` / "
                    "`--only-extra ...` (scripts/prof_r03.sh), MI355X round 3, mean over the launches of the energy kernel.  The "
                    "counters sit on the L2's fabric side: requests served by the Infinity Cache are counted (upper bound on HBM bytes)",
          "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; "
                        "WRITE_SIZE exact; both in KB",
          "workloads": {}}
    tags = sorted({t for (_, t) in pm})
    for t in tags:
        if ("FETCH_SIZE", t) not in pm or ("WRITE_SIZE", t) not in pm:
            continue
        f, w = pm[("FETCH_SIZE", t)], pm[("WRITE_SIZE", t)]
        key = "T1M" if t == "T1M_replayed" else t
        shape, alg = shapes.get("T1M" if t.startswith("T1M") else t, (None, None))
        tr["workloads"][key] = {"kernel": f[1], "launches": f[2], "FETCH_SIZE_KB": f[0], "WRITE_SIZE_KB": w[0],
                                "read_bytes_corrected": 2 * f[0] * 1024, "write_bytes": w[0] * 1024,
                                "traffic_bytes_per_launch": 2 * f[0] * 1024 + w[0] * 1024, "shape": shape,
                                "alg_bytes_per_launch": alg,
                                "traffic_over_alg": (2 * f[0] * 1024 + w[0] * 1024) / alg if alg else None}
    json.dump(tr, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    for k, v in tr["workloads"].items():
        print(k, v["shape"], round(v["traffic_bytes_per_launch"] / 1e6, 2), "MB", v["traffic_over_alg"])
