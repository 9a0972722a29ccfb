#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv (or, with --tail N, the average durations of the last N rows of a *_kernel_trace.csv)."""
import collections
import csv
import sys

path = sys.argv[1]
if "--tail" in sys.argv:
    n = int(sys.argv[sys.argv.index("--tail") + 1])
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Kernel_Name"].split("(")[0][-48:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        print(f"{k:50s} n {len(v):4d} avg {sum(v) / len(v):9.2f} us")
else:
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0][-50:]
        print(f"{name:52s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:9.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
