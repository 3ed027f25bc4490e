#!/usr/bin/env python3
"""Sum a counter over the LAST n dispatches of trs_potrf_narrow_kernel in a rocprofv3 counter_collection.csv tree:
    python tools/potrf_cube_traffic_sum.py <dir> <n>"""
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(path)) if "trs_potrf_narrow_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2])
by = {}
for r in rows:
    by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for name, vals in by.items():
    vals.sort()
    print(name, sum(v for _, v in vals[-n:]) * 1024 / 1e9, "GB (KiB counters)")
