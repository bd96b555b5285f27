#!/usr/bin/env python3
"""Condenses a scripts/profile_round.sh output directory into the text summary kept under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pattern):
    hits = glob.glob(os.path.join(root, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("mi355rec::(anonymous namespace)::", "").replace("void ", "")
    return name[:78]


stats = find("trace", "*kernel_stats.csv")
print("== rocprofv3 --kernel-trace --stats  (python bench.py --steps 50 --warmup 5 --no-cpu-baseline) ==")
if stats:
    rows = list(csv.DictReader(open(stats)))
    print("%-78s %8s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "%"))
    for r in rows[:14]:
        print("%-78s %8s %14s %12.1f %7.2f" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
else:
    print("kernel_stats.csv not found under", root)

for label, sub, counter in (("FETCH_SIZE", "pmc_fetch", "FETCH_SIZE"), ("WRITE_SIZE", "pmc_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    print("\n== rocprofv3 --pmc %s (separate pass) ==" % label)
    if not f:
        print("counter_collection.csv not found")
        continue
    per = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    print("%-78s %8s %16s %16s" % ("kernel", "launches", "sum (KiB)", "per launch (KiB)"))
    for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]:
        print("%-78s %8d %16.1f %16.2f" % (k, n, v, v / n))
print("\nNote (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> double it;"
      "\nWRITE_SIZE and narrow accesses are uncalibrated; factors (85 MB) sit in the 256 MB Infinity Cache, so the memory-side"
      "\ncounters under-report what the kernels move.")
