#!/usr/bin/env python3
"""Condenses a scripts/pmc_round.sh output directory into summary.txt (stdout) and pmc_traffic.json."""
import csv
import datetime
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
OURS = ("mf_", "sim_", "slim_", "ials_", "gram_", "score_", "spscore", "wide_", "segment_")


def short(name):
    name = name.replace("mi355rec::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def find(sub, pattern):
    hits = glob.glob(os.path.join(root, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


traffic = {}
for path in ("mf", "mf_group", "funk", "sim", "slim_dense", "slim_symmetric", "ials", "score", "asy"):
    print("=" * 30, path, "=" * 30)
    stats = find(path + "/trace", "*kernel_stats.csv")
    durations = {}
    call_ms, n_launches = None, 0
    log = os.path.join(root, path + ".trace.log")
    if os.path.isfile(log):
        for line in open(log, errors="replace"):
            if line.startswith("path_call_ms="):
                fields = dict(f.split("=") for f in line.split())
                call_ms, n_launches = float(fields["path_call_ms"]), int(fields.get("n_launches", "0"))
    if stats:
        print("%-72s %7s %14s %12s %6s" % ("kernel (rocprofv3 --kernel-trace --stats)", "calls", "total_ns", "avg_ns", "%"))
        rows = list(csv.DictReader(open(stats)))
        for r in rows[:16 if path == "sim" else 8]:
            k = short(r["Name"])
            durations[k] = float(r["AverageNs"])
            print("%-72s %7s %14s %12.1f %6.2f" % (k, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
        if call_ms is not None and rows:
            # consistency: the handle's stream events around its LAST call bound the kernels of that call:
            # (launches of the dominant kernel in the call) x (rocprofv3's average duration of that kernel) <= the call
            top = rows[0]
            need_ms = float(top["AverageNs"]) * 1e-6 * max(1, min(n_launches, int(top["Calls"])))
            print("  handle's stream events: last call %.3f ms, %d launches of the dominant kernel; %s: %d x %.1f ns = %.3f ms" % (
                call_ms, n_launches, short(top["Name"])[:40], max(1, min(n_launches, int(top["Calls"]))), float(top["AverageNs"]), need_ms))
            if short(top["Name"]).startswith(OURS) and need_ms > call_ms * 1.10 + 0.05:
                raise SystemExit("INCONSISTENT with the handle's own timing: the workload profiled is not the workload timed")
    per = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for sub in ("fetch", "write", "sq"):
        f = find(path + "/" + sub, "*counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith(OURS):
                continue
            c = per[k][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    for k, counters in sorted(per.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        line = "  pmc %-60s" % k
        for name, (n, total) in sorted(counters.items()):
            line += "  %s %.4g/launch (%d launches)" % (name, total / max(n, 1), n)
        print(line)
        f_kib = counters.get("FETCH_SIZE", [0, 0.0]); w_kib = counters.get("WRITE_SIZE", [0, 0.0])
        if f_kib[0] or w_kib[0]:
            fetch = 2.0 * 1024.0 * f_kib[1] / max(f_kib[0], 1)       # KiB -> B; x2: gfx950 counts 128-B requests as 64 B (MI355X_MICROARCH.md)
            write = 1024.0 * w_kib[1] / max(w_kib[0], 1)
            traffic[k] = {"fetch_bytes_per_launch_x2_corrected": fetch, "write_bytes_per_launch": write,
                          "hbm_bytes_per_launch": fetch + write, "avg_ns": durations.get(k)}
try:
    import subprocess
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    head = ""
head = head or os.environ.get("MI355REC_GIT_HEAD", "unknown (the GPU box receives a snapshot without .git; see the commit that adds this file)")
doc = {"collected": datetime.date.today().isoformat(), "git_head": head, "workload": "ML-20M-shaped synthetic URM, scripts/run_path.py <path>",
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B); WRITE_SIZE uncalibrated; memory-side "
               "counters include Infinity-Cache hits", "kernels": traffic}
json.dump(doc, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print("\nwrote", os.path.join(root, "pmc_traffic.json"))
