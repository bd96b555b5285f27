#!/bin/bash
# round 3, call D: float32 norm order, SLIM batched loads, AsySVD pipelined kernel, group epoch kernel trace
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 900 python -m pytest tests/test_sim_gpu.py tests/test_slim_gpu.py tests/test_sharding_gpu.py tests/test_graph_gpu.py tests/test_ease_gpu.py -q -m gpu -k "not netflix" -rfx --durations=8 2>&1 | grep -v "^  File\|^Extension\|DeprecationWarning\|^$" | tail -60 ) > gpurun_out/r3d/pytest_a.log
( timeout 600 python -m pytest tests/test_mf_gpu.py -q -m gpu -k "asysvd or group or fused or schedule or golden" -rfx --durations=5 2>&1 | grep -v "^  File\|^Extension\|DeprecationWarning\|^$" | tail -40 ) > gpurun_out/r3d/pytest_mf.log
( cd /tmp && MI355REC_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3d/trace_group -o run -- python $R/scripts/mf_group.py 32 --epochs 6 > $R/gpurun_out/r3d/mf_group_traced.log 2>&1 )
python - <<'P' > gpurun_out/r3d/mf_group_kernel_stats.txt 2>&1
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r3d/trace_group/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("rocprofv3 --kernel-trace --stats -- python scripts/mf_group.py 32 --epochs 6   (MI355REC_NO_GRAPH=1: plain launches)")
print("%-80s %8s %14s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "%"))
for r in rows[:16]:
    print("%-80s %8s %14s %12.1f %6.2f" % (r["Name"].replace("mi355rec::(anonymous namespace)::", "")[:80], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
P
find gpurun_out/r3d/trace_group -name "*.csv" -size +1M -delete
( timeout 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 3 > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err )
for f in pytest_a pytest_mf mf_group_traced; do echo "== $f"; cat gpurun_out/r3d/$f.log | cut -c1-300 | tail -45; done
echo "== kernel stats"; cat gpurun_out/r3d/mf_group_kernel_stats.txt | cut -c1-200
echo "== bench.err"; tail -8 gpurun_out/r3d/bench.err | cut -c1-300
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r3d/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
    for k, v in d["extra"].get("paths", {}).items():
        print(k, {a: b for a, b in v.items() if a in ("samples_per_s", "frac", "seconds_per_epoch", "speedup_vs_cpu_baseline", "avg_launch_us", "us_per_step", "users_per_s")})
    print(d["extra"].get("paths_error"), d["extra"].get("ials_error"))
except Exception as e:
    print("no bench line:", e)
P
