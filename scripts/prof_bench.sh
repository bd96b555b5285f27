#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command itself (plain launches: rocprofv3 on ROCm 7.2 crashes while tracing hipGraph
# replays), then the default bench line.  Usage (through gpurun, from the repo root): bash scripts/prof_bench.sh <tag>
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MI355REC_NO_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- \
    python $R/bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(out + "/summary.txt", "w") as w:
    w.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline   (MI355REC_NO_GRAPH=1: plain launches)\n")
    w.write("%-72s %8s %14s %12s %6s\n" % ("kernel", "calls", "total_ns", "avg_ns", "%"))
    for r in rows[:14]:
        w.write("%-72s %8s %14s %12.1f %6.2f\n" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
print(open(out + "/summary.txt").read())
PY
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -2 $OUT/bench_default.err
find $OUT -name "*.csv" -size +2M -delete
