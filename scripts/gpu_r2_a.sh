#!/bin/bash
# round 2, call A: MF redesign -- parity tests, bench line, kernel trace
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_mf_gpu.py -q --maxfail=10 2>&1 | tail -60 > gpurun_out/r2a/pytest_mf.log
timeout 400 python bench.py --steps 100 --warmup 10 --no-extras --no-sim --cpu-seconds 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
export TMPDIR=/tmp
REPO=$PWD
cd /tmp && MI355REC_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r2a/prof -- python $REPO/bench.py --steps 20 --warmup 2 --no-extras --no-sim --no-cpu-baseline > $REPO/gpurun_out/r2a/prof_bench.json 2> $REPO/gpurun_out/r2a/prof.err
cd $REPO
find gpurun_out/r2a/prof -name "*kernel_stats*" | head -1 | xargs -I{} head -30 {} > gpurun_out/r2a/kernel_stats_head.txt
find gpurun_out/r2a/prof -name "*.db" -delete; find gpurun_out/r2a/prof -name "*kernel_trace*" -delete
tail -5 gpurun_out/r2a/pytest_mf.log; cat gpurun_out/r2a/bench.json | head -c 1500
