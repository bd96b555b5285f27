#!/bin/bash
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_mf_gpu.py -q --maxfail=10 -x 2>&1 | grep -v "^  File\|^Extension" | tail -80 > gpurun_out/r2b/pytest_mf.log
export TMPDIR=/tmp
REPO=$PWD
cd /tmp && MI355REC_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/r2b/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 2 --no-extras --no-sim --no-cpu-baseline > $REPO/gpurun_out/r2b/prof_bench.json 2> $REPO/gpurun_out/r2b/prof.err
cd $REPO
f=$(find gpurun_out/r2b/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r2b/kernel_stats.csv
find gpurun_out/r2b/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'head -400 {} > gpurun_out/r2b/kernel_trace_head.csv'
rm -rf gpurun_out/r2b/prof
tail -15 gpurun_out/r2b/pytest_mf.log; head -20 gpurun_out/r2b/kernel_stats.csv
