#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            -> per-kernel time (summary copied to profiles/)
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2)
# Usage: scripts/profile_round.sh <tag>        (writes gpurun_out/prof_<tag>/...)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MI355REC_NO_GRAPH=1   # rocprofv3 (ROCm 7.2) segfaults while tracing hipGraph replays; kernels are identical either way
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
cd $R
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
