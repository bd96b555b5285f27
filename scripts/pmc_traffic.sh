#!/bin/bash
# HBM traffic counters for the BPR gradient kernel: two SEPARATE tiny passes (FETCH_SIZE, WRITE_SIZE), 2 epochs each
# (counter collection serialises dispatches and is slow with thousands of launches).  Run through gpurun from the repo root.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${1:-r1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MI355REC_NO_GRAPH=1
ARGS="--steps 2 --warmup 0 --no-cpu-baseline --no-extras --no-sim"
timeout 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
echo "fetch pass rc=$?"
timeout 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
echo "write pass rc=$?"
cd $R && python scripts/summarize_prof.py $OUT | sed -n '/pmc FETCH/,$p'
