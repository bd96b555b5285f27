#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench workload (plain launches: MI355REC_NO_GRAPH=1, see profile_round.sh)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT/trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MI355REC_NO_GRAPH=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
echo rc=$?
cd $R; python scripts/summarize_prof.py $OUT | sed -n '1,14p'
