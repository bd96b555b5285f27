#!/bin/bash
# Round 4, session O: distance-aware naps of ticket waiters: speed and counter traffic of the two SLIM kernels.
mkdir -p gpurun_out/r4o
timeout 200 python scripts/slim_sweep.py nap 2>&1 | tail -8 | tee gpurun_out/r4o/nap.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4o
cd /tmp && export TMPDIR=/tmp MI355REC_NO_GRAPH=1
for P in slim_dense slim_symmetric; do
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$P/fetch -o run -- python $R/scripts/run_path.py $P > $OUT/$P.fetch.log 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$P/write -o run -- python $R/scripts/run_path.py $P > $OUT/$P.write.log 2>&1
done
cd $R
python scripts/summarize_pmc.py $OUT 2>&1 | grep -E "flow_kernel|====" | head -12
find $OUT -name "*.csv" -size +2M -delete
