#!/bin/bash
# rocprofv3 evidence for EVERY dominant kernel (run through gpurun from the repo root):
#   --kernel-trace --stats                      per-kernel time
#   --pmc FETCH_SIZE / --pmc WRITE_SIZE         HBM-side traffic, SEPARATE passes (TCC slots), per MI355X_MICROARCH.md
#   --pmc SQ_* for the similarity and IALS kernels (LDS activity / bank conflicts, VALU issue)
# Usage: scripts/pmc_round.sh <tag>     (writes gpurun_out/pmc_<tag>/..., summary in gpurun_out/pmc_<tag>/summary.txt + pmc_traffic.json)
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MI355REC_NO_GRAPH=1
python $R/scripts/run_path.py sim > /dev/null 2>&1      # (generates and caches the synthetic URM)
for P in mf mf_group funk sim score slim_dense slim_symmetric; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$P/trace -o run -- python $R/scripts/run_path.py $P > $OUT/$P.trace.log 2>&1
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$P/fetch -o run -- python $R/scripts/run_path.py $P > $OUT/$P.fetch.log 2>&1
  timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$P/write -o run -- python $R/scripts/run_path.py $P > $OUT/$P.write.log 2>&1
done
for P in ials asy; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$P/trace -o run -- python $R/scripts/run_path.py $P > $OUT/$P.trace.log 2>&1
done
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sim/sq -o run -- python $R/scripts/run_path.py sim > $OUT/sim.sq.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/ials/sq -o run -- python $R/scripts/run_path.py ials > $OUT/ials.sq.log 2>&1
cd $R
python scripts/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt | head -120
