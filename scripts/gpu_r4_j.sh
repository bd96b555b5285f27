#!/bin/bash
# Round 4, session J: FunkSVD with the batch index, the global-bias ring entry and the header requested together.
mkdir -p gpurun_out/r4j
for W in 0 1; do
  MI355REC_MF_WARM_NEXT=$W timeout 150 python scripts/mf_ticks.py funk > gpurun_out/r4j/ticks_funk_warm$W.txt 2>&1; echo "warm_next=$W"; grep -v "since first" gpurun_out/r4j/ticks_funk_warm$W.txt
done
timeout 150 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | head -4
timeout 900 python -m pytest tests/test_mf_gpu.py tests/test_sharding_gpu.py -q -m gpu -x > gpurun_out/r4j/pytest_mf.txt 2>&1; echo "mf rc=$?"; tail -5 gpurun_out/r4j/pytest_mf.txt
