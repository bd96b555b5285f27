#!/bin/bash
# Round 4, session Q: a third of the header slots per launch + loop over the slots in use (single model).
mkdir -p gpurun_out/r4q
timeout 100 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | head -4
MI355REC_MF_FULL_GRID=1 timeout 100 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | head -2
timeout 100 python scripts/mf_ticks.py funk 2>&1 | grep -v "since first" | head -2
MI355REC_MF_FULL_GRID=1 timeout 100 python scripts/mf_ticks.py funk 2>&1 | grep -v "since first" | head -2
timeout 900 python -m pytest tests/test_mf_gpu.py tests/test_sharding_gpu.py -q -m gpu -x > gpurun_out/r4q/pytest_mf.txt 2>&1; echo "mf rc=$?"; tail -4 gpurun_out/r4q/pytest_mf.txt
