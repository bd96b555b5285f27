#!/bin/bash
# Round 4, session R: a pair task's other records by slot (requested with the header).
mkdir -p gpurun_out/r4r
timeout 100 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | head -9 | tee gpurun_out/r4r/ticks_bpr.txt
timeout 200 python scripts/group_time.py 32 | tee gpurun_out/r4r/group.txt
timeout 900 python -m pytest tests/test_mf_gpu.py tests/test_sharding_gpu.py -q -m gpu -x > gpurun_out/r4r/pytest_mf.txt 2>&1; echo "mf rc=$?"; tail -4 gpurun_out/r4r/pytest_mf.txt
