#!/bin/bash
# Round 4, session K: long streams on the in-LDS schedule, 256 mini-batches at a time.
mkdir -p gpurun_out/r4k
timeout 150 python scripts/mf_ticks.py funk > gpurun_out/r4k/ticks_funk.txt 2>&1; grep -v "since first" gpurun_out/r4k/ticks_funk.txt
MI355REC_MF_WHOLE_STREAM_SCHEDULE=1 timeout 150 python scripts/mf_ticks.py funk 2>&1 | grep -v "since first" | head -3
timeout 150 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | head -4
timeout 900 python -m pytest tests/test_mf_gpu.py tests/test_sharding_gpu.py -q -m gpu -x > gpurun_out/r4k/pytest_mf.txt 2>&1; echo "mf rc=$?"; tail -5 gpurun_out/r4k/pytest_mf.txt
