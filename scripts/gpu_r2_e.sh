#!/bin/bash
mkdir -p gpurun_out/r2e
timeout 1200 python -m pytest tests/test_slim_gpu.py -q --maxfail=6 2>&1 | grep -v "^  File\|^Extension" | tail -60 > gpurun_out/r2e/pytest_slim.log
timeout 300 python scratch/slim_time.py > gpurun_out/r2e/slim_time.log 2>&1
tail -30 gpurun_out/r2e/pytest_slim.log; cat gpurun_out/r2e/slim_time.log
