#!/bin/bash
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_mf_gpu.py -q --maxfail=6 2>&1 | grep -v "^  File\|^Extension" | tail -60 > gpurun_out/r2d/pytest_mf.log
timeout 300 python scratch/mf_ticks.py > gpurun_out/r2d/ticks.log 2>&1

tail -12 gpurun_out/r2d/pytest_mf.log; cat gpurun_out/r2d/ticks.log
