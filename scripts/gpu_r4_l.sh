#!/bin/bash
# Round 4, session L: similarity constructor (column sort behind the value upload, chunk-parallel column costs).
mkdir -p gpurun_out/r4l
timeout 100 python scripts/sim_create_phases.py binary > gpurun_out/r4l/sim_create.txt 2>&1; echo "sim create rc=$?"; tail -9 gpurun_out/r4l/sim_create.txt
timeout 900 python -m pytest tests/test_sim_gpu.py tests/test_sharding_gpu.py -q -m gpu -x > gpurun_out/r4l/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4l/pytest.txt
