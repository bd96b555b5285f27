#!/bin/bash
# Round 4, session F: the schedule of the next SLIM epoch behind the running kernel; the chunked part build; the RCCL one-rank test.
mkdir -p gpurun_out/r4f
timeout 600 python -m pytest tests/test_slim_gpu.py tests/test_sharding_gpu.py tests/test_rccl_direct.py -q -m gpu -rx > gpurun_out/r4f/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4f/pytest.txt
tail -40 gpurun_out/r4f/pytest.txt
timeout 300 python scripts/slim_sweep.py presched > gpurun_out/r4f/slim_presched.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/r4f/slim_presched.txt
