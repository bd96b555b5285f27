#!/bin/bash
# Round 4, session G: compute units left free for the schedule of the next symmetric epoch.
mkdir -p gpurun_out/r4g
timeout 300 python scripts/slim_sweep.py presched > gpurun_out/r4g/slim_presched.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/r4g/slim_presched.txt
timeout 600 python -m pytest tests/test_slim_gpu.py -q -m gpu > gpurun_out/r4g/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4g/pytest.txt
tail -5 gpurun_out/r4g/pytest.txt
