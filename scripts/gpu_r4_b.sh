#!/bin/bash
# Round 4, second GPU session: SLIM-BPR with step descriptors + wavefront-level queue; knob sweeps and phase clocks.
mkdir -p gpurun_out/r4b
timeout 300 python -m pytest tests/test_slim_gpu.py -q -m gpu -x > gpurun_out/r4b/pytest_slim.txt 2>&1; echo "pytest slim rc=$?" | tee -a gpurun_out/r4b/pytest_slim.txt
tail -5 gpurun_out/r4b/pytest_slim.txt
timeout 400 python scripts/slim_sweep.py both > gpurun_out/r4b/slim_sweep.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/r4b/slim_sweep.txt
