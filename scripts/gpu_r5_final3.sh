#!/bin/bash
# Round 5, the full GPU suite on the final tree (+ the stand-alone K-th selection benchmark after its last edit).
mkdir -p gpurun_out/r5final3
timeout 30 ./scratch/kth_select > gpurun_out/r5final3/kth_select.txt 2>&1; tail -3 gpurun_out/r5final3/kth_select.txt | cut -c1-200
( timeout 500 python -m pytest tests -q -m gpu --durations=6 2>&1 | grep -v "^  File\|DeprecationWarning\|^$" | tail -16 ) > gpurun_out/r5final3/pytest_gpu_full_suite.txt; tail -3 gpurun_out/r5final3/pytest_gpu_full_suite.txt
