#!/bin/bash
# Round 5, session F: evidence -- rocprofv3 kernel stats of the bench command, the default bench line, the full GPU suite.
mkdir -p gpurun_out/r5f
bash scripts/prof_bench.sh r5 > gpurun_out/r5f/prof_bench.log 2>&1; tail -22 gpurun_out/r5f/prof_bench.log | cut -c1-150
( timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -v "^  File\|DeprecationWarning\|^$" | tail -30 ) > gpurun_out/r5f/pytest_gpu_full_suite.txt; tail -14 gpurun_out/r5f/pytest_gpu_full_suite.txt
