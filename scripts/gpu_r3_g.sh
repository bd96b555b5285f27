#!/bin/bash
# round 3, call G: the batched-load SLIM flow kernel with the done-word hot spot removed
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
( timeout 200 python scripts/slim_time.py 2>&1 | tail -10 ) > gpurun_out/r3g/slim_time.log
cat gpurun_out/r3g/slim_time.log
( timeout 300 python -m pytest tests/test_slim_gpu.py -q -m gpu -rfx --timeout 100 2>&1 | grep -v "^  File\|^Extension\|DeprecationWarning\|^$" | tail -15 ) > gpurun_out/r3g/pytest_slim.log
cat gpurun_out/r3g/pytest_slim.log | cut -c1-250
