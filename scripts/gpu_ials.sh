#!/bin/bash
mkdir -p gpurun_out/ials
timeout 900 python -m pytest tests/test_ials_gpu.py -q -x 2>&1 | grep -v "^  File\|^Extension" | tail -30 > gpurun_out/ials/pytest.log
MI355REC_IALS_PHASES=1 timeout 300 python scratch/ials_time.py 200 > gpurun_out/ials/time.log 2>&1
tail -12 gpurun_out/ials/pytest.log; tail -6 gpurun_out/ials/time.log
