#!/bin/bash
# Round 4, session P: the plain-sgd instances of the mini-batch kernels: one model, 32 models, parity.
mkdir -p gpurun_out/r4p
timeout 100 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | tee gpurun_out/r4p/ticks_bpr.txt
timeout 200 python scripts/group_time.py 32 | tee gpurun_out/r4p/group.txt
timeout 100 python scripts/mf_ticks.py funk 2>&1 | grep -v "since first" | head -3
timeout 900 python -m pytest tests/test_mf_gpu.py -q -m gpu -x > gpurun_out/r4p/pytest_mf.txt 2>&1; echo "mf rc=$?"; tail -4 gpurun_out/r4p/pytest_mf.txt
