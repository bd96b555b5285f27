#!/bin/bash
# round 3, call E2: AsySVD kernel re-check, rocprofv3 evidence for every dominant kernel (stats + PMC passes), bench under rocprofv3 + default bench
mkdir -p gpurun_out/r3e2
export TMPDIR=/tmp
( timeout 240 python -m pytest tests/test_mf_gpu.py -q -m gpu -k "asysvd" -rfx 2>&1 | tail -6 ) > gpurun_out/r3e2/pytest_asy.log
cat gpurun_out/r3e2/pytest_asy.log | cut -c1-200
timeout 900 bash scripts/pmc_round.sh r3 > gpurun_out/r3e2/pmc_round.log 2>&1
tail -100 gpurun_out/r3e2/pmc_round.log | cut -c1-260
timeout 500 bash scripts/prof_bench.sh r3 > gpurun_out/r3e2/prof_bench.log 2>&1
tail -30 gpurun_out/r3e2/prof_bench.log | cut -c1-300
