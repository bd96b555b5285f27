#!/bin/bash
# Round 4, final session: the whole -m gpu suite, smoke(), the per-path rocprofv3 / PMC collection, bench under rocprofv3 + the default bench line.
mkdir -p gpurun_out/r4final
timeout 900 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r4final/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4final/pytest_gpu.txt
tail -14 gpurun_out/r4final/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4final/smoke.txt 2>&1; tail -2 gpurun_out/r4final/smoke.txt
bash scripts/pmc_round.sh r4 > gpurun_out/pmc_r4.log 2>&1; grep -c "pmc " gpurun_out/pmc_r4/summary.txt
bash scripts/prof_bench.sh r4 > gpurun_out/prof_bench_r4.log 2>&1; tail -3 gpurun_out/prof_bench_r4.log
