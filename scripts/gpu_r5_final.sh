#!/bin/bash
# Round 5, final evidence: PMC round, rocprofv3 stats of the bench command + the default bench line, phase clocks, the full GPU suite.
mkdir -p gpurun_out/r5final
( timeout 1500 bash scripts/pmc_round.sh r5 > gpurun_out/r5final/pmc_round.log 2>&1 ); echo "pmc rc=$?"; tail -5 gpurun_out/r5final/pmc_round.log | cut -c1-200
bash scripts/prof_bench.sh r5 > gpurun_out/r5final/prof_bench.log 2>&1; tail -20 gpurun_out/r5final/prof_bench.log | cut -c1-150
timeout 300 python scripts/sim_phases.py > gpurun_out/r5final/sim_phases.txt 2>&1; tail -4 gpurun_out/r5final/sim_phases.txt | cut -c1-250
bash scripts/gpu_r5_d.sh quick > gpurun_out/r5final/ials_two_stage.txt 2>&1; grep "epoch" gpurun_out/r5final/ials_two_stage.txt
( timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -v "^  File\|DeprecationWarning\|^$" | tail -30 ) > gpurun_out/r5final/pytest_gpu_full_suite.txt; tail -5 gpurun_out/r5final/pytest_gpu_full_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
