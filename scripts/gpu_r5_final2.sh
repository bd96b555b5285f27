#!/bin/bash
# Round 5, last refresh on the final tree: the default bench line, the similarity phase clocks (kernel and constructor), the
# similarity parity subset that covers every accumulator mode, smoke().
mkdir -p gpurun_out/r5final2
python bench.py > gpurun_out/r5final2/bench_n1.json 2> gpurun_out/r5final2/bench_n1.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r5final2/bench_n1.json
timeout 300 python scripts/sim_phases.py > gpurun_out/r5final2/sim_phases.txt 2>&1; grep "un-instrumented" gpurun_out/r5final2/sim_phases.txt | cut -c1-120
python scripts/sim_create_phases.py > gpurun_out/r5final2/sim_create_phases.txt 2>&1; grep "rep 2" gpurun_out/r5final2/sim_create_phases.txt
timeout 300 python -m pytest tests/test_sim_gpu.py tests/test_graph_gpu.py -q -m gpu -x -k "quantised or threshold_first or seeded or golden or heavy or full_size or graph or P3 or RP3 or feature_weighting or resident" 2>&1 | tail -2 > gpurun_out/r5final2/pytest_sim_subset.txt; cat gpurun_out/r5final2/pytest_sim_subset.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
