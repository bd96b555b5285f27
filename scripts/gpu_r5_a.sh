#!/bin/bash
# Round 5, session A: threshold-first top-K of the similarity kernel -- parity (fast == full selection, oracle), phases, timing.
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_sim_gpu.py -q -m gpu -x -k "threshold_first or golden or topk_sizes or edge_cases or seeded_ml1m or heavy_columns or quantised" > gpurun_out/r5a/pytest_sim_new.txt 2>&1; echo "sim-new rc=$?"; tail -5 gpurun_out/r5a/pytest_sim_new.txt
timeout 300 python scripts/sim_phases.py > gpurun_out/r5a/sim_phases.txt 2>&1; cat gpurun_out/r5a/sim_phases.txt
