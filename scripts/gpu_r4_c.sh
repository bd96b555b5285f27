#!/bin/bash
# Round 4, third GPU session: the BPR-MF dataflow epoch (tests first, under a timeout: a wrong hand-off aborts after 5 s), its timing
# at the headline shape; SLIM turn phases / sleeping waiters.
mkdir -p gpurun_out/r4c
timeout 300 python -m pytest tests/test_mf_gpu.py -q -m gpu -x -k "dataflow or bpr_replay_parity or bpr_batch_sizes or golden or baseline_config_2 or native or fused or schedule or group" > gpurun_out/r4c/pytest_mf_a.txt 2>&1; echo "pytest mf (flow subset) rc=$?"; tail -15 gpurun_out/r4c/pytest_mf_a.txt
timeout 200 python scripts/mf_flow_time.py 100 > gpurun_out/r4c/mf_flow_time.txt 2>&1; echo "flow time rc=$?"; cat gpurun_out/r4c/mf_flow_time.txt
timeout 300 python -m pytest tests/test_mf_gpu.py -q -m gpu -x > gpurun_out/r4c/pytest_mf_full.txt 2>&1; echo "pytest mf full rc=$?"; tail -5 gpurun_out/r4c/pytest_mf_full.txt
timeout 120 python -m pytest tests/test_slim_gpu.py -q -m gpu -x > gpurun_out/r4c/pytest_slim.txt 2>&1; echo "pytest slim rc=$?"; tail -3 gpurun_out/r4c/pytest_slim.txt
timeout 200 python scripts/slim_sweep.py turn > gpurun_out/r4c/slim_turn.txt 2>&1; echo "slim turn rc=$?"; cat gpurun_out/r4c/slim_turn.txt
