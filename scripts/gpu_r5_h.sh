#!/bin/bash
# round 5, call h: constructor -- coalesced stream fill, wave-reduced value scan, narrower cost sort.  Parity + phases.
mkdir -p gpurun_out/r5h
timeout 300 python -m pytest tests/test_sim_gpu.py tests/test_graph_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r5h/pytest.txt
cat gpurun_out/r5h/pytest.txt
python scripts/sim_create_phases.py > gpurun_out/r5h/create_phases.txt 2>&1
grep -A7 "resident URM rep 1" gpurun_out/r5h/create_phases.txt; grep "rep 2" gpurun_out/r5h/create_phases.txt
