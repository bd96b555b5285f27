#!/bin/bash
# Round 4, session D: MF suite after the dataflow epoch was taken out, IALS full-shape rows against _update_row, similarity constructor
# phases, RCCL one-rank probe.
mkdir -p gpurun_out/r4d
timeout 300 python -m pytest tests/test_mf_gpu.py -q -m gpu -x > gpurun_out/r4d/pytest_mf.txt 2>&1; echo "pytest mf rc=$?"; tail -4 gpurun_out/r4d/pytest_mf.txt
timeout 300 python -m pytest tests/test_ials_gpu.py -q -m gpu -x -k "config_5" --durations=3 > gpurun_out/r4d/pytest_ials.txt 2>&1; echo "pytest ials rc=$?"; tail -8 gpurun_out/r4d/pytest_ials.txt
timeout 200 python scripts/sim_create_phases.py binary > gpurun_out/r4d/sim_create.txt 2>&1; echo "sim create rc=$?"; cat gpurun_out/r4d/sim_create.txt
timeout 400 python scripts/rccl_probe.py > gpurun_out/r4d/rccl_probe.txt 2>&1; echo "rccl probe rc=$?"; cat gpurun_out/r4d/rccl_probe.txt
