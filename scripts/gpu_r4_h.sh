#!/bin/bash
# Round 4, session H: FunkSVD's epoch in graph segments; the MF suite; RCCL through the binding (unique id fix).
mkdir -p gpurun_out/r4h
timeout 200 python scripts/funk_time.py > gpurun_out/r4h/funk.txt 2>&1; echo "funk rc=$?"; cat gpurun_out/r4h/funk.txt
timeout 300 python -m pytest tests/test_rccl_direct.py -q -m gpu -rx > gpurun_out/r4h/pytest_rccl.txt 2>&1; echo "rccl rc=$?"; tail -5 gpurun_out/r4h/pytest_rccl.txt
timeout 900 python -m pytest tests/test_mf_gpu.py -q -m gpu -x > gpurun_out/r4h/pytest_mf.txt 2>&1; echo "mf rc=$?"; tail -5 gpurun_out/r4h/pytest_mf.txt
