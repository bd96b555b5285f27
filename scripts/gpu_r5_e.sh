#!/bin/bash
# Round 5, session E: advisor items (SLIM many-epoch drift, chunked wide top-K parts) + sharding suite.
mkdir -p gpurun_out/r5e
timeout 900 python -m pytest tests/test_slim_gpu.py -q -m gpu -x -k "many_epochs" > gpurun_out/r5e/pytest_slim_drift.txt 2>&1; echo "slim drift rc=$?"; tail -15 gpurun_out/r5e/pytest_slim_drift.txt | cut -c1-300
timeout 900 python -m pytest tests/test_sharding_gpu.py -q -m gpu -x > gpurun_out/r5e/pytest_sharding.txt 2>&1; echo "sharding rc=$?"; tail -4 gpurun_out/r5e/pytest_sharding.txt
