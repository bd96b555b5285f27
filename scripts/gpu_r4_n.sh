#!/bin/bash
# Round 4, session N: four samples per wavefront (16 lanes x 2 chunks per row) at k = 128: group, single model, parity.
mkdir -p gpurun_out/r4n
timeout 200 python scripts/group_time.py 32
MI355REC_MF_LANES16=1 timeout 200 python scripts/group_time.py 32
timeout 100 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | head -3
MI355REC_MF_LANES16=1 timeout 100 python scripts/mf_ticks.py bpr 2>&1 | grep -v "since first" | head -8
MI355REC_MF_LANES16=1 timeout 100 python scripts/mf_ticks.py funk 2>&1 | grep -v "since first" | head -3
MI355REC_MF_LANES16=1 timeout 600 python -m pytest tests/test_mf_gpu.py -q -m gpu -x -k "headline or group or k128 or funksvd_ml20m or config" > gpurun_out/r4n/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4n/pytest.txt
