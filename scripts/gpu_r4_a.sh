#!/bin/bash
# Round 4, first GPU session: the rebuilt SLIM-BPR kernels (owned rows / granule cells) against the oracle, their timing, R models
# side by side; then the measurements round 3 ended without (overlapped epochs on a 32-model group, LDS scatter patterns).
# Usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r4_a.sh'
mkdir -p gpurun_out/r4a
timeout 420 python -m pytest tests/test_slim_gpu.py -q -m gpu -x --durations=8 > gpurun_out/r4a/pytest_slim.txt 2>&1; echo "pytest slim rc=$?" | tee -a gpurun_out/r4a/pytest_slim.txt
tail -25 gpurun_out/r4a/pytest_slim.txt
timeout 150 python scripts/slim_time.py > gpurun_out/r4a/slim_time.txt 2>&1; echo "slim_time rc=$?"; cat gpurun_out/r4a/slim_time.txt
MI355REC_SLIM_OWNERS=0 timeout 100 python scripts/slim_time.py ml20m > gpurun_out/r4a/slim_time_no_owners.txt 2>&1; echo "slim_time (no owners) rc=$?"; cat gpurun_out/r4a/slim_time_no_owners.txt
timeout 200 python scripts/slim_replicas.py dense 2 4 8 > gpurun_out/r4a/slim_replicas_dense.txt 2>&1; echo "replicas rc=$?"; cat gpurun_out/r4a/slim_replicas_dense.txt
timeout 200 python scripts/slim_replicas.py symmetric 2 4 > gpurun_out/r4a/slim_replicas_sym.txt 2>&1; echo "replicas rc=$?"; cat gpurun_out/r4a/slim_replicas_sym.txt
timeout 90 python scripts/mf_overlap.py 32 24 > gpurun_out/r4a/mf_overlap_32.txt 2>&1; echo "overlap rc=$?"; cat gpurun_out/r4a/mf_overlap_32.txt
timeout 60 bash -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_scatter_patterns.hip -o /tmp/lsp 2>/dev/null && /tmp/lsp" > gpurun_out/r4a/lds_scatter_patterns.txt 2>&1
cat gpurun_out/r4a/lds_scatter_patterns.txt
timeout 240 python -m pytest tests/test_mf_gpu.py -q -m gpu -x -k "not ml20m and not asysvd_full" > gpurun_out/r4a/pytest_mf.txt 2>&1; echo "pytest mf rc=$?"; tail -5 gpurun_out/r4a/pytest_mf.txt
