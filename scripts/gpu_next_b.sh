#!/bin/bash
# Second GPU session of the next round (about 8 minutes): where the batched-load SLIM kernel stalls, and the two microbenchmarks
# behind the similarity kernel's next step.   Usage: gpurun --timeout 600 -- 'bash scripts/gpu_next_b.sh'
mkdir -p gpurun_out/next_b
timeout 420 python scripts/slim_batched_debug.py ml1m 0.1 0.3 0.6 1.0 > gpurun_out/next_b/slim_batched_ml1m.txt 2>&1
cat gpurun_out/next_b/slim_batched_ml1m.txt
timeout 60 bash -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_scatter_patterns.hip -o /tmp/lsp 2>/dev/null && /tmp/lsp" > gpurun_out/next_b/lds_scatter_patterns.txt 2>&1
cat gpurun_out/next_b/lds_scatter_patterns.txt
