#!/bin/bash
# First GPU session of the next round (about 14 minutes of box time): the whole -m gpu suite on the current tree, the default bench
# line, and the measurements this round ended without: overlapped epochs for a 32-model group, the IALS section with split rows.
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_next_a.sh'
mkdir -p gpurun_out/next_a
timeout 660 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/next_a/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/next_a/pytest_gpu.txt
tail -4 gpurun_out/next_a/pytest_gpu.txt
timeout 70 python scripts/mf_overlap.py 32 24 > gpurun_out/next_a/mf_overlap_32.txt 2>&1; echo "overlap rc=$?"
cat gpurun_out/next_a/mf_overlap_32.txt
timeout 60 python scripts/ials_only.py > gpurun_out/next_a/ials.json 2> gpurun_out/next_a/ials.err; echo "ials rc=$?"
timeout 200 python bench.py > gpurun_out/next_a/bench.json 2> gpurun_out/next_a/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/next_a/bench.json
