#!/bin/bash
# Round 4, last session: the whole -m gpu suite and the default bench line on the final tree.
mkdir -p gpurun_out/r4final2
timeout 900 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r4final2/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4final2/pytest_gpu.txt
tail -5 gpurun_out/r4final2/pytest_gpu.txt
timeout 300 python bench.py > gpurun_out/r4final2/bench_default.json 2> gpurun_out/r4final2/bench_default.err; echo "bench rc=$?"; tail -2 gpurun_out/r4final2/bench_default.err
