#!/bin/bash
# full GPU suite + smoke + default bench (what the driver runs at round end)
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -40 > gpurun_out/full/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
tail -15 gpurun_out/full/pytest_gpu.log; tail -3 gpurun_out/full/smoke.log; head -c 3000 gpurun_out/full/bench.json; tail -5 gpurun_out/full/bench.err
