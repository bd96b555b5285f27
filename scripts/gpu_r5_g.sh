#!/bin/bash
# round 5, call g: CSR -> CSC as a sort of the cells (no gather); merged value scan.  Parity of every mode + where the constructor's time goes.
mkdir -p gpurun_out/r5g
timeout 600 python -m pytest tests/test_sim_gpu.py tests/test_graph_gpu.py tests/test_ease_gpu.py tests/test_scoring_gpu.py tests/test_sharding_gpu.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r5g/pytest.txt
cat gpurun_out/r5g/pytest.txt
python scripts/sim_create_phases.py > gpurun_out/r5g/create_phases.txt 2>&1
grep -v "^----\|rep 0" gpurun_out/r5g/create_phases.txt | tail -24
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_create -o create -- python /root/repo/scripts/sim_create_phases.py > /dev/null 2>&1
f=$(ls /tmp/prof_create/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" /root/repo/gpurun_out/r5g/create_kernel_stats.csv && head -25 "$f" | cut -c1-160
