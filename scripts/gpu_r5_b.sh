#!/bin/bash
# Round 5, session B: full similarity suite + the bench line with the threshold-first selection.
mkdir -p gpurun_out/r5b
timeout 1200 python -m pytest tests/test_sim_gpu.py tests/test_graph_gpu.py tests/test_ease_gpu.py tests/test_sharding_gpu.py -q -m gpu -x > gpurun_out/r5b/pytest_sim.txt 2>&1; echo "sim rc=$?"; tail -4 gpurun_out/r5b/pytest_sim.txt
timeout 900 python bench.py > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.log; echo "bench rc=$?"; tail -3 gpurun_out/r5b/bench.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5b/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["paths"], indent=1)[:3000])
print(json.dumps(d["extra"]["itemknn"], indent=1)[:2500])
PY
