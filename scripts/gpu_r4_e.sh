#!/bin/bash
# Round 4, session E: the whole -m gpu suite on the current tree, similarity constructor phases, the default bench line.
mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests -q -m gpu -x --durations=12 > gpurun_out/r4e/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4e/pytest_gpu.txt
tail -22 gpurun_out/r4e/pytest_gpu.txt
timeout 100 python scripts/sim_create_phases.py binary > gpurun_out/r4e/sim_create.txt 2>&1; echo "sim create rc=$?"; tail -9 gpurun_out/r4e/sim_create.txt
timeout 400 python bench.py > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r4e/bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r4e/bench.json"))
print("value %.1f M samples/s, ms_per_step %.3f, roofline frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"]))
for k, v in d.get("paths", {}).items():
    print("  %-52s %s" % (k, v))
PY
