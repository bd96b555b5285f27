#!/bin/bash
# round 3, call F: the wave-per-column norm sums + the host-side changes since the last full suite, then the final default bench
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
( timeout 700 python -m pytest tests/test_sim_gpu.py tests/test_sharding_gpu.py tests/test_scoring_gpu.py tests/test_rccl_direct.py tests/test_graph_gpu.py tests/test_ease_gpu.py -q -m gpu -k "not netflix" -rfx --timeout 400 2>&1 | grep -v "^  File\|^Extension\|DeprecationWarning\|^$" | tail -25 ) > gpurun_out/r3f/pytest.log
cat gpurun_out/r3f/pytest.log | cut -c1-250
( timeout 420 python bench.py > gpurun_out/r3f/bench_default.json 2> gpurun_out/r3f/bench_default.err )
tail -4 gpurun_out/r3f/bench_default.err | cut -c1-200
python - <<'P'
import json
d = json.loads(open("gpurun_out/r3f/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print({k: v for k, v in d["extra"]["itemknn"].items() if k in ("cosine_build_s", "create_s", "fit_s", "kernel_ms_this_rank")})
for k, v in d["extra"].get("paths", {}).items():
    print("  ", k, {a: b for a, b in v.items() if a in ("samples_per_s", "frac", "seconds_per_epoch", "speedup_vs_cpu_baseline", "avg_launch_us", "us_per_step", "users_per_s", "traffic")})
P
