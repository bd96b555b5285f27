#!/bin/bash
# round 3, call B: MF pair tasks / used-slot loop / occ8, sim int32 mode + fixes, slim element-wise bar, bench with all CPU baselines
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mf_gpu.py -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r3b/pytest_mf.log
( timeout 400 python scripts/mf_group.py 1 8 32 64 --epochs 20 2>&1 | tail -12 ) > gpurun_out/r3b/mf_group.log
( MI355REC_MF_GROUP_OCC8=1 timeout 200 python scripts/mf_group.py 32 64 --epochs 20 2>&1 | tail -4 ) > gpurun_out/r3b/mf_group_occ8.log
( timeout 200 python scripts/mf_group.py 32 --k 64 --epochs 20 2>&1 | tail -4 ) > gpurun_out/r3b/mf_group_k64.log
( timeout 900 python -m pytest tests/test_sim_gpu.py tests/test_rccl_direct.py -q -m gpu -rx 2>&1 | tail -40 ) > gpurun_out/r3b/pytest_sim.log
( timeout 600 python -m pytest tests/test_slim_gpu.py -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r3b/pytest_slim.log
( timeout 300 python scripts/sim_phases.py 2>&1 | tail -30 ) > gpurun_out/r3b/sim_phases.log
( timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err )
for f in pytest_mf mf_group mf_group_occ8 mf_group_k64 pytest_sim pytest_slim sim_phases; do echo "== $f"; tail -40 gpurun_out/r3b/$f.log; done
python - <<'P'
import json
d = json.loads(open("gpurun_out/r3b/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("cpu", d.get("cpu_baseline"))
for k, v in d["extra"].get("paths", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("samples_per_s", "frac", "seconds_per_epoch", "speedup_vs_cpu_baseline", "avg_launch_us", "us_per_step")}, (v.get("cpu_baseline") or {}).get("value"))
print(d["extra"].get("paths_error"))
print({k: v for k, v in d["extra"]["itemknn"].items() if k in ("cosine_build_s", "fit_s", "kernel_ms_this_rank")})
P
tail -3 gpurun_out/r3b/bench.err
