#!/bin/bash
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests/test_sim_gpu.py tests/test_scoring_gpu.py tests/test_rccl_direct.py tests/test_sharding_gpu.py tests/test_ials_gpu.py -q -m gpu --maxfail=8 2>&1 | grep -v "^  File\|^Extension" | tail -60 > gpurun_out/r2f/pytest.log
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/r2f/bench_n1.json 2> gpurun_out/r2f/bench_n1.err
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=2 BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1
(RANK=0 LOCAL_RANK=0 timeout 600 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2f/bench_n2_gloo.json 2> gpurun_out/r2f/bench_n2_gloo.err &
 RANK=1 LOCAL_RANK=1 timeout 600 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> gpurun_out/r2f/bench_n2_gloo_r1.err; wait)
tail -25 gpurun_out/r2f/pytest.log; python - <<'PY'
import json
for f in ("gpurun_out/r2f/bench_n1.json","gpurun_out/r2f/bench_n2_gloo.json"):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], json.dumps(d["extra"].get("itemknn",{}))[:1500])
        print(json.dumps(d["extra"].get("paths",{}))[:3000])
    except Exception as e: print(f, "ERR", e)
PY
tail -5 gpurun_out/r2f/bench_n1.err gpurun_out/r2f/bench_n2_gloo.err gpurun_out/r2f/bench_n2_gloo_r1.err
