#!/bin/bash
# round 3, call E1: what the driver runs at round end (whole GPU suite, smoke, default bench) + a 2-rank dry run of the N > 1 bench path on one GPU (gloo)
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
( timeout 1300 python -m pytest tests -q -m gpu -rfx --durations=10 --timeout 500 2>&1 | grep -v "^  File\|^Extension\|DeprecationWarning\|^$" | tail -60 ) > gpurun_out/r3e/pytest_gpu.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3e/smoke.log 2>&1 )
( timeout 420 python bench.py > gpurun_out/r3e/bench_default.json 2> gpurun_out/r3e/bench_default.err )
( BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --no-netflix > gpurun_out/r3e/bench_n2_dryrun.json 2> gpurun_out/r3e/bench_n2_dryrun.err )
echo "== pytest"; cat gpurun_out/r3e/pytest_gpu.log | cut -c1-300 | tail -45
echo "== smoke"; tail -4 gpurun_out/r3e/smoke.log
echo "== bench.err"; tail -6 gpurun_out/r3e/bench_default.err | cut -c1-300
echo "== dryrun.err"; tail -6 gpurun_out/r3e/bench_n2_dryrun.err | cut -c1-300
python - <<'P'
import json
for f in ("gpurun_out/r3e/bench_default.json", "gpurun_out/r3e/bench_n2_dryrun.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "n_gpus", d["n_gpus"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
        for k, v in d["extra"].get("paths", {}).items():
            print("  ", k, {a: b for a, b in v.items() if a in ("samples_per_s", "frac", "seconds_per_epoch", "speedup_vs_cpu_baseline", "avg_launch_us", "us_per_step", "users_per_s")})
        print("  ", d["extra"].get("paths_error"), d["extra"].get("ials_error"), d["extra"].get("communicator"))
        print("   itemknn", {k: v for k, v in d["extra"]["itemknn"].items() if k in ("cosine_build_s", "fit_s", "kernel_ms_this_rank", "transport")})
        print("   ials", {k: v for k, v in d["extra"].get("ials", {}).items() if k != "emulated_8_way"})
    except Exception as e:
        print(f, "no bench line:", e)
P
