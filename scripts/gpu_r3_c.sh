#!/bin/bash
# round 3, call C: whole GPU suite (minus the Netflix-shape cases) on the rocPRIM build, group schedule, new GEMM, bench with watchdog
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu -k "not netflix" -rfx --durations=12 2>&1 | grep -v "^  File\|^Extension\|DeprecationWarning\|^$" | tail -90 ) > gpurun_out/r3c/pytest.log
( timeout 300 python scripts/mf_group.py 8 32 64 --epochs 20 2>&1 | tail -6 ) > gpurun_out/r3c/mf_group.log
( MI355REC_MF_GROUP_PER_MEMBER_SCHEDULE=1 timeout 200 python scripts/mf_group.py 32 --epochs 20 2>&1 | tail -3 ) > gpurun_out/r3c/mf_group_per_member.log
( timeout 420 python bench.py --steps 100 --warmup 10 --cpu-seconds 6 > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err )
for f in pytest mf_group mf_group_per_member; do echo "== $f"; cat gpurun_out/r3c/$f.log | cut -c1-400; done
echo "== bench.err"; tail -40 gpurun_out/r3c/bench.err | cut -c1-300
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r3c/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
    print("cpu", d.get("cpu_baseline"))
    for k, v in d["extra"].get("paths", {}).items():
        print(k, {a: b for a, b in v.items() if a in ("samples_per_s", "frac", "seconds_per_epoch", "speedup_vs_cpu_baseline", "avg_launch_us", "us_per_step", "users_per_s")}, (v.get("cpu_baseline") or {}).get("value"))
    print(d["extra"].get("paths_error"), d["extra"].get("ials_error"))
    print({k: v for k, v in d["extra"]["itemknn"].items() if k in ("cosine_build_s", "fit_s", "kernel_ms_this_rank")})
    print({k: v for k, v in d["extra"].get("ials", {}).items() if k != "emulated_8_way"})
    print(d["extra"].get("ials", {}).get("emulated_8_way"))
except Exception as e:
    print("no bench line:", e)
P
