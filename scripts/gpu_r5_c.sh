#!/bin/bash
# Round 5, session C: resident-URM constructor (test + bench fit), BASELINE config 4 (Netflix shape) evidence, launcher dry run.
mkdir -p gpurun_out/r5c
timeout 600 python -m pytest tests/test_sim_gpu.py -q -m gpu -x -k "resident or threshold_first" > gpurun_out/r5c/pytest_resident.txt 2>&1; echo "resident rc=$?"; tail -3 gpurun_out/r5c/pytest_resident.txt
( MI355REC_SIM_CREATE_PHASES=1 timeout 300 python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from bench import load_urm, TOPK
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X, ResidentURM
urm = load_urm("ml20m")
res = ResidentURM(urm)
for label, kw in (("host arrays (PCIe upload)", {}), ("resident URM", {"resident": res})):
    for rep in range(2):
        print("----", label, "rep", rep, flush=True)
        t = time.perf_counter()
        s = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, **kw)
        s.synchronize()
        print("constructor %.3f ms (phase clocks drain the stream between phases)" % ((time.perf_counter() - t) * 1e3), flush=True)
        s.close()
PY
) > gpurun_out/r5c/sim_create_phases.txt 2>&1; tail -32 gpurun_out/r5c/sim_create_phases.txt
timeout 300 python bench.py --no-paths --no-ials --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r5c/bench_ml20m_short.json 2> gpurun_out/r5c/bench_ml20m_short.log; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/bench_ml20m_short.json").read().strip().splitlines()[-1])
ik = d["extra"]["itemknn"]
print({k: ik[k] for k in ("cosine_build_s", "create_s", "fit_s", "create_incl_pcie_upload_s", "fit_incl_pcie_upload_s", "kernel_ms_this_rank")})
PY
( BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 20 --warmup 3 --no-netflix --no-ials > gpurun_out/r5c/bench_n2_dryrun_self_launched.json 2> gpurun_out/r5c/bench_n2_dryrun.err ); echo "n2 dry run rc=$?"; tail -2 gpurun_out/r5c/bench_n2_dryrun.err; python -c "
import json; d=json.loads(open('gpurun_out/r5c/bench_n2_dryrun_self_launched.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['extra']['communicator'], d['extra'].get('launched_by'), d['extra']['itemknn']['cosine_build_s'])"
timeout 100 python bench.py --gpus 8 --steps 2 > gpurun_out/r5c/bench_n8_refused.out 2> gpurun_out/r5c/bench_n8_refused.err; echo "n8 on one GPU rc=$? (must be non-zero)"; tail -1 gpurun_out/r5c/bench_n8_refused.err
timeout 900 python bench.py --workload netflix --no-paths --no-ials --steps 50 --warmup 5 --cpu-seconds 6 > gpurun_out/r5c/bench_n1_netflix.json 2> gpurun_out/r5c/bench_n1_netflix.log; echo "netflix rc=$?"; tail -2 gpurun_out/r5c/bench_n1_netflix.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/bench_n1_netflix.json").read().strip().splitlines()[-1])
ik = d["extra"]["itemknn"]
print({k: ik[k] for k in ("cosine_build_s", "create_s", "fit_s", "fit_incl_pcie_upload_s", "kernel_ms_this_rank")})
e = ik["emulated_8_way"]
print(e["slowest_part_ms"], e["kernel_speedup_vs_1gpu"], e["predicted_build_speedup_one_exchange_at_the_end"], e["predicted_build_speedup"])
PY
