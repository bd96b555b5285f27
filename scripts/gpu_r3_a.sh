#!/bin/bash
# round 3, call A: new numerics tests + MF fused / group + sim phases
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mf_gpu.py -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r3a/pytest_mf.log
( timeout 600 python -m pytest tests/test_sim_gpu.py tests/test_rccl_direct.py tests/test_slim_gpu.py -q -m gpu -rx 2>&1 | tail -40 ) > gpurun_out/r3a/pytest_sim_slim.log
( timeout 400 python scripts/mf_group.py 1 4 8 16 32 64 --epochs 20 2>&1 | tail -12 ) > gpurun_out/r3a/mf_group.log
( MI355REC_MF_NO_FUSE=1 timeout 200 python scripts/mf_group.py 1 32 --epochs 20 2>&1 | tail -4 ) > gpurun_out/r3a/mf_group_nofuse.log
( timeout 300 python scripts/sim_phases.py 2>&1 | tail -30 ) > gpurun_out/r3a/sim_phases.log
( timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err )
for f in pytest_mf pytest_sim_slim mf_group mf_group_nofuse sim_phases; do echo "== $f"; tail -40 gpurun_out/r3a/$f.log; done
head -c 1500 gpurun_out/r3a/bench.json; tail -3 gpurun_out/r3a/bench.err
