#!/bin/bash
# Round 5, session D: two-stage IALS epochs (systems to HBM, wave-specialised solve kernel, two workgroups per CU).
mkdir -p gpurun_out/r5d
for env in "MI355REC_IALS_TWO_STAGE=1" "MI355REC_IALS_TWO_STAGE=1 MI355REC_IALS_SAME_PANEL_WAVE=1" "MI355REC_IALS_TWO_STAGE=0"; do
  echo "== $env"
  env $env timeout 300 python scratch/ials_time.py 2>&1 | tail -1
  env $env MI355REC_IALS_PHASES=1 timeout 300 python scratch/ials_time.py 2>&1 | tail -3 | head -2
done | tee gpurun_out/r5d/ials_time.txt
if [ "$1" != "quick" ]; then
timeout 900 python -m pytest tests/test_ials_gpu.py -q -m gpu -x > gpurun_out/r5d/pytest_ials.txt 2>&1; echo "ials rc=$?"; tail -5 gpurun_out/r5d/pytest_ials.txt
fi
