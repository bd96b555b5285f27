#!/bin/bash
# Round 5, session D: two-stage IALS epochs (systems to HBM, wave-specialised solve kernel, two workgroups per CU).
mkdir -p gpurun_out/r5d
for mode in 1 0; do
  echo "== MI355REC_IALS_TWO_STAGE=$mode"
  MI355REC_IALS_TWO_STAGE=$mode timeout 300 python scratch/ials_time.py 2>&1 | tail -2
  MI355REC_IALS_TWO_STAGE=$mode MI355REC_IALS_PHASES=1 timeout 300 python scratch/ials_time.py 2>&1 | tail -3
done | tee gpurun_out/r5d/ials_time.txt
timeout 900 python -m pytest tests/test_ials_gpu.py -q -m gpu -x > gpurun_out/r5d/pytest_ials.txt 2>&1; echo "ials rc=$?"; tail -5 gpurun_out/r5d/pytest_ials.txt
