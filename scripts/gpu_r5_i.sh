#!/bin/bash
# round 5, call i (the MI355REC_SIM_KTH12 switch it used was removed afterwards: the one-pass bound did not pay, see sim.hip): where the "K-th maximum" phase goes (skew of the scan against the selection itself), and the one-pass 12-bit bound.
mkdir -p gpurun_out/r5i
export SIM_PHASES_BINARY_ONLY=1
for k in 0 1; do
  MI355REC_SIM_KTH12=$k python scripts/sim_phases.py > gpurun_out/r5i/phases_kth12_$k.txt 2>&1
  echo "== KTH12=$k"; head -3 gpurun_out/r5i/phases_kth12_$k.txt | cut -c1-700
done
MI355REC_SIM_KTH12=1 timeout 300 python -m pytest tests/test_sim_gpu.py -q -m gpu -x -k "threshold_first or golden or seeded" 2>&1 | tail -2 | tee gpurun_out/r5i/pytest_kth12.txt
