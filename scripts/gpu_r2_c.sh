#!/bin/bash
mkdir -p gpurun_out/r2c
for v in base "MI355REC_MF_PREFETCH=1" "MI355REC_MF_LPR=16" "MI355REC_MF_LPR=16 MI355REC_MF_PREFETCH=1"; do
  echo "=== $v" >> gpurun_out/r2c/ticks.log
  if [ "$v" = base ]; then timeout 300 python scratch/mf_ticks.py >> gpurun_out/r2c/ticks.log 2>&1; else env $v timeout 300 python scratch/mf_ticks.py >> gpurun_out/r2c/ticks.log 2>&1; fi
done
cat gpurun_out/r2c/ticks.log
