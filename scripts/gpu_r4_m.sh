#!/bin/bash
# Round 4, session M: HBM ceiling for the mini-batch kernels' pattern; similarity constructor after the cost kernel.
mkdir -p gpurun_out/r4m
timeout 120 bash -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/row_gather_scatter.hip -o /tmp/rgs 2>/dev/null && /tmp/rgs" > gpurun_out/r4m/row_gather_scatter.txt 2>&1; cat gpurun_out/r4m/row_gather_scatter.txt
timeout 100 python scripts/sim_create_phases.py binary > gpurun_out/r4m/sim_create.txt 2>&1; echo "sim create rc=$?"; tail -8 gpurun_out/r4m/sim_create.txt
