#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ials
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/a -o p -- python $R/scratch/ials_prof.py > $OUT/a.log 2>&1
echo rc=$?
cd $R; python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_ials/a/**/*counter_collection.csv",recursive=True)
if f:
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,d in agg.items():
        if "ials_row" in k or "gram" in k:
            print(k); 
            for c,v in sorted(d.items()): print("   %-22s %.4e"%(c,v))
PY
