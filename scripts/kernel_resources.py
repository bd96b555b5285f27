"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel (VGPRs, AGPRs, scratch, LDS, occupancy).
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> usage.txt; python scripts/kernel_resources.py usage.txt [filter]"""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass-analysis", line) or re.search(r"remark: (.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    except OSError:
        name = r["name"]
    if flt and flt not in name:
        continue
    name = re.sub(r"\(.*", "", name.replace("void ", ""))
    print("%-70s VGPR %3s AGPR %3s scratch %4s LDS %6s occ %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                                                               r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
