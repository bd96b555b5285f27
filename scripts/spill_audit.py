"""Where do a kernel's scratch (spill) instructions sit?  Reads hipcc -S output and reports, per kernel whose mangled name
contains the filter, every scratch_load / scratch_store with the loop depth of its basic block (from the assembler's own
"Loop: Header=... Depth=N" block comments) next to the depth of the blocks that hold the kernel's hot instruction.
Usage: hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s; python scripts/spill_audit.py x.s <name filter> <hot opcode>"""
import re
import sys


def audit(path, name_filter, hot_opcode):
    """{mangled kernel name: {"n": instructions, "scratch": {(kind, depth): count}, "hot": {depth: count}}}"""
    lines = open(path).read().split("\n")
    kernel, depth, out = None, 0, {}
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            kernel = m.group(1) if name_filter in m.group(1) else None
            depth = 0
            if kernel:
                out[kernel] = {"scratch": {}, "hot": {}, "n": 0}
        elif kernel and re.match(r"^\.LBB\d+_\d+:", ln):
            depths = [int(d) for d in re.findall(r"Depth=(\d+)", ln)]
            j = i + 1
            while j < len(lines) and re.match(r"^\s+;", lines[j]):
                depths += [int(d) for d in re.findall(r"Depth=(\d+)", lines[j])]
                j += 1
            depth = max(depths) if depths else 0
        elif kernel and ln.startswith("\t.end_amdhsa_kernel"):
            kernel = None
        elif kernel:
            op = ln.strip().split(" ")[0].split("\t")[0]
            if op.startswith("scratch_"):
                key = (op.split("_")[1], depth)
                out[kernel]["scratch"][key] = out[kernel]["scratch"].get(key, 0) + 1
            elif op.startswith(hot_opcode):
                out[kernel]["hot"][depth] = out[kernel]["hot"].get(depth, 0) + 1
            if op and not op.startswith((";", ".")):
                out[kernel]["n"] += 1
        i += 1
    return {k: v for k, v in out.items() if v["n"]}


if __name__ == "__main__":
    hot = sys.argv[3]
    for k, v in audit(sys.argv[1], sys.argv[2], hot).items():
        sc = ", ".join("%d %ss at loop depth %d" % (n, kind, d) for (kind, d), n in sorted(v["scratch"].items(), key=lambda t: (t[0][1], t[0][0]))) or "none"
        ht = ", ".join("%d at depth %d" % (n, d) for d, n in sorted(v["hot"].items()))
        print("%s\n    %d instructions; scratch: %s\n    %s: %s" % (k, v["n"], sc, hot, ht))
