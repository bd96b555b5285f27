"""Part 0 (and 7) of an 8-way sharded similarity build at the ML-20M shape, whole and in pieces, with the packed-counts kernel forced on,
forced off and chosen by the library; then the phase clocks of three column ranges of the whole shape.  What sharding.cost_sized_pieces /
default_chunks and sim.hip's PACKED_MAX_PAIRS_PER_COLUMN were decided on.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_urm, TOPK
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
from recsys2019_deeplearning_evaluation_amd.sharding import cost_sized_pieces, piece_order, FIXED_PAIRS_PER_CELL

urm = load_urm("ml20m")
n, G = urm.shape[1], 8
w = -(-n // G)
buf = DeviceArray(2 * w * TOPK)


def run(tag, env):
    for k, v in env.items():
        os.environ[k] = v
    s = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
    cost = np.asarray(s.column_costs(), np.float64)
    cols0 = s.part_columns(0, G)
    row_cost = np.zeros(w); row_cost[:len(cols0)] = cost[cols0] + FIXED_PAIRS_PER_CELL * n
    rows = cost_sized_pieces(row_cost, 4)
    print("%s" % tag, flush=True)
    for part in (0, 7):
        best = 1e9
        for _ in range(4):
            s.compute_part_device(part, G, buf.address(), buf.address(w * TOPK)); s.synchronize()
            best = min(best, s.stats()["kernel_ms"])
        print("    part %d of 8 in one go: %.4f ms   (work items, split columns, parts) = %s" % (part, best, s.schedule_info()), flush=True)
    pieces = [(0, 8), (0, 64), (0, 512), (8, 504)] + [(rows[c][0], rows[c][1] - rows[c][0]) for c in piece_order(rows)]
    for r0, cnt in pieces:
        best = 1e9
        for _ in range(3):
            s.compute_part_chunk_device(0, G, r0, cnt, buf.address(), buf.address(cnt * TOPK)); s.synchronize()
            best = min(best, s.stats()["kernel_ms"])
        print("    part 0, rows %4d + %4d: %.4f ms, %.2f M pair-adds per column   %s" % (r0, cnt, best, cost[cols0[r0:r0 + cnt]].mean() / 1e6, s.schedule_info()), flush=True)
    if not env:
        print("    cost_sized_pieces (row order; built in reversed order): %s" % (rows,), flush=True)
    for a, b in ((0, 512), (0, 4096), (512, 4096)):
        best = 1e9
        for _ in range(3):
            s.compute_slabs(a, b); best = min(best, s.stats()["kernel_ms"])
        print("    columns [%d, %d) of the whole shape: %.4f ms, %.2f M pair-adds per column" % (a, b, best, cost[a:b].mean() / 1e6), flush=True)
        if os.environ.get("SIM_PART_PHASES"):
            os.environ["MI355REC_SIM_PHASES"] = "1"
            sys.stderr.flush()
            s.compute_slabs(a, b)
            sys.stderr.flush()
            del os.environ["MI355REC_SIM_PHASES"]
    s.close()
    for k in env:
        del os.environ[k]


run("the library's choice (packed-counts kernel below 1.0 M pair-adds per column of the call)", {})
run("MI355REC_SIM_PACKED=1 (packed-counts kernel wherever it applies: the rule until this measurement was 2.0 M)", {"MI355REC_SIM_PACKED": "1"})
run("MI355REC_SIM_PACKED=0 (32-bit kernel only)", {"MI355REC_SIM_PACKED": "0"})
