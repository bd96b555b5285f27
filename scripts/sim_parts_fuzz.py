"""Random shapes on which the heavy columns of a multi-GPU part can dominate it: every part of a G-way build with the heavy-column rule
(sim.hip, MI355REC_SIM_PACKED_DEMOTE) chosen by the library, forced off and forced on, and without the packed-counts kernel -- outputs compared
cell for cell, the schedules printed (did the rule fire?).  Run on the GPU box; prints one line per case and a verdict."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 14
modes = (("library", {}), ("rule off", {"MI355REC_SIM_PACKED_DEMOTE": "0"}), ("rule on", {"MI355REC_SIM_PACKED_DEMOTE": "1"}), ("32-bit only", {"MI355REC_SIM_PACKED": "0"}))
bad = fired = 0
for case in range(n_cases):
    n_users = int(rng.choice([40000, 90000, 138493, 250000]))
    n_items = int(rng.choice([1500, 4000, 9000, 17770, 26744]))
    nnz = int(rng.choice([3e6, 8e6, 2e7]))
    nnz = min(nnz, n_users * n_items // 8)
    zipf = float(rng.choice([0.6, 0.8, 1.0]))
    G = int(rng.choice([2, 4, 8]))
    topK = int(rng.choice([10, 100]))
    sim_kind = str(rng.choice(["cosine", "jaccard", "tversky"]))
    X = synthetic_urm(n_users, n_items, nnz, 1, min(n_items - 1, 4000), seed=1000 + case, values="binary", zipf_exponent=zipf)
    w = -(-n_items // G)
    buf = DeviceArray(2 * w * topK)
    ref, sched = None, {}
    same = True
    shrink = int(rng.integers(0, 3))
    for tag, env in modes:
        for k, v in env.items():
            os.environ[k] = v
        s = Compute_Similarity_MI355X(X, topK=topK, shrink=shrink, normalize=True, similarity=sim_kind)
        outs = []
        for part in range(G):
            s.compute_part_device(part, G, buf.address(), buf.address(w * topK)); s.synchronize()
            rows = len(s.part_columns(part, G))        # (a part with fewer columns than the widest leaves the last row of the slabs alone)
            outs.append(buf.to_host().reshape(2, w, topK)[:, :rows].copy())
            if part == 0:
                sched[tag] = s.schedule_info()
        cnt = max(1, len(s.part_columns(0, G)) // 3)
        s.compute_part_chunk_device(0, G, 0, cnt, buf.address(), buf.address(cnt * topK)); s.synchronize()
        outs.append(buf.to_host()[:2 * cnt * topK].copy())
        s.close()
        for k in env:
            del os.environ[k]
        if ref is None:
            ref = outs
        else:
            same = same and all(np.array_equal(a, b) for a, b in zip(ref, outs))
    buf.close()
    rule_fired = sched["library"] != sched["rule off"]
    fired += rule_fired
    bad += not same
    print("%3d %s users %6d items %5d nnz %8d zipf %.1f G %d topK %3d %-7s shrink %d  part 0 schedule: library %s, rule off %s, rule on %s, 32-bit %s%s" % (
        case, "OK " if same else "BAD", n_users, n_items, X.nnz, zipf, G, topK, sim_kind, shrink, sched["library"], sched["rule off"], sched["rule on"], sched["32-bit only"],
        "  <- rule fired" if rule_fired else ""), flush=True)
print("%d cases, %d differences, the library's rule fired in %d" % (n_cases, bad, fired))
sys.exit(1 if bad else 0)
