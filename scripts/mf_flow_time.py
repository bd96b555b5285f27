"""BPR-MF k=128 B=1000 at the ML-20M shape (the headline): the dataflow epoch (one persistent launch per epoch) against the chain of
139 mini-batch launches (MI355REC_MF_NO_FLOW=1).  Usage: mf_flow_time.py [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_urm, BATCH, K_FACTORS
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
urm = load_urm("ml20m")
res = {}
for mode in ("sgd", "adagrad"):
    for flow in (True, False):
        if flow:
            os.environ.pop("MI355REC_MF_NO_FLOW", None)
        else:
            os.environ["MI355REC_MF_NO_FLOW"] = "1"
        m = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3, sgd_mode=mode,
                                             init_std_dev=0.1, random_seed=42)
        m.epochIteration_Cython(10)
        best = 0.0
        for _ in range(3):
            m.epochIteration_Cython(epochs)
            st = m.stats()
            best = max(best, st["n_units"] / (st["call_ms"] * 1e-3))
        m.set_profiling(5 if flow else 5 * 139)
        m.epochIteration_Cython(5)
        pst = m.stats()
        per_epoch = (urm.shape[0] // BATCH + 1) * BATCH
        print("%-8s %-28s %8.1f M samples/s (%.3f ms per epoch); timed launches %d, mean %.1f us" % (
            mode, "dataflow epoch" if flow else "one launch per mini-batch", best / 1e6, per_epoch / best * 1e3, pst["n_timed"],
            pst["kernel_ms"] / max(1, pst["n_timed"]) * 1e3), flush=True)
        res[(mode, flow)] = (m.get_USER_factors(), m.get_ITEM_factors())
        m.close()
    same = all(np.array_equal(a, b) for a, b in zip(res[(mode, True)], res[(mode, False)]))
    print("%-8s factors after the same epochs bit-identical: %s" % (mode, same), flush=True)
