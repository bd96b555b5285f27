"""SLIM-BPR epoch timing at the ML-1M and ML-20M shapes (diagnostics; run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys2019_deeplearning_evaluation_amd import SLIM_BPR_MI355X_Epoch
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
for shape in (sys.argv[1:] or ("ml1m", "ml20m")):
    X = named_urm(shape, "binary")
    for symmetric in (False, True):
        for mode in ("sgd", "adagrad"):
            ep = SLIM_BPR_MI355X_Epoch(X, symmetric=symmetric, learning_rate=1e-3, sgd_mode=mode, random_seed=3)
            ep.epochIteration_Cython()
            t = time.perf_counter()
            n = 3
            ep.epochIteration_Cython(n)
            dt = (time.perf_counter() - t) / n
            st = ep.stats()
            print("%s %-9s %-7s %8.3f ms per epoch (flow kernel %.3f ms), %6.2f M samples/s, owned rows / cold steps %s" % (
                shape, "symmetric" if symmetric else "dense", mode, dt * 1e3, st["kernel_ms"] / n, (X.shape[0] + 1) / dt / 1e6,
                ep.schedule_info()), flush=True)
            ep._dealloc()
