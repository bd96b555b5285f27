"""Kernel time of the similarity build by lanes per walk entry (MI355REC_SIM_G) at a shape; constructor time next to it."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from bench import load_urm, TOPK
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
name = sys.argv[1] if len(sys.argv) > 1 else "ml20m"
urm = load_urm(name)
cases = [("binary", urm)]
if "--ratings" in sys.argv:
    real = urm.copy(); real.data = (1 + (np.arange(real.nnz) % 5)).astype(np.float32)
    cases.append(("ratings", real))
if "--wide" in sys.argv:
    real = urm.copy(); real.data = (1 + (np.arange(real.nnz) % 5) + 1e-3 * np.random.default_rng(0).random(real.nnz)).astype(np.float32)
    cases = [("jittered ratings (8-byte cells)", real)]
for label, X in cases:
    for G in sys.argv[2].split(","):
        os.environ["MI355REC_SIM_G"] = G
        t0 = time.perf_counter()
        s = Compute_Similarity_MI355X(X, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
        s.synchronize()
        t_c = time.perf_counter() - t0
        s.compute_slabs()
        best = min((s.compute_slabs(), s.stats()["kernel_ms"])[1] for _ in range(5))
        os.environ["MI355REC_SIM_PHASES"] = "1"
        sys.stderr.flush()
        s.compute_slabs()
        del os.environ["MI355REC_SIM_PHASES"]
        print("%s %s G=%s: kernel %.3f ms (best of 5), constructor %.2f ms" % (name, label, G, best, t_c * 1e3), flush=True)
        s.close()
