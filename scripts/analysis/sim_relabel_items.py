"""Same-address pressure: the ML-20M-shaped URM with its item ids relabelled at random (popularity no longer follows the id, so the sorted
profiles of different users no longer hold the same popular items at the same positions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sps
from bench import load_urm, TOPK
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
X = load_urm("ml20m")
perm = np.random.default_rng(1).permutation(X.shape[1])
Xp = sps.csr_matrix(X[:, perm]); Xp.sort_indices()
for label, M in (("ids by popularity", X), ("ids relabelled at random", Xp)):
    s = Compute_Similarity_MI355X(M, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
    s.compute_slabs()
    best = min((s.compute_slabs(), s.stats()["kernel_ms"])[1] for _ in range(5))
    os.environ["MI355REC_SIM_PHASES"] = "1"
    s.compute_slabs()
    del os.environ["MI355REC_SIM_PHASES"]
    print(label, "kernel %.3f ms" % best, flush=True)
    s.close()
