"""One 1024-thread workgroup per CU against two 512-thread ones, same work: a 138 493 x 9 000 binary URM with 20 M stored values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
X = synthetic_urm(138493, n_items, 20000263, 20, min(9254, n_items - 1), seed=5)
print("URM", X.shape, X.nnz, flush=True)
for one in (False, True, False, True):
    if one: os.environ["MI355REC_SIM_ONE_WG_PER_CU"] = "1"
    else: os.environ.pop("MI355REC_SIM_ONE_WG_PER_CU", None)
    s = Compute_Similarity_MI355X(X, topK=100, shrink=0, normalize=True, similarity="cosine")
    s.compute_slabs()
    best = min((s.compute_slabs(), s.stats()["kernel_ms"])[1] for _ in range(5))
    os.environ["MI355REC_SIM_PHASES"] = "1"
    s.compute_slabs()
    del os.environ["MI355REC_SIM_PHASES"]
    print("one 1024-thread workgroup per CU" if one else "512-thread workgroups, several per CU", "kernel %.3f ms" % best, flush=True)
    s.close()
