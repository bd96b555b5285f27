"""Packed-counts kernel on a URM whose most popular items have 65 536 users or more (their columns are accumulated in parts and added up by the
32-bit launch): results against the one-launch build, bit for bit.  Usage: sim_packed_many_users.py <users> <items> <nnz>  (200000 12000 12000000)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sps
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
n_users, n_items = int(sys.argv[1]), int(sys.argv[2])
X = synthetic_urm(n_users, n_items, int(sys.argv[3]), 5, min(400, n_items - 1), seed=7)
print("URM", X.shape, X.nnz, "users of item 0:", X[:, 0].nnz, flush=True)
out = {}
for mode in ("nopacked", "heavy0", "heavy1"):
    os.environ.pop("MI355REC_SIM_NO_PACKED", None); os.environ.pop("MI355REC_SIM_PACKED_HEAVY", None)
    if mode == "nopacked": os.environ["MI355REC_SIM_NO_PACKED"] = "1"
    if mode == "heavy0": os.environ["MI355REC_SIM_PACKED_HEAVY"] = "0"
    s = Compute_Similarity_MI355X(X, topK=50, shrink=0, normalize=True, similarity="cosine")
    idx, val, _ = s.compute_slabs()
    out[mode] = (idx.copy(), val.copy())
    print(mode, "kernel %.3f ms" % s.stats()["kernel_ms"], flush=True)
    s.close()
for mode in ("heavy0", "heavy1"):
    print(mode, "idx equal", np.array_equal(out[mode][0], out["nopacked"][0]), "val equal", np.array_equal(out[mode][1], out["nopacked"][1]), flush=True)
