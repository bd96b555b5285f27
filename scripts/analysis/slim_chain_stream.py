"""A SLIM-BPR epoch stream with the reference sampler's distribution at a named shape (uniform user, uniform seen item, uniform unseen
item), written for slim_chain.c / slim_chain_weighted.c.  Usage: slim_chain_stream.py [ml20m] [out.bin]"""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
shape = sys.argv[1] if len(sys.argv) > 1 else "ml20m"
X = named_urm(shape, "binary")
n_users, n_items = X.shape
rng = np.random.default_rng(1)
n = n_users + 1
L = np.diff(X.indptr)
u = rng.integers(0, n_users, n)
i = X.indices[X.indptr[u] + (rng.random(n) * L[u]).astype(np.int64)]
j = rng.integers(0, n_items, n)
# rejection (few collisions): redraw until not in profile
for t in range(n):
    row = X.indices[X.indptr[u[t]]:X.indptr[u[t] + 1]]
    while True:
        p = np.searchsorted(row, j[t])
        if p < len(row) and row[p] == j[t]:
            j[t] = rng.integers(0, n_items)
        else:
            break
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/slim_stream_%s.bin" % shape
with open(out, "wb") as f:
    np.array([n_users, n_items], np.int32).tofile(f); np.array([X.nnz], np.int64).tofile(f); np.array([n], np.int32).tofile(f)
    X.indptr.astype(np.int32).tofile(f); X.indices.astype(np.int32).tofile(f)
    u.astype(np.int32).tofile(f); i.astype(np.int32).tofile(f); j.astype(np.int32).tofile(f)
print("written", n, "steps; mean L of sampled users", L[u].mean())
