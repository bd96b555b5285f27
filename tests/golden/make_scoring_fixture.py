"""Reference-generated fixture for device scoring + ranking: the REFERENCE's own
BaseMatrixFactorizationRecommender._compute_item_score (Base/BaseMatrixFactorizationRecommender.py:38-77) and BaseRecommender.recommend
(Base/BaseRecommender.py:131-222), imported from /root/reference, on small seeded inputs -- scores after the seen-item filter and the
ranked lists, without and with biases, with and without `items_to_compute`.  Writes tests/golden/scoring.npz.
Run where the reference tree exists:  python tests/golden/make_scoring_fixture.py"""
import os
import sys

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_loader                                                   # noqa: E402

Base = ref_loader.load_python_reference("Base.BaseMatrixFactorizationRecommender", "BaseMatrixFactorizationRecommender")
assert Base is not None, "needs /root/reference"

rng = np.random.default_rng(20260924)
n_users, n_items, k, cutoff = 240, 417, 24, 15
dense = rng.random((n_users, n_items)) < 0.06
dense[5, :] = False                                 # a user who has seen nothing
dense[6, :] = True; dense[6, :9] = False            # a user with fewer unseen items than the cut-off
X = sps.csr_matrix(dense.astype(np.float32))
X.sort_indices()
U = rng.normal(0, 0.3, (n_users, k)); V = rng.normal(0, 0.3, (n_items, k))
bu = rng.normal(0, 0.5, n_users); bi = rng.normal(0, 0.5, n_items); mu = 0.37
users = rng.choice(n_users, 64, replace=False)
users[:2] = (5, 6)
allowed = np.sort(rng.choice(n_items, 150, replace=False))

out = {"indptr": X.indptr.astype(np.int32), "indices": X.indices.astype(np.int32), "shape": np.array(X.shape), "U": U, "V": V, "bu": bu,
       "bi": bi, "mu": np.array(mu), "users": users.astype(np.int32), "allowed": allowed.astype(np.int32), "cutoff": np.array(cutoff)}
for tag, use_bias, items in (("plain", False, None), ("bias", True, None), ("restricted", False, allowed), ("bias_restricted", True, allowed)):
    rec = Base(X, verbose=False)
    rec.USER_factors, rec.ITEM_factors = U.copy(), V.copy()
    rec.use_bias = use_bias
    if use_bias:
        rec.USER_bias, rec.ITEM_bias, rec.GLOBAL_bias = bu.copy(), bi.copy(), mu
    ranked, scores = rec.recommend(users, cutoff=cutoff, remove_seen_flag=True, items_to_compute=items, return_scores=True)
    width = max(len(r) for r in ranked)
    table = np.full((len(users), cutoff), -1, np.int32)
    for r, lst in enumerate(ranked):
        table[r, :len(lst)] = lst
    out["ranked_" + tag] = table
    out["scores_" + tag] = np.asarray(scores, np.float64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "scoring.npz"), **out)
print("written tests/golden/scoring.npz:", {k: v.shape for k, v in out.items() if k.startswith(("ranked", "scores"))})
