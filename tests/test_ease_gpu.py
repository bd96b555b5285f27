"""EASE_R with the Gram step on the device against fixtures produced by the reference's
EASE_R_Recommender.  The inverse is the reference's own float32 np.linalg.inv call on both sides, so the bar is set by
its conditioning: 1e-4 of the largest coefficient."""
import numpy as np
import pytest
import scipy.sparse as sps

from recsys2019_deeplearning_evaluation_amd import EASE_R_Recommender
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
from _util import load_golden, unpack_csr

pytestmark = pytest.mark.gpu


def test_golden_fixture(gpu):
    z, cases = load_golden("ease_r")
    X = unpack_csr(z, "X")
    for n, kw in enumerate(cases):
        rec = EASE_R_Recommender(X.copy(), verbose=False)
        rec.fit(verbose=False, **kw)
        want = z["W_%d" % n]
        got = rec.W_sparse.toarray() if sps.issparse(rec.W_sparse) else rec.W_sparse
        assert got.shape == want.shape and (np.diag(got) == 0).all()
        if kw["topK"] is not None:
            assert ((got != 0) == (want != 0)).all(), kw
        assert np.abs(got - want).max() < 1e-4 * np.abs(want).max(), (kw, np.abs(got - want).max())


def test_gram_is_exact_and_recommend_runs(gpu):
    X = named_urm("ml1m", "binary", scale=0.1)
    rec = EASE_R_Recommender(X.copy(), verbose=False)
    rec.fit(topK=None, l2_norm=100.0, verbose=False)
    G = (X.T @ X).toarray().astype(np.float64)
    G[np.diag_indices_from(G)] = np.diff(X.tocsc().indptr) + 100.0
    P = np.linalg.inv(G)
    B = P / (-np.diag(P))
    np.fill_diagonal(B, 0.0)
    assert np.abs(rec.W_sparse - B).max() < 1e-4 * np.abs(B).max()
    users = np.arange(0, X.shape[0], 7)
    items = rec.recommend(users, cutoff=5, remove_seen_flag=True)
    assert len(items) == len(users) and all(len(r) == 5 for r in items)
    seen = [set(X.indices[X.indptr[u]:X.indptr[u + 1]]) for u in users]
    assert all(not (set(r) & s) for r, s in zip(items, seen))
