"""P3alpha / RP3beta on the device against fixtures produced by the reference's own
GraphBased/P3alphaRecommender.py and RP3betaRecommender.py (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from recsys2019_deeplearning_evaluation_amd import P3alphaRecommender, RP3betaRecommender
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
from _util import load_golden, unpack_csr

pytestmark = pytest.mark.gpu


def test_golden_fixture(gpu):
    z, cases = load_golden("graph_based")
    X = unpack_csr(z, "X")
    for n, case in enumerate(cases):
        rec = (P3alphaRecommender if case["cls"] == "P3" else RP3betaRecommender)(X.copy(), verbose=False)
        rec.fit(**case["kw"])
        want = z["W_%d" % n]
        got = rec.W_sparse.toarray()
        assert ((got != 0) == (want != 0)).all(), (n, case)
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), (n, case)


def test_ml1m_family_properties_and_scoring(gpu):
    X = named_urm("ml1m", "binary", scale=0.2)
    rec = RP3betaRecommender(X, verbose=False)
    rec.fit(topK=25, alpha=0.9, beta=0.5, normalize_similarity=True)
    W = rec.W_sparse
    assert W.shape == (X.shape[1], X.shape[1]) and (np.diff(W.tocsc().indptr) <= 25).all() and W.diagonal().max() == 0
    assert (W.data > 0).all()
    assert len(rec.recommend(0, cutoff=7)) == 7 and rec.similarity_stats["n_units"] == X.shape[1]


def test_implicit_mode_is_inside_the_reference_tie_class(gpu):
    """implicit=True makes every stored value 1, so the walk has exact ties and the reference's argsort picks arbitrarily
    among them: the per-row top-K is checked with the tie-aware comparator against the dense product Piu . Pui."""
    from _util import check_topk_against_dense
    X = named_urm("ml1m", "real", scale=0.08)
    rec = P3alphaRecommender(X.copy(), verbose=False)
    rec.fit(topK=10, alpha=0.8, min_rating=2, implicit=True, normalize_similarity=False)
    U = rec.URM_train
    assert (U.data == 1).all()
    Pui = U.multiply(1.0 / np.maximum(np.asarray(U.sum(axis=1)), 1e-30)).power(0.8)
    Xb = U.T.tocsr().astype(np.float64)
    Piu = Xb.multiply(1.0 / np.maximum(np.asarray(Xb.sum(axis=1)), 1e-30)).power(0.8)
    S = (Piu @ Pui).toarray()
    np.fill_diagonal(S, 0.0)
    W = rec.W_rowwise.tocsr()
    for i in range(S.shape[0]):
        row = W[i]
        order = np.lexsort((row.indices, -row.data))
        idx = -np.ones(10, np.int32); val = np.zeros(10, np.float32)
        idx[:len(order)] = row.indices[order]; val[:len(order)] = row.data[order]
        check_topk_against_dense(idx, val, S[i], 10, 1e-5)
