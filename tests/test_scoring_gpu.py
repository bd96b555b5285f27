"""Device scoring + ranking against the host formulas of the reference
(BaseMatrixFactorizationRecommender._compute_item_score + BaseRecommender.recommend, re-provided in recommender_base)."""
import numpy as np
import pytest

from recsys2019_deeplearning_evaluation_amd import MI355XScorer, MatrixFactorization_BPR_MI355X
from recsys2019_deeplearning_evaluation_amd import recommender_base as RB
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm

pytestmark = pytest.mark.gpu


def _host_reference(X, U, V, users, bias=None, allowed=None, remove_seen=True):
    scores = U[users].astype(np.float64) @ V.T.astype(np.float64)
    if bias is not None:
        bu, bi, mu = bias
        scores = scores + bi + mu + bu[users][:, None]
    if allowed is not None:
        scores[:, ~allowed.astype(bool)] = -np.inf
    if remove_seen:
        for r, u in enumerate(users):
            scores[r, X.indices[X.indptr[u]:X.indptr[u + 1]]] = -np.inf
    return scores


def _check_ranking(ranked_row, score_row, cutoff, tol):
    got = ranked_row[ranked_row >= 0]
    finite = np.isfinite(score_row)
    k = min(cutoff, int(finite.sum()))
    assert len(got) == k and len(set(got.tolist())) == k
    if k == 0:
        return
    order = np.sort(score_row[finite])[::-1]
    t = order[k - 1]
    assert np.isfinite(score_row[got]).all()
    assert (score_row[got] >= t - tol).all()                       # nothing clearly below the k-th score
    assert np.isin(np.flatnonzero(score_row > t + tol), got).all()   # everything clearly above it is there
    assert (np.diff(score_row[got]) <= tol).all()                  # descending


@pytest.mark.parametrize("k", [1, 7, 64, 128, 200])
@pytest.mark.parametrize("use_bias", [False, True])
def test_scores_and_ranking_match_host(gpu, k, use_bias):
    X = named_urm("ml1m", "binary", scale=0.3)
    rng = np.random.default_rng(k)
    U = rng.normal(0, 0.3, (X.shape[0], k)).astype(np.float32); V = rng.normal(0, 0.3, (X.shape[1], k)).astype(np.float32)
    bias = (rng.normal(size=X.shape[0]).astype(np.float32), rng.normal(size=X.shape[1]).astype(np.float32), 0.7) if use_bias else None
    sc = MI355XScorer(U, V, X, *(bias if bias else ()))
    users = rng.choice(X.shape[0], 333, replace=False)
    ranked, scores = sc.recommend(users, 25, remove_seen=True, return_scores=True)
    want = _host_reference(X, U, V, users, bias)
    scale = np.abs(want[np.isfinite(want)]).max()
    assert (np.isfinite(scores) == np.isfinite(want)).all()
    assert np.abs(scores[np.isfinite(want)] - want[np.isfinite(want)]).max() < 1e-5 * scale
    for r in range(len(users)):
        _check_ranking(ranked[r], want[r], 25, 1e-5 * scale)
    st = sc.stats()
    assert st["algorithmic_flops"] == 2.0 * len(users) * X.shape[1] * k
    sc.close()


def test_filters_cutoffs_and_edge_cases(gpu):
    X = named_urm("ml1m", "binary", scale=0.2)
    rng = np.random.default_rng(0)
    U = rng.normal(size=(X.shape[0], 16)).astype(np.float32); V = rng.normal(size=(X.shape[1], 16)).astype(np.float32)
    sc = MI355XScorer(U, V, X)
    users = np.arange(40)
    allowed = np.zeros(X.shape[1], np.uint8); allowed[rng.choice(X.shape[1], 30, replace=False)] = 1
    for cutoff in (1, 30, 31, 200, X.shape[1]):
        for remove_seen in (False, True):
            ranked, _ = sc.recommend(users, cutoff, remove_seen=remove_seen, allowed_items=allowed)
            want = _host_reference(X, U, V, users, None, allowed, remove_seen)
            for r in range(len(users)):
                _check_ranking(ranked[r], want[r], min(cutoff, X.shape[1]), 1e-5 * 10)
    U2 = U * 2
    sc.update(U2, V)
    _, s2 = sc.recommend(users, 5, remove_seen=False, return_scores=True)
    np.testing.assert_allclose(s2, (U2[users].astype(np.float64) @ V.T.astype(np.float64)), rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        sc.recommend(np.array([X.shape[0]]), 5)
    sc.close()


def test_recommender_recommend_is_served_by_the_device_and_equals_the_host_path(gpu):
    X = named_urm("ml1m", "binary", scale=0.15)
    rec = MatrixFactorization_BPR_MI355X(X, verbose=False)
    rec.fit(epochs=5, batch_size=200, num_factors=24, learning_rate=0.05, sgd_mode="adagrad", random_seed=3)
    users = np.arange(60)
    dev_lists, dev_scores = rec.recommend(users, cutoff=10, return_scores=True)
    host_lists, host_scores = RB.BaseRecommender.recommend(rec, users, cutoff=10, return_scores=True)
    assert rec._scorer is not None and rec._scorer.stats()["n_units"] == 60
    fin = np.isfinite(host_scores)
    assert (np.isfinite(dev_scores) == fin).all()
    assert np.abs(dev_scores[fin] - host_scores[fin]).max() < 1e-5 * np.abs(host_scores[fin]).max()
    agree = np.mean([a == b for a, b in zip(dev_lists, host_lists)])
    assert agree > 0.9                                     # identical except where two scores are within float32 noise
    assert rec.recommend(3, cutoff=4) == dev_lists[3][:4]
    rec.set_items_to_ignore([0, 1, 2])
    assert not {0, 1, 2} & set(rec.recommend(5, cutoff=20, remove_custom_items_flag=True))


@pytest.mark.parametrize("user_based", [False, True])
def test_similarity_model_scoring_matches_host(gpu, user_based):
    """ItemKNN / UserKNN recommend(): URM[u] . W (resp. W[u] . URM) + filter + rank on the device == the host path."""
    from recsys2019_deeplearning_evaluation_amd import ItemKNNCFRecommender, UserKNNCFRecommender
    X = named_urm("ml1m", "real", scale=0.2)
    rec = (UserKNNCFRecommender if user_based else ItemKNNCFRecommender)(X, verbose=False)
    rec.fit(topK=30, shrink=3, similarity="cosine")
    users = np.arange(0, X.shape[0], 7)
    dev_lists, dev_scores = rec.recommend(users, cutoff=15, return_scores=True)
    host_lists, host_scores = RB.BaseRecommender.recommend(rec, users, cutoff=15, return_scores=True)
    assert rec._sp_scorer is not None
    fin = np.isfinite(host_scores)
    assert (np.isfinite(dev_scores) == fin).all()
    scale = np.abs(host_scores[fin]).max()
    assert np.abs(dev_scores[fin] - host_scores[fin]).max() < 1e-5 * scale
    for r in range(len(users)):
        _check_ranking(np.array(dev_lists[r] + [-1] * (15 - len(dev_lists[r]))), host_scores[r].astype(np.float64), 15, 1e-5 * scale)
    allowed_items = np.arange(0, X.shape[1], 3)
    only = rec.recommend(users[:5], cutoff=10, items_to_compute=allowed_items)
    assert all(set(l) <= set(allowed_items.tolist()) for l in only)
    assert rec.recommend(int(users[2]), cutoff=5) == dev_lists[2][:5]


def test_large_catalogues_and_full_rankings_stay_on_the_device(gpu):
    """Score rows that do not fit LDS (> ~32 k items) and the reference's default cutoff=None (rank ALL items): rows stay in
    HBM and are ordered by one segmented radix sort; same results as the host path, no NotImplementedError."""
    from recsys2019_deeplearning_evaluation_amd import MI355XSparseScorer
    from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
    import scipy.sparse as sps
    X = synthetic_urm(400, 40000, 30000, 5, 300, seed=2, values="binary")
    rng = np.random.default_rng(1)
    U = rng.normal(size=(X.shape[0], 32)).astype(np.float32); V = rng.normal(size=(X.shape[1], 32)).astype(np.float32)
    sc = MI355XScorer(U, V, X)
    users = rng.choice(X.shape[0], 50, replace=False)
    want = _host_reference(X, U, V, users)
    scale = np.abs(want[np.isfinite(want)]).max()
    for cutoff in (10, 5000, X.shape[1] - 1):                    # in-LDS selection impossible in all three cases (40 000 items)
        ranked, scores = sc.recommend(users, cutoff, remove_seen=True, return_scores=True)
        assert np.abs(scores[np.isfinite(want)] - want[np.isfinite(want)]).max() < 1e-5 * scale
        for r in range(0, len(users), 7):
            _check_ranking(ranked[r], want[r], cutoff, 1e-5 * scale)
    sc.close()
    # a small catalogue with a cutoff above the in-LDS selection limit (the Evaluator never asks for it, recommend(user) does)
    Xs = named_urm("ml1m", "binary", scale=0.3)
    Us = rng.normal(size=(Xs.shape[0], 8)).astype(np.float32); Vs = rng.normal(size=(Xs.shape[1], 8)).astype(np.float32)
    rec = RB.BaseMatrixFactorizationRecommender(Xs, verbose=False)
    rec.USER_factors, rec.ITEM_factors = Us, Vs
    from recsys2019_deeplearning_evaluation_amd import GpuScoringMixin
    dev = type("R", (GpuScoringMixin, RB.BaseMatrixFactorizationRecommender), {})(Xs, verbose=False)
    dev.USER_factors, dev.ITEM_factors = Us, Vs
    assert dev.recommend(5) == rec.recommend(5)                   # cutoff=None: the full ranking, identical lists
    # sparse (similarity-model) scorer on a wide catalogue
    W = sps.random(X.shape[1], X.shape[1], 2e-4, format="csr", random_state=5, dtype=np.float32)
    sp = MI355XSparseScorer(X, W, X)
    ranked, scores = sp.recommend(users, 20, remove_seen=True, return_scores=True)
    host = X[users].dot(W).toarray().astype(np.float64)
    for r, u in enumerate(users):
        host[r, X.indices[X.indptr[u]:X.indptr[u + 1]]] = -np.inf
    fin = np.isfinite(host)
    assert (np.isfinite(scores) == fin).all() and np.abs(scores[fin] - host[fin]).max() < 1e-5 * max(np.abs(host[fin]).max(), 1e-30)
    for r in range(0, len(users), 5):
        _check_ranking(ranked[r], host[r], 20, 1e-5 * max(np.abs(host[fin]).max(), 1e-30))
    sp.close()


def test_scorer_cache_follows_the_model_and_the_urm(gpu):
    """The lazily built device scorer must notice a new URM_train (seen items), replaced AND in-place edited factors, and a
    W_sparse whose predecessor's address has been recycled."""
    from recsys2019_deeplearning_evaluation_amd import GpuScoringMixin, GpuSimilarityScoringMixin
    import scipy.sparse as sps
    X = named_urm("ml1m", "binary", scale=0.15)
    rng = np.random.default_rng(3)
    MF = type("MF", (GpuScoringMixin, RB.BaseMatrixFactorizationRecommender), {})
    rec = MF(X, verbose=False)
    rec.USER_factors = rng.normal(size=(X.shape[0], 8)).astype(np.float32); rec.ITEM_factors = rng.normal(size=(X.shape[1], 8)).astype(np.float32)
    users = np.arange(30)
    host = lambda: RB.BaseRecommender.recommend(rec, users, cutoff=10)
    assert rec.recommend(users, cutoff=10) == host()
    rec.ITEM_factors *= -1.0                                      # in place: same object
    assert rec.recommend(users, cutoff=10) == host()
    rec.USER_factors = rec.USER_factors[::-1].copy()              # replaced
    assert rec.recommend(users, cutoff=10) == host()
    X2 = X.copy().tolil(); X2[:30, :] = 0; X2 = X2.tocsr()        # cold-start view: nothing is "seen" any more for these users
    rec.set_URM_train(X2)
    assert rec.recommend(users, cutoff=10) == host()
    KNN = type("KNN", (GpuSimilarityScoringMixin, RB.BaseItemSimilarityMatrixRecommender), {})
    knn = KNN(X, verbose=False)
    for seed in range(4):                                         # a fresh matrix each time; old ones are freed in between
        knn.W_sparse = sps.random(X.shape[1], X.shape[1], 0.03, format="csr", random_state=seed, dtype=np.float32)
        dev_lists, dev_scores = knn.recommend(users, cutoff=10, return_scores=True)
        _, host_scores = RB.BaseRecommender.recommend(knn, users, cutoff=10, return_scores=True)
        fin = np.isfinite(host_scores)
        assert np.abs(dev_scores[fin] - host_scores[fin]).max() < 1e-5 * max(1e-30, np.abs(host_scores[fin]).max())


@pytest.mark.parametrize("tag", ["plain", "bias", "restricted", "bias_restricted"])
def test_reference_generated_fixture(gpu, tag):
    """tests/golden/scoring.npz holds what the REFERENCE's own BaseMatrixFactorizationRecommender._compute_item_score and
    BaseRecommender.recommend return (tests/golden/make_scoring_fixture.py, run where /root/reference exists): scores after the
    seen-item filter and ranked lists, incl. a user who has seen nothing and one with fewer unseen items than the cut-off.
    float64 reference against the device's float32 MFMA scores: 1e-5 of the score scale; rankings equal wherever the reference's
    neighbouring scores are further apart than that."""
    import os
    import scipy.sparse as sps
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scoring.npz"))
    X = sps.csr_matrix((np.ones(len(z["indices"]), np.float32), z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    bias = (z["bu"], z["bi"], float(z["mu"])) if "bias" in tag else ()
    sc = MI355XScorer(z["U"], z["V"], X, *bias)
    allowed = None
    if "restricted" in tag:
        allowed = np.zeros(X.shape[1], np.uint8)
        allowed[z["allowed"]] = 1
    cutoff = int(z["cutoff"])
    ranked, scores = sc.recommend(z["users"], cutoff, remove_seen=True, allowed_items=allowed, return_scores=True)
    want_scores, want_ranked = z["scores_" + tag], z["ranked_" + tag]
    finite = np.isfinite(want_scores)
    assert (np.isfinite(scores) == finite).all()
    scale = np.abs(want_scores[finite]).max()
    tol = 1e-5 * scale
    assert np.abs(scores[finite] - want_scores[finite]).max() < tol
    for r in range(len(z["users"])):
        want = want_ranked[r][want_ranked[r] >= 0]
        got = ranked[r][ranked[r] >= 0]
        assert len(got) == len(want)
        _check_ranking(ranked[r], want_scores[r], cutoff, tol)
        # positions whose reference score is separated from both neighbours by more than the tolerance must hold the same item
        ws = want_scores[r][want]
        clear = np.ones(len(want), bool)
        clear[1:] &= (ws[:-1] - ws[1:]) > 2 * tol
        clear[:-1] &= (ws[:-1] - ws[1:]) > 2 * tol
        np.testing.assert_array_equal(got[clear], want[clear])
    sc.close()
