"""The plugin surface against the reference's OWN harness (runs where /root/reference exists; CPU only).

  * `EvaluatorHoldout.evaluateRecommender` (Base/Evaluation/Evaluator.py:225, 382, 436) on this package's recommender surface
    holding a given model gives the metrics the reference's own recommender classes give for the same model;
  * the same through `reference_binding.bind()`: the package's recommenders rebuilt as SUBCLASSES of the reference's Base
    classes are instances of those classes and evaluate identically;
  * models saved by this package load with the reference's `DataIO` (Base/DataIO.py:186) and `load_model`, and vice versa.
No kernel runs here: the models are given, the test is about the boundary.
"""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sps

from oracle import ref_loader
from recsys2019_deeplearning_evaluation_amd import recommender_base as RB
from recsys2019_deeplearning_evaluation_amd.reference_binding import bind
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm

pytestmark = pytest.mark.skipif(not ref_loader.reference_tree_available(), reason="needs the reference tree (/root/reference)")


def _ref(dotted, name):
    return ref_loader.load_python_reference(dotted, name)


def _quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture(scope="module")
def split():
    X = named_urm("ml1m", "real", scale=0.12)
    rng = np.random.default_rng(0)
    mask = rng.random(X.nnz) < 0.8
    coo = X.tocoo()
    train = sps.csr_matrix((coo.data[mask], (coo.row[mask], coo.col[mask])), shape=X.shape)
    test = sps.csr_matrix((coo.data[~mask], (coo.row[~mask], coo.col[~mask])), shape=X.shape)
    return train, test


def _metrics(evaluator, rec):
    results, _ = _quiet(evaluator.evaluateRecommender, rec)
    return results


def _assert_same(a, b):
    assert a.keys() == b.keys()
    for cutoff in a:
        for metric, value in a[cutoff].items():
            assert np.isclose(float(value), float(b[cutoff][metric]), rtol=1e-9, atol=1e-12), (cutoff, metric, value, b[cutoff][metric])


def test_evaluator_holdout_on_factor_models(split):
    train, test = split
    EvaluatorHoldout = _ref("Base.Evaluation.Evaluator", "EvaluatorHoldout")
    RefMF = _ref("Base.BaseMatrixFactorizationRecommender", "BaseMatrixFactorizationRecommender")
    evaluator = _quiet(EvaluatorHoldout, test, [5, 10])
    rng = np.random.default_rng(1)
    U = rng.normal(0, 0.1, (train.shape[0], 12)); V = rng.normal(0, 0.1, (train.shape[1], 12))
    bu = rng.normal(0, 0.1, train.shape[0]); bi = rng.normal(0, 0.1, train.shape[1])
    for use_bias in (False, True):
        recs = []
        for cls in (RefMF, RB.BaseMatrixFactorizationRecommender):
            rec = _quiet(cls, train)
            rec.USER_factors, rec.ITEM_factors, rec.use_bias = U, V, use_bias
            if use_bias:
                rec.USER_bias, rec.ITEM_bias, rec.GLOBAL_bias = bu, bi, 0.3
            recs.append(rec)
        _assert_same(_metrics(evaluator, recs[0]), _metrics(evaluator, recs[1]))
    # ignored items / users flow through the same calls (Evaluator.py:183-190, BaseRecommender.py:82-97)
    ev2 = _quiet(EvaluatorHoldout, test, [10], ignore_items=np.arange(0, train.shape[1], 7), ignore_users=np.arange(0, 50))
    _assert_same(_metrics(ev2, recs[0]), _metrics(ev2, recs[1]))


def test_evaluator_holdout_on_similarity_models(split):
    train, test = split
    EvaluatorHoldout = _ref("Base.Evaluation.Evaluator", "EvaluatorHoldout")
    RefItem = _ref("Base.BaseSimilarityMatrixRecommender", "BaseItemSimilarityMatrixRecommender")
    RefUser = _ref("Base.BaseSimilarityMatrixRecommender", "BaseUserSimilarityMatrixRecommender")
    evaluator = _quiet(EvaluatorHoldout, test, [10])
    Wi = sps.random(train.shape[1], train.shape[1], 0.05, format="csr", random_state=3, dtype=np.float32)
    Wu = sps.random(train.shape[0], train.shape[0], 0.02, format="csr", random_state=4, dtype=np.float32)
    for ref_cls, own_cls, W in ((RefItem, RB.BaseItemSimilarityMatrixRecommender, Wi), (RefUser, RB.BaseUserSimilarityMatrixRecommender, Wu)):
        a, b = _quiet(ref_cls, train), _quiet(own_cls, train)
        a.W_sparse = b.W_sparse = W
        _assert_same(_metrics(evaluator, a), _metrics(evaluator, b))


def test_recommenders_rebuilt_on_the_reference_bases(split):
    train, test = split
    EvaluatorHoldout = _ref("Base.Evaluation.Evaluator", "EvaluatorHoldout")
    RefMF = _ref("Base.BaseMatrixFactorizationRecommender", "BaseMatrixFactorizationRecommender")
    RefItem = _ref("Base.BaseSimilarityMatrixRecommender", "BaseItemSimilarityMatrixRecommender")
    RefUser = _ref("Base.BaseSimilarityMatrixRecommender", "BaseUserSimilarityMatrixRecommender")
    RefES = _ref("Base.Incremental_Training_Early_Stopping", "Incremental_Training_Early_Stopping")
    RefBase = _ref("Base.BaseRecommender", "BaseRecommender")
    R = bind(RefMF, RefItem, RefUser, RefES, device_scoring=False)      # recommend() = the reference's own host implementation
    evaluator = _quiet(EvaluatorHoldout, test, [10])
    rng = np.random.default_rng(2)
    bpr = _quiet(R.MatrixFactorization_BPR_MI355X, train)
    assert isinstance(bpr, RefMF) and isinstance(bpr, RefES) and isinstance(bpr, RefBase)
    assert type(bpr).fit.__qualname__.startswith("_BPRLogic")            # the device fit(), the reference's everything else
    assert type(bpr).recommend is RefBase.recommend and type(bpr).save_model is RefMF.save_model
    bpr.USER_factors = rng.normal(0, 0.1, (train.shape[0], 8)); bpr.ITEM_factors = rng.normal(0, 0.1, (train.shape[1], 8))
    plain = _quiet(RefMF, train)
    plain.USER_factors, plain.ITEM_factors = bpr.USER_factors, bpr.ITEM_factors
    _assert_same(_metrics(evaluator, bpr), _metrics(evaluator, plain))
    knn = _quiet(R.ItemKNNCFRecommender, train)
    assert isinstance(knn, RefItem) and knn.RECOMMENDER_NAME == "ItemKNNCFRecommender"
    slim = _quiet(R.SLIM_BPR_MI355X, train)
    assert isinstance(slim, RefItem) and isinstance(slim, RefES)
    for name in ("MatrixFactorization_FunkSVD_MI355X", "MatrixFactorization_AsySVD_MI355X", "IALSRecommender", "UserKNNCFRecommender",
                 "P3alphaRecommender", "RP3betaRecommender"):
        assert isinstance(_quiet(getattr(R, name), train), RefBase)
    # with device scoring the package's recommend() sits in front of the reference's (needs the GPU at call time, not here)
    Rd = bind(RefMF, RefItem, RefUser, RefES)
    from recsys2019_deeplearning_evaluation_amd.scoring import GpuScoringMixin
    assert Rd.IALSRecommender.recommend is GpuScoringMixin.recommend and issubclass(Rd.IALSRecommender, RefMF)


def test_saved_models_are_interchangeable_with_dataio(split, tmp_path):
    train, _ = split
    DataIO = _ref("Base.DataIO", "DataIO")
    RefMF = _ref("Base.BaseMatrixFactorizationRecommender", "BaseMatrixFactorizationRecommender")
    RefItem = _ref("Base.BaseSimilarityMatrixRecommender", "BaseItemSimilarityMatrixRecommender")
    folder = str(tmp_path) + os.sep
    rng = np.random.default_rng(5)
    own = _quiet(RB.BaseMatrixFactorizationRecommender, train)
    own.USER_factors = rng.normal(size=(train.shape[0], 6)).astype(np.float32); own.ITEM_factors = rng.normal(size=(train.shape[1], 6)).astype(np.float32)
    own.use_bias = True
    own.USER_bias = rng.normal(size=train.shape[0]); own.ITEM_bias = rng.normal(size=train.shape[1]); own.GLOBAL_bias = np.array(0.25)
    _quiet(own.save_model, folder, "own_mf")
    loaded = _quiet(DataIO(folder_path=folder).load_data, "own_mf")                         # package -> reference DataIO
    assert set(loaded) == {"USER_factors", "ITEM_factors", "use_bias", "ITEM_bias", "USER_bias", "GLOBAL_bias"}
    np.testing.assert_array_equal(loaded["USER_factors"], own.USER_factors)
    assert loaded["use_bias"] is True and float(loaded["GLOBAL_bias"]) == 0.25
    ref = _quiet(RefMF, train)
    _quiet(ref.load_model, folder, "own_mf")                                                  # ... and the reference's load_model
    np.testing.assert_array_equal(ref.ITEM_bias, own.ITEM_bias)
    ref.USER_factors = ref.USER_factors * 2
    _quiet(ref.save_model, folder, "ref_mf")                                                  # reference -> package
    back = _quiet(RB.BaseMatrixFactorizationRecommender, train)
    _quiet(back.load_model, folder, "ref_mf")
    np.testing.assert_array_equal(back.USER_factors, own.USER_factors * 2)
    assert back.use_bias is True
    W = sps.random(train.shape[1], train.shape[1], 0.05, format="csr", random_state=6, dtype=np.float32)
    a = _quiet(RB.BaseItemSimilarityMatrixRecommender, train); a.W_sparse = W
    _quiet(a.save_model, folder, "own_knn")
    b = _quiet(RefItem, train)
    _quiet(b.load_model, folder, "own_knn")
    assert (b.W_sparse != W).nnz == 0


@pytest.mark.parametrize("tag", ["plain", "bias", "restricted", "bias_restricted"])
def test_host_scoring_path_equals_the_reference_generated_fixture(tag):
    """tests/golden/scoring.npz was written by the reference's own BaseMatrixFactorizationRecommender (make_scoring_fixture.py);
    the package's host-side `_compute_item_score` + `recommend` (what a recommender without a device scorer runs, and what the GPU
    tests' rankings are compared with) must reproduce it exactly: same scores, same ranked lists."""
    import scipy.sparse as sps
    from recsys2019_deeplearning_evaluation_amd import recommender_base as RB
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scoring.npz"))
    X = sps.csr_matrix((np.ones(len(z["indices"]), np.float32), z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    rec = RB.BaseMatrixFactorizationRecommender(X, verbose=False)
    rec.USER_factors, rec.ITEM_factors = z["U"].copy(), z["V"].copy()
    rec.use_bias = "bias" in tag
    if rec.use_bias:
        rec.USER_bias, rec.ITEM_bias, rec.GLOBAL_bias = z["bu"].copy(), z["bi"].copy(), float(z["mu"])
    items = z["allowed"] if "restricted" in tag else None
    ranked, scores = RB.BaseRecommender.recommend(rec, z["users"], cutoff=int(z["cutoff"]), remove_seen_flag=True, items_to_compute=items,
                                                  return_scores=True)
    np.testing.assert_array_equal(np.asarray(scores, np.float64), z["scores_" + tag])
    for r, lst in enumerate(ranked):
        want = z["ranked_" + tag][r]
        assert list(lst) == want[want >= 0].tolist()
