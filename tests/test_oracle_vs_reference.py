"""Where the compiled reference (oracle/_ref, built by oracle/build_ref.py) is available, the oracle is compared
with it directly on fresh seeded inputs (larger than the committed fixtures)."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sps

from oracle import oracle as O
from oracle import ref_loader
from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm


def _quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture(scope="module")
def ref():
    mods = {k: ref_loader.load(k) for k in ("mf", "slim", "sim")}
    if any(v is None for v in mods.values()):
        pytest.skip("oracle/_ref not built (run python oracle/build_ref.py where /root/reference exists)")
    return mods


@pytest.mark.parametrize("mode", ["sgd", "adagrad", "rmsprop", "adam"])
def test_bpr_and_funk_bit_exact(ref, mode):
    Xb = synthetic_urm(400, 150, 9000, 5, 80, seed=1, values="binary")
    Xr = synthetic_urm(400, 150, 9000, 5, 80, seed=2, values="real")
    kw = dict(n_factors=16, algorithm_name="MF_BPR", batch_size=32, random_seed=9, sgd_mode=mode, learning_rate=0.05,
              user_reg=0.01, positive_reg=0.02, negative_reg=0.03)
    # NB the reference and the oracle share glibc's global rand() state: run them one after the other
    a = ref["mf"](Xb, **kw)
    for _ in range(3):
        _quiet(a.epochIteration_Cython)
    b = O.OracleMF(Xb, **kw)
    for _ in range(3):
        b.epochIteration_Cython()
    np.testing.assert_array_equal(a.get_USER_factors(), b.get_USER_factors())
    np.testing.assert_array_equal(a.get_ITEM_factors(), b.get_ITEM_factors())
    kw = dict(n_factors=16, algorithm_name="FUNK_SVD", batch_size=64, random_seed=10, sgd_mode=mode, learning_rate=0.01,
              user_reg=0.01, item_reg=0.3, bias_reg=0.02, use_bias=True, negative_interactions_quota=0.4)
    a = ref["mf"](Xr, **kw)
    _quiet(a.epochIteration_Cython)
    b = O.OracleMF(Xr, **kw)
    b.epochIteration_Cython()
    np.testing.assert_array_equal(a.get_USER_factors(), b.get_USER_factors())
    np.testing.assert_array_equal(a.get_ITEM_factors(), b.get_ITEM_factors())
    np.testing.assert_array_equal(a.get_ITEM_bias(), b.get_ITEM_bias())
    np.testing.assert_array_equal(a.get_GLOBAL_bias(), b.get_GLOBAL_bias())


@pytest.mark.parametrize("symmetric", [False, True])
def test_slim_bit_exact(ref, symmetric):
    X = synthetic_urm(300, 90, 6000, 5, 60, seed=3, values="binary")
    for mode in ["sgd", "adagrad", "adam"]:
        kw = dict(symmetric=symmetric, random_seed=4, sgd_mode=mode, learning_rate=0.05, li_reg=0.01, lj_reg=0.02)
        a = ref["slim"](X, topK=False, final_model_sparse_weights=False, **kw)
        for _ in range(2):
            _quiet(a.epochIteration_Cython)
        b = O.OracleSLIM(X, topK=False, final_model_sparse_weights=False, **kw)
        for _ in range(2):
            b.epochIteration_Cython()
        Sa = _quiet(a.get_S)
        Sa = Sa.toarray() if sps.issparse(Sa) else np.asarray(Sa)
        np.testing.assert_array_equal(Sa, b.get_S_dense())
        _quiet(a._dealloc)


@pytest.mark.parametrize("n_users", [299, 300, 303])
def test_slim_sparse_store_bit_exact(ref, n_users):
    """Sparse-tree store: 300 steps per epoch (4 rebalance points), 301 and 304 (5: the rule divides as C integers)."""
    X = synthetic_urm(n_users, 40, 3000, 3, 30, seed=3, values="binary")
    for topK, mode, regs in [(5, "sgd", (0.0, 0.0)), (7, "adam", (0.01, 0.02)), (False, "adagrad", (0.01, 0.0)), (3, "rmsprop", (0.0, 0.03)),
                             (60, "sgd", (0.0, 0.0))]:
        kw = dict(symmetric=True, random_seed=4, sgd_mode=mode, learning_rate=0.05, li_reg=regs[0], lj_reg=regs[1], topK=topK,
                  train_with_sparse_weights=True)
        a = ref["slim"](X, **kw)
        for _ in range(2):
            _quiet(a.epochIteration_Cython)
        Sa = [_quiet(a.get_S)]
        _quiet(a.epochIteration_Cython)                  # get_S pruned the model: training goes on from the pruned one
        Sa.append(_quiet(a.get_S))
        _quiet(a._dealloc)
        b = O.OracleSLIM(X, **kw)
        for _ in range(2):
            b.epochIteration_Cython()
        Sb = [b.get_S()]
        b.epochIteration_Cython()
        Sb.append(b.get_S())
        for x, y in zip(Sa, Sb):
            np.testing.assert_array_equal(x.indptr, y.indptr)
            np.testing.assert_array_equal(x.indices, y.indices)
            np.testing.assert_array_equal(x.data, y.data)


@pytest.mark.parametrize("similarity", ["cosine", "adjusted", "asymmetric", "pearson", "jaccard", "dice", "tversky"])
def test_similarity_bit_exact(ref, similarity):
    X = synthetic_urm(500, 200, 12000, 5, 100, seed=5, values="real")
    for shrink, normalize in [(0, True), (10, True), (4, False)]:
        kw = dict(shrink=shrink, normalize=normalize, similarity=similarity, asymmetric_alpha=0.4, tversky_alpha=0.6, tversky_beta=1.4)
        Wa = _quiet(lambda: ref["sim"](X, topK=0, **kw).compute_similarity())
        Wb = O.OracleSimilarity(X, topK=0, **kw).compute_similarity()
        np.testing.assert_array_equal(np.asarray(Wa), Wb)
        Ta = _quiet(lambda: ref["sim"](X, topK=15, **kw).compute_similarity())
        Tb = O.OracleSimilarity(X, topK=15, **kw).compute_similarity(exact_numpy_topk=True)
        assert abs(Ta - Tb).max() == 0


def test_column_range_matches_reference(ref):
    X = synthetic_urm(300, 120, 6000, 5, 60, seed=6, values="real")
    for s, e in [(None, None), (10, 50), (0, 30), (100, 500), (40, 20)]:
        Wa = _quiet(lambda: ref["sim"](X, topK=8).compute_similarity(start_col=s, end_col=e))
        Wb = O.OracleSimilarity(X, topK=8).compute_similarity(start_col=s, end_col=e, exact_numpy_topk=True)
        assert abs(Wa - Wb).max() == 0


def test_shim_alone_is_enough_to_import_the_compiled_reference(monkeypatch, tmp_path):
    """The GPU box has no /root/reference: the compiled modules must import with oracle/ref_shim only."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import ref_loader\n"
            "assert not ref_loader.reference_tree_available()\n"
            "assert all(ref_loader.load(k) is not None for k in ('mf','slim','sim'))\n" % root)
    env = dict(os.environ, RECSYS_REFERENCE_ROOT=str(tmp_path / "nowhere"))
    subprocess.run([sys.executable, "-c", code], check=True, env=env)


def test_baseline_config_1_itemknn_cosine_ml1m_shape(ref):
    """BASELINE.json configs[0]: ItemKNNCF cosine top-k=100 on an ML-1M-shaped URM through the reference Cython
    Compute_Similarity on the CPU (plumbing, no GPU): reference == oracle, entry for entry."""
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    X = named_urm("ml1m", "real")
    Wa = _quiet(lambda: ref["sim"](X, topK=100, shrink=0, normalize=True, similarity="cosine").compute_similarity())
    orc = O.OracleSimilarity(X, topK=100, shrink=0, normalize=True, similarity="cosine")
    Wb = orc.compute_similarity()                       # C top-K (tie-free input => same as NumPy's)
    assert Wa.shape == (3706, 3706) and Wa.nnz == Wb.nnz == 3706 * 100
    assert abs(Wa - Wb).max() == 0


@pytest.mark.parametrize("mode", ["lin", "log", "exp"])
def test_euclidean_bit_exact(ref, mode):
    EUC = ref_loader.load_python_reference("Base.Similarity.Compute_Similarity_Euclidean", "Compute_Similarity_Euclidean")
    X = synthetic_urm(150, 90, 2500, seed=5, values="real")
    for kw in (dict(normalize=False, normalize_avg_row=False, shrink=0), dict(normalize=True, normalize_avg_row=True, shrink=3)):
        want = _quiet(lambda: EUC(X, topK=10, similarity_from_distance_mode=mode, **kw).compute_similarity()).toarray()
        got = O.OracleSimilarityEuclidean(X, topK=10, similarity_from_distance_mode=mode, **kw).compute_similarity().toarray()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_euclidean_with_row_weights_bit_exact(ref, dtype):
    """Square inputs only: the reference multiplies the distances to the columns by the weights of the rows (:174-175)."""
    EUC = ref_loader.load_python_reference("Base.Similarity.Compute_Similarity_Euclidean", "Compute_Similarity_Euclidean")
    X = synthetic_urm(120, 120, 2600, seed=8, values="real")
    w = np.random.default_rng(8).uniform(0.1, 3.0, 120).astype(dtype)
    for kw in (dict(normalize=False, normalize_avg_row=False, shrink=0, similarity_from_distance_mode="lin"),
               dict(normalize=True, normalize_avg_row=True, shrink=2, similarity_from_distance_mode="log")):
        want = _quiet(lambda: EUC(X, topK=12, row_weights=w, **kw).compute_similarity()).toarray()
        got = O.OracleSimilarityEuclidean(X, topK=12, row_weights=w, **kw).compute_similarity().toarray()
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        _quiet(lambda: EUC(X[:100], topK=12, row_weights=w[:100]).compute_similarity())
    with pytest.raises(ValueError):
        O.OracleSimilarityEuclidean(X[:100], topK=12, row_weights=w[:100]).compute_similarity()


def test_package_similarity_matrix_topk_equals_the_reference_function():
    ref_topk = ref_loader.load_python_reference("Base.Recommender_utils", "similarityMatrixTopK")
    if ref_topk is None:
        pytest.skip("reference tree not available")
    from recsys2019_deeplearning_evaluation_amd.recommender_base import similarityMatrixTopK
    rng = np.random.default_rng(2)
    A = sps.random(70, 70, 0.4, random_state=np.random.RandomState(2), format="csr", dtype=np.float32)
    A.data = rng.standard_normal(A.nnz).astype(np.float32)
    for k in (1, 6, 70):
        want = _quiet(lambda: ref_topk(A.copy(), k=k, verbose=False)).toarray()
        assert np.array_equal(similarityMatrixTopK(A, k).toarray(), want)
        want_d = _quiet(lambda: ref_topk(A.toarray(), k=k, verbose=False))
        want_d = want_d.toarray() if sps.issparse(want_d) else np.asarray(want_d)
        got_d = similarityMatrixTopK(A.toarray(), k).toarray()
        assert np.array_equal(got_d, want_d)        # same rule on dense input: non-zero cells only (Recommender_utils.py:104-108)
