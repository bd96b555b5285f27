"""The CPU oracle (oracle/) reproduces the fixtures generated from the reference itself (tests/golden/)."""
import numpy as np
import pytest
import scipy.sparse as sps

from oracle import oracle as O
from _util import load_golden, unpack_csr


def test_similarity_oracle_matches_reference_outputs():
    z, cases = load_golden("similarity")
    X = unpack_csr(z, "X")
    for n, kw in enumerate(cases):
        dense = O.OracleSimilarity(X, topK=0, **kw).compute_similarity()
        np.testing.assert_array_equal(dense, z["dense_%d" % n])
        top = O.OracleSimilarity(X, topK=6, **kw).compute_similarity(exact_numpy_topk=True)
        np.testing.assert_array_equal(top.toarray(), z["top6_%d" % n])
    dense = O.OracleSimilarity(X, topK=0, shrink=2, row_weights=z["row_weights"]).compute_similarity()
    np.testing.assert_array_equal(dense, z["dense_rw"])


def test_similarity_oracle_c_topk_is_inside_reference_tie_class():
    z, cases = load_golden("similarity")
    X = unpack_csr(z, "X")
    for n, kw in enumerate(cases):
        top = O.OracleSimilarity(X, topK=6, **kw).compute_similarity().toarray()
        ref = z["top6_%d" % n]
        # same multiset of values per column; the chosen indices may differ only among equal values
        for c in range(ref.shape[1]):
            np.testing.assert_allclose(np.sort(top[:, c]), np.sort(ref[:, c]), rtol=0, atol=0)


def test_topk_rule_python_reference_vs_cython_reference():
    """Where the reference's two implementations disagree, and which one this package follows.

    Fixture: Compute_Similarity_Python and Compute_Similarity_Cython outputs on tie-free data whose adjusted / pearson
    similarities are partly negative.  Statements checked, column by column:
      1. the oracle's full-column rule (zeros compete, then are dropped) == Compute_Similarity_Python, indices AND values;
      2. the Cython class equals it whenever at least min(K, touched) touched cells are positive (always, on non-negative data);
      3. otherwise the Cython class has picked zero-padding entries of its partition array and read stale neighbour ids
         (Compute_Similarity_Cython.pyx:523-545): its column then contains duplicates summed by the COO->CSR conversion or
         extra negative cells -- it is NOT the K largest cells of the column.  The device follows rule 1."""
    z, cases = load_golden("similarity_topk_rules")
    X = unpack_csr(z, "X")
    n = X.shape[1]
    differing = 0
    for k, kw in enumerate(cases):
        py, cy, dense = z["python_%d" % k], z["cython_%d" % k], z["dense_%d" % k]
        mine = O.OracleSimilarity(X, **kw).compute_similarity_full_column_rule().toarray()
        assert ((mine != 0) == (py != 0)).all(), kw                                   # 1: same neighbours ...
        atol = 1e-6 * np.abs(py).max()      # (the Python class works in float32, the Cython class and the oracle in float64)
        np.testing.assert_allclose(mine, py, rtol=1e-6, atol=atol)                    #    ... same values
        for c in range(n):
            col = dense[:, c]
            n_pos, n_touched = int((col > 0).sum()), int((col != 0).sum())
            same = np.allclose(py[:, c], cy[:, c], rtol=1e-6, atol=atol)
            if n_pos >= min(kw["topK"], n_touched):
                assert same, (kw, c)                                                  # 2
            elif not same:
                differing += 1
                support = np.flatnonzero(cy[:, c])
                doubled = ~np.isclose(cy[support, c], dense[support, c], rtol=1e-5)   # duplicates summed by scipy
                assert doubled.any() or (cy[:, c] < 0).any(), (kw, c)                 # 3
        if kw["similarity"] in ("cosine", "asymmetric"):
            np.testing.assert_allclose(py, cy, rtol=1e-6, atol=atol)                  # non-negative data: identical
    assert differing >= 10            # the fixture really exercises the difference


def test_mf_oracle_is_bit_exact_with_reference_outputs():
    z, cases = load_golden("matrix_factorization")
    mats = {"Xb": unpack_csr(z, "Xb"), "Xr": unpack_csr(z, "Xr")}
    for n, case in enumerate(cases):
        m = O.OracleMF(mats[case["matrix"]], **case["kw"])
        for _ in range(case["epochs"]):
            m.epochIteration_Cython()
        np.testing.assert_array_equal(m.get_USER_factors(), z["U_%d" % n])
        np.testing.assert_array_equal(m.get_ITEM_factors(), z["V_%d" % n])
        if case["kw"].get("use_bias"):
            np.testing.assert_array_equal(m.get_USER_bias(), z["bu_%d" % n])
            np.testing.assert_array_equal(m.get_ITEM_bias(), z["bi_%d" % n])
            np.testing.assert_array_equal(m.get_GLOBAL_bias(), z["mu_%d" % n])


def test_mf_oracle_replay_equals_native_sampling():
    z, cases = load_golden("matrix_factorization")
    mats = {"Xb": unpack_csr(z, "Xb"), "Xr": unpack_csr(z, "Xr")}
    for n, case in enumerate(cases[:4]):
        X = mats[case["matrix"]]
        a = O.OracleMF(X, **case["kw"])
        a.record_samples(10 ** 6)
        for _ in range(case["epochs"]):
            a.epochIteration_Cython()
        u, i, j, r = a.recorded()
        b = O.OracleMF(X, **case["kw"])
        b.replay(u, i, j, r)
        np.testing.assert_array_equal(a.get_USER_factors(), b.get_USER_factors())
        np.testing.assert_array_equal(a.get_ITEM_factors(), b.get_ITEM_factors())
        np.testing.assert_array_equal(a.get_USER_factors(), z["U_%d" % n])


def test_slim_oracle_is_bit_exact_with_reference_outputs():
    z, cases = load_golden("slim_bpr")
    X = unpack_csr(z, "X")
    for n, case in enumerate(cases):
        e = O.OracleSLIM(X, topK=False, final_model_sparse_weights=False, **case["kw"])
        for _ in range(case["epochs"]):
            e.epochIteration_Cython()
        np.testing.assert_array_equal(e.get_S_dense(), z["S_%d" % n])


def test_slim_sparse_store_oracle_is_bit_exact_with_reference_outputs():
    """train_with_sparse_weights=True: the fixtures hold the reference's get_S() (csr of the tree) in the middle and at the end
    of training -- rebalance points at the fifths of the epoch, the selection inside get_S, ties, topK=False."""
    z, cases = load_golden("slim_bpr_sparse")
    for n, case in enumerate(cases):
        X = unpack_csr(z, "X%d" % case["n_users"])
        e = O.OracleSLIM(X, **case["kw"])
        for m, epochs in enumerate(case["epochs"]):
            for _ in range(epochs):
                e.epochIteration_Cython()
            S = e.get_S()
            np.testing.assert_array_equal(S.indptr, z["indptr_%d_%d" % (n, m)])
            np.testing.assert_array_equal(S.indices, z["indices_%d_%d" % (n, m)])
            np.testing.assert_array_equal(S.data, z["data_%d_%d" % (n, m)])


def test_ials_oracle_matches_reference_outputs():
    z, cases = load_golden("ials")
    X = unpack_csr(z, "X")
    for n, case in enumerate(cases):
        kw = case["kw"]
        Cm = O.oracle_ials_confidence(X, kw["confidence_scaling"], kw["alpha"], kw.get("epsilon", 1.0))
        Cc = sps.csc_matrix(Cm)
        V = z["V0_%d" % n].copy()
        U = np.zeros((X.shape[0], kw["num_factors"]))
        for _ in range(case["epochs"]):
            O.oracle_ials_epoch(Cm, Cc, U, V, kw["reg"])
        np.testing.assert_allclose(U, z["U_%d" % n], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(V, z["V_%d" % n], rtol=1e-10, atol=1e-12)


def test_euclidean_oracle_is_bit_exact_with_reference_outputs():
    """Same NumPy float32 operations in the same order: the restatement reproduces the reference's dense columns and
    its top-K csr_matrix exactly (integer ratings and jittered ones alike)."""
    z, cases = load_golden("euclidean")
    X = unpack_csr(z, "X")
    for n, kw in enumerate(cases):
        orc = O.OracleSimilarityEuclidean(X, topK=X.shape[1], **kw)
        assert np.array_equal(orc.dense().astype(np.float32), z["dense_%d" % n]), kw
        top = O.OracleSimilarityEuclidean(X, topK=6, **kw).compute_similarity().toarray()
        assert np.array_equal(top, z["top_%d" % n]), kw
    Xj = unpack_csr(z, "Xj")
    assert np.array_equal(O.OracleSimilarityEuclidean(Xj, topK=Xj.shape[1], **cases[0]).dense().astype(np.float32), z["densej_0"])
    with pytest.raises(ValueError):
        O.OracleSimilarityEuclidean(X, similarity_from_distance_mode="sqrt")


def test_euclidean_oracle_with_row_weights_is_bit_exact_with_reference_outputs():
    """row_weights (Compute_Similarity_Euclidean.py:62-72, :153, :174-175) on a square matrix, float32 and float64 weights; any other
    shape fails in NumPy's broadcasting exactly where the reference does."""
    z, cases = load_golden("euclidean_row_weights")
    X = unpack_csr(z, "X")
    assert X.shape[0] == X.shape[1]
    for n, case in enumerate(cases):
        kw = dict(case["kw"], row_weights=z[case["weights"]])
        assert np.array_equal(O.OracleSimilarityEuclidean(X, topK=X.shape[1], **kw).dense().astype(np.float32), z["dense_%d" % n]), case
        assert np.array_equal(O.OracleSimilarityEuclidean(X, topK=5, **kw).compute_similarity().toarray(), z["top_%d" % n]), case
    with pytest.raises(ValueError):
        O.OracleSimilarityEuclidean(X, row_weights=np.ones(X.shape[0] + 1))
    with pytest.raises(ValueError):                  # 48 column distances times 40 row weights
        O.OracleSimilarityEuclidean(X[:40], row_weights=np.ones(40)).dense()
