"""Parity of the HIP Compute_Similarity path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerance: 1e-5 relative on similarity values (north_star); top-K indices exact up to the reference's own tie
class (values closer than the tolerance), see _util.check_topk_against_dense.
"""
import numpy as np
import pytest
import scipy.sparse as sps

from oracle import oracle as O
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity, Compute_Similarity_MI355X, ItemKNNCFRecommender
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm, synthetic_urm
from _util import check_topk_against_dense, load_golden, rel_err, unpack_csr

pytestmark = pytest.mark.gpu
RTOL = 1e-5
ALL_SIMS = ["cosine", "adjusted", "asymmetric", "pearson", "jaccard", "tanimoto", "dice", "tversky"]


def _check_build(X, topK, rtol=RTOL, **kw):
    dev = Compute_Similarity_MI355X(X, topK=topK, **kw)
    idx, val, s = dev.compute_slabs()
    orc = O.OracleSimilarity(X, topK=0, **kw)
    for c in range(X.shape[1]):
        check_topk_against_dense(idx[c], val[c], orc.column(c)[0], topK, rtol)
    dev.close()
    return idx, val


@pytest.mark.parametrize("similarity", ALL_SIMS)
def test_golden_fixture_dense_and_topk(gpu, similarity):
    z, cases = load_golden("similarity")
    X = unpack_csr(z, "X")
    for n, kw in enumerate(cases):
        if kw["similarity"] != similarity and not (similarity == "tanimoto" and kw["similarity"] == "jaccard"):
            continue
        kw = dict(kw, similarity=similarity)
        dense_ref = z["dense_%d" % n]
        dev = Compute_Similarity_MI355X(X, topK=0, **kw)
        W = dev.compute_similarity()
        assert rel_err(W, dense_ref) < RTOL
        dev.close()
        dev = Compute_Similarity_MI355X(X, topK=6, **kw)
        idx, val, _ = dev.compute_slabs()
        for c in range(X.shape[1]):
            check_topk_against_dense(idx[c], val[c], dense_ref[:, c], 6, RTOL)
        dev.close()


def test_topk_indices_equal_compute_similarity_python(gpu):
    """The device's top-K rule IS the rule of the reference's Compute_Similarity_Python (K largest cells of the FULL column,
    zeros compete, then are dropped): identical neighbour sets and order on the reference-generated fixture, including the
    adjusted / pearson columns where the Cython class emits stale-id artefacts instead (see
    tests/test_oracle_golden.py::test_topk_rule_python_reference_vs_cython_reference)."""
    z, cases = load_golden("similarity_topk_rules")
    X = unpack_csr(z, "X")
    n = X.shape[1]
    for k, kw in enumerate(cases):
        py = z["python_%d" % k]
        dev = Compute_Similarity_MI355X(X, **kw)
        idx, val, _ = dev.compute_slabs()
        W = dev.compute_similarity().toarray()
        dev.close()
        assert ((W != 0) == (py != 0)).all(), kw
        assert rel_err(W, py) < RTOL
        for c in range(n):
            want = np.flatnonzero(py[:, c])
            want = want[np.argsort(-py[want, c], kind="stable")]
            got = idx[c][idx[c] >= 0]
            np.testing.assert_array_equal(got, want)          # bit-exact indices, in descending-value order
        through_dispatcher = Compute_Similarity(X, **kw).compute_similarity().toarray()
        assert np.array_equal(through_dispatcher, W)


def test_golden_row_weights(gpu):
    z, _ = load_golden("similarity")
    X = unpack_csr(z, "X")
    dev = Compute_Similarity_MI355X(X, topK=0, shrink=2, row_weights=z["row_weights"])
    assert rel_err(dev.compute_similarity(), z["dense_rw"]) < RTOL


@pytest.mark.parametrize("similarity", ["cosine", "jaccard", "asymmetric", "tversky"])
@pytest.mark.parametrize("values", ["real", "binary"])
def test_seeded_ml1m_family(gpu, similarity, values):
    X = named_urm("ml1m", values, scale=0.25)           # 1510 x 926
    _check_build(X, 50, similarity=similarity, shrink=3, normalize=True, asymmetric_alpha=0.35, tversky_alpha=0.8, tversky_beta=1.1)


@pytest.mark.parametrize("topK", [1, 5, 100, 128, 129, 1000, 5000])
def test_topk_sizes(gpu, topK):
    X = synthetic_urm(700, 1200, 40000, 5, 300, seed=8, values="real")
    _check_build(X, topK, similarity="cosine", shrink=0)


def test_csr_assembly_matches_reference_layout(gpu):
    """compute_similarity() returns csr (n, n) float32 with column = source item, <= topK per column."""
    X = named_urm("ml1m", "real", scale=0.15)
    W = Compute_Similarity(X, topK=20, shrink=5, similarity="cosine").compute_similarity()
    assert sps.isspmatrix_csr(W) and W.dtype == np.float32 and W.shape == (X.shape[1], X.shape[1])
    Wo = O.OracleSimilarity(X, topK=20, shrink=5).compute_similarity(exact_numpy_topk=True)
    assert (np.diff(sps.csc_matrix(W).indptr) <= 20).all()
    # tie-free input: identical sparsity pattern and values within tolerance
    assert (W != 0).astype(np.int8).sum() == (Wo != 0).astype(np.int8).sum()
    diff = abs(W - Wo)
    assert diff.max() <= RTOL * abs(Wo).max()
    assert W.diagonal().max() == 0 and W.diagonal().min() == 0


def test_column_ranges_equal_full_build(gpu):
    # binary data: co-occurrence counts are exact in fp32 whatever the accumulation order, so two builds are
    # bit-identical (on real-valued data the LDS float atomics reorder sums at the 1e-7 level)
    X = named_urm("ml1m", "binary", scale=0.2)
    n = X.shape[1]
    dev = Compute_Similarity_MI355X(X, topK=30, shrink=1)
    full_idx, full_val, _ = dev.compute_slabs()
    cuts = [0, n // 5, n // 2, n - 1, n]
    for s, e in zip(cuts[:-1], cuts[1:]):
        idx, val, s0 = dev.compute_slabs(s, e)
        np.testing.assert_array_equal(idx, full_idx[s0:s0 + len(idx)])
        np.testing.assert_array_equal(val, full_val[s0:s0 + len(idx)])
    # out-of-range bounds fall back to the full range exactly like Compute_Similarity_Cython.pyx:447-451
    idx, val, s0 = dev.compute_slabs(-3, 10 ** 9)
    assert s0 == 0 and len(idx) == n
    dev.close()


def test_deterministic_across_runs_on_binary_data(gpu):
    """Integer co-occurrence counts are exact in fp32, ties are broken by the lower index: bit-reproducible."""
    X = named_urm("ml1m", "binary", scale=0.2)
    a = Compute_Similarity_MI355X(X, topK=40, similarity="jaccard").compute_slabs()
    b = Compute_Similarity_MI355X(X, topK=40, similarity="jaccard").compute_slabs()
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_edge_cases(gpu):
    # empty columns, empty rows, a single dense row, duplicate-free tiny matrix, topK > n_cols
    X = sps.csr_matrix(np.array([[1, 0, 2, 0, 0], [0, 0, 0, 0, 0], [3, 0, 1, 0, 4], [1, 0, 1, 0, 1]], dtype=np.float32))
    for similarity in ALL_SIMS:
        dev = Compute_Similarity_MI355X(X, topK=10, similarity=similarity, shrink=0)
        assert dev.TopK == 5
        idx, val, _ = dev.compute_slabs()
        orc = O.OracleSimilarity(X, topK=0, similarity=similarity, shrink=0)
        for c in range(5):
            check_topk_against_dense(idx[c], val[c], orc.column(c)[0], 5, RTOL)
        assert (idx[1] == -1).all() and (idx[3] == -1).all()      # cold columns have no neighbours
        dev.close()
    with pytest.raises(ValueError):
        Compute_Similarity_MI355X(X, row_weights=[1.0, 2.0])


@pytest.mark.parametrize("values,n_cols,topK", [("real", 6000, 5000), ("binary", 70000, 12000)])
def test_topk_beyond_the_lds_selection(gpu, values, n_cols, topK):
    """topK > 4096 (the in-LDS selection's candidate buffer) and per-tile candidates beyond the merge buffer: the reference only clamps
    topK to n_cols (.pyx:146).  Dense columns + one segmented sort per block of columns; same rule (K largest of the full column,
    zeros compete, then are dropped; ties to the lower id)."""
    X = synthetic_urm(900, n_cols, 40 * n_cols // 10, 5, 600, seed=14, values=values, zipf_exponent=0.5)
    dev = Compute_Similarity_MI355X(X, topK=topK, shrink=1, similarity="pearson" if values == "real" else "cosine")
    rng = np.random.default_rng(6)
    s = int(rng.integers(0, n_cols - 300))
    idx, val, s0 = dev.compute_slabs(s, s + 300)
    assert idx.shape == (300, topK) and s0 == s
    orc = O.OracleSimilarity(X, topK=0, shrink=1, similarity="pearson" if values == "real" else "cosine")
    for c in range(s, s + 300, 13):
        check_topk_against_dense(idx[c - s], val[c - s], orc.column(c)[0], topK, RTOL)
    if values == "real":                      # the csr_matrix entry point, whole matrix
        W = dev.compute_similarity()
        assert W.shape == (n_cols, n_cols)
        c = s + 7
        col = np.zeros(n_cols); Wc = W.tocsc()
        col[Wc.indices[Wc.indptr[c]:Wc.indptr[c + 1]]] = Wc.data[Wc.indptr[c]:Wc.indptr[c + 1]]
        got = idx[7][idx[7] >= 0]
        np.testing.assert_allclose(col[got], val[7][:len(got)], rtol=1e-6)
        assert (col != 0).sum() == len(got)
    dev.close()


def test_itemknn_recommender_end_to_end(gpu):
    X = named_urm("ml1m", "real", scale=0.2)
    rec = ItemKNNCFRecommender(X, verbose=False)
    rec.fit(topK=25, shrink=10, similarity="cosine")
    Wo = O.OracleSimilarity(rec.URM_train, topK=25, shrink=10).compute_similarity(exact_numpy_topk=True)
    assert abs(rec.W_sparse - Wo).max() <= RTOL * abs(Wo).max()
    users = np.arange(50)
    ranked, scores = rec.recommend(users, cutoff=10, return_scores=True)
    expected = rec.URM_train[users].dot(Wo).toarray()
    assert rel_err(scores[np.isfinite(scores)], expected[np.isfinite(scores)]) < 1e-4
    assert all(len(r) == 10 for r in ranked)


def test_full_size_ml20m_shape_properties(gpu):
    """BASELINE size (138k x 27k, 20M nnz): too big for the oracle in a test, so size-independent properties:
    symmetric measure => s(a,b) == s(b,a) wherever both survive the top-K, values in (0, 1], sorted, no diagonal,
    and a random sample of columns is checked against the oracle column by column."""
    X = named_urm("ml20m", "binary")
    dev = Compute_Similarity_MI355X(X, topK=100, shrink=0, similarity="cosine")
    idx, val, _ = dev.compute_slabs()
    n = X.shape[1]
    assert idx.shape == (n, 100)
    valid = idx >= 0
    assert (val[valid] > 0).all() and (val[valid] <= 1.0 + 1e-5).all()
    assert (np.diff(val, axis=1) <= 1e-7).all()
    assert (idx != np.arange(n)[:, None]).all()
    W = sps.csr_matrix((val[valid], (idx[valid], np.broadcast_to(np.arange(n)[:, None], idx.shape)[valid])), shape=(n, n))
    both = W.multiply(W.T > 0) - W.T.multiply(W > 0)
    assert abs(both).max() < 1e-5
    orc = O.OracleSimilarity(X, topK=0)
    cost = dev.column_costs()
    heavy = np.argsort(-cost)[:40]                       # the head of the catalogue: the columns the schedule splits
    sample = np.unique(np.concatenate([heavy, np.random.default_rng(0).choice(n, 200, replace=False)]))
    dense_cols = {int(c): orc.column(int(c))[0] for c in sample}
    for c in sample:
        check_topk_against_dense(idx[c], val[c], dense_cols[int(c)], 100, RTOL)
    st = dev.stats()
    assert st["n_units"] == n and st["kernel_ms"] > 0
    # the head of the catalogue as one GPU's share of an 8-way sharded build: its columns are several times a
    # workgroup's fair share, so the default schedule splits them over workgroups (publish + last-arriver sum): checked
    # against the ORACLE column by column, and -- integer counts -- bit-identical to the unsplit build
    for rep in range(3):
        idx_h, val_h, _ = dev.compute_slabs(None, 300)
        assert dev.schedule_info()[1] > 0
        assert (idx_h == idx[:300]).all() and (val_h == val[:300]).all()
    for c in heavy[heavy < 300]:
        check_topk_against_dense(idx_h[c], val_h[c], dense_cols[int(c)], 100, RTOL)
    assert (heavy < 300).sum() >= 20
    dev.close()


def test_baseline_config_4_netflix_shape_properties(gpu):
    """BASELINE.json configs[3]: full cosine similarity on the Netflix-Prize-shaped URM (480 189 x 17 770, 100 M nnz);
    here on ONE GPU plus the cost-balanced 8-way column cut the multi-GPU build would use.  Properties as for the
    ML-20M case, and mid-popularity columns are checked against the oracle."""
    from recsys2019_deeplearning_evaluation_amd.sharding import balanced_column_ranges
    X = named_urm("netflix", "binary")
    dev = Compute_Similarity_MI355X(X, topK=100, shrink=0, similarity="cosine")
    idx, val, _ = dev.compute_slabs()
    n = X.shape[1]
    valid = idx >= 0
    assert idx.shape == (n, 100) and (val[valid] > 0).all() and (val[valid] <= 1.0 + 1e-5).all()
    assert (np.diff(val, axis=1) <= 1e-7).all() and (idx != np.arange(n)[:, None]).all()
    cost = dev.column_costs()
    ranges = balanced_column_ranges(cost, 8)
    loads = np.array([cost[s:e].sum() for s, e in ranges], dtype=float)
    assert loads.max() < 1.1 * loads.mean() + cost.max()
    # a column range computed on its own equals the same rows of the full build (what each rank of the sharded build does)
    s, e = ranges[3]
    part_idx, part_val, s0 = dev.compute_slabs(s, e)
    np.testing.assert_array_equal(part_idx, idx[s0:s0 + len(part_idx)])
    np.testing.assert_array_equal(part_val, val[s0:s0 + len(part_idx)])
    orc = O.OracleSimilarity(X, topK=0)
    by_cost = np.argsort(cost)
    sample = np.unique(np.concatenate([by_cost[-12:], by_cost[[n // 2, n // 2 + 1, n // 3]],
                                       np.random.default_rng(1).choice(n, 200, replace=False)]))
    for c in sample:
        check_topk_against_dense(idx[c], val[c], orc.column(int(c))[0], 100, RTOL)
    # rank 0's share of the 8-way build holds the head columns, which its schedule splits over workgroups: oracle-checked
    s, e = ranges[0]
    head_idx, head_val, s0 = dev.compute_slabs(s, e)
    assert dev.schedule_info()[1] > 0
    for c in by_cost[-12:]:
        if s <= c < e:
            check_topk_against_dense(head_idx[c - s0], head_val[c - s0], orc.column(int(c))[0], 100, RTOL)
    dev.close()


@pytest.mark.parametrize("values,similarity", [("real", "cosine"), ("binary", "jaccard"), ("real", "pearson")])
def test_wide_matrices_use_accumulator_tiles(gpu, values, similarity):
    """More columns than LDS cells (32 256): the accumulator is tiled over the neighbour ids and the per-tile top-K
    candidates are merged.  70 000 columns = 3 tiles; every column is checked against the oracle."""
    X = synthetic_urm(2500, 70000, 420000, 20, 600, seed=11, values=values, zipf_exponent=0.6)
    dev = Compute_Similarity_MI355X(X, topK=40, shrink=2, similarity=similarity)
    idx, val, _ = dev.compute_slabs()
    orc = O.OracleSimilarity(X, topK=0, shrink=2, similarity=similarity)
    for c in range(0, X.shape[1], 7):
        check_topk_against_dense(idx[c], val[c], orc.column(c)[0], 40, RTOL)
    # a column range and the dense variant go through the same tiles
    part_idx, part_val, s0 = dev.compute_slabs(33000, 33100)
    np.testing.assert_array_equal(part_idx, idx[33000:33100])
    dev.close()
    dense = Compute_Similarity_MI355X(X[:, :40000], topK=0, shrink=2, similarity=similarity)
    W = dense.compute_similarity(start_col=32000, end_col=32300)
    orc2 = O.OracleSimilarity(X[:, :40000], topK=0, shrink=2, similarity=similarity)
    for c in (32000, 32255, 32256, 32299):
        assert rel_err(W[:, c], orc2.column(c)[0]) < RTOL or np.abs(orc2.column(c)[0]).max() == 0
    dense.close()


def test_userknn_recommender_on_a_wide_user_base(gpu):
    """UserKNN = the same build on URM.T: 40 000 users -> two accumulator tiles."""
    from recsys2019_deeplearning_evaluation_amd import UserKNNCFRecommender
    X = synthetic_urm(40000, 900, 500000, 5, 200, seed=12, values="binary")
    rec = UserKNNCFRecommender(X, verbose=False)
    rec.fit(topK=15, shrink=1, similarity="cosine")
    assert rec.W_sparse.shape == (40000, 40000) and (np.diff(rec.W_sparse.tocsc().indptr) <= 15).all()
    orc = O.OracleSimilarity(rec.URM_train.T.tocsr(), topK=0, shrink=1)
    Wc = rec.W_sparse.tocsc()
    for c in (0, 17, 32255, 32256, 39999):
        col = np.zeros(40000, np.float32); col[Wc.indices[Wc.indptr[c]:Wc.indptr[c + 1]]] = Wc.data[Wc.indptr[c]:Wc.indptr[c + 1]]
        want = orc.column(c)[0]
        order = np.argsort(-col, kind="stable")[:15]
        check_topk_against_dense(np.where(col[order] > 0, order, -1).astype(np.int32)[np.argsort(np.where(col[order] > 0, 0, 1), kind="stable")],
                                 np.sort(col[order])[::-1], want, 15, RTOL)
    scores = rec._compute_item_score(np.arange(5))
    assert scores.shape == (5, 900)


@pytest.mark.parametrize("weighting", ["BM25", "TF-IDF"])
@pytest.mark.parametrize("user_based", [False, True])
def test_knn_with_feature_weighting(gpu, weighting, user_based):
    """BM25 / TF-IDF run as a device pre-pass of the build; the recommender's URM_train becomes the re-weighted matrix exactly as in
    the reference (ItemKNNCFRecommender.py:40-48): checked against the NumPy restatement of Base/IR_feature_weighting.py (itself
    checked against reference outputs in tests/test_host_logic.py), then the build against the oracle on that matrix."""
    from recsys2019_deeplearning_evaluation_amd import UserKNNCFRecommender
    from oracle.feature_weighting import apply_feature_weighting
    X = named_urm("ml1m", "real", scale=0.15)
    rec = (UserKNNCFRecommender if user_based else ItemKNNCFRecommender)(X, verbose=False)
    rec.fit(topK=20, shrink=5, similarity="cosine", feature_weighting=weighting)
    want = apply_feature_weighting(X, weighting, user_based)
    assert sps.isspmatrix_csr(rec.URM_train) and rec.URM_train.shape == X.shape
    assert abs(rec.URM_train - X).max() > 0                       # the recommender's URM is re-weighted, like the reference's
    np.testing.assert_array_equal(rec.URM_train.indptr, want.indptr)
    np.testing.assert_allclose(rec.URM_train.toarray(), want.toarray(), rtol=RTOL, atol=1e-7)
    M = rec.URM_train.T.tocsr() if user_based else rec.URM_train
    Wo = O.OracleSimilarity(M, topK=20, shrink=5).compute_similarity(exact_numpy_topk=True)
    assert abs(rec.W_sparse - Wo).max() <= RTOL * abs(Wo).max()
    with pytest.raises(ValueError):
        ItemKNNCFRecommender(X, verbose=False).fit(feature_weighting="nope")


def test_feature_weighting_reference_fixture_on_device(gpu):
    """The reference-generated fixture of okapi_BM_25(X.T).T / TF_IDF(X.T).T against the device pre-pass, both document orientations."""
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
    z, _ = load_golden("feature_weighting")
    X = unpack_csr(z, "X").astype(np.float32)
    for weighting, key in (("BM25", "bm25_T"), ("TF-IDF", "tfidf_T")):
        a = Compute_Similarity_MI355X(X, topK=5, feature_weighting=weighting, weighting_documents="columns")
        np.testing.assert_allclose(a.weighted_matrix().toarray(), z[key], rtol=RTOL, atol=1e-7)
        b = Compute_Similarity_MI355X(X.T.tocsr(), topK=5, feature_weighting=weighting, weighting_documents="rows")
        np.testing.assert_allclose(b.weighted_matrix().toarray(), z[key].T, rtol=RTOL, atol=1e-7)
        a.close(); b.close()
    with pytest.raises(ValueError):
        Compute_Similarity_MI355X(X, topK=5).weighted_matrix()


@pytest.mark.parametrize("values,similarity", [("binary", "cosine"), ("real", "cosine"), ("real", "adjusted"), ("binary", "tversky")])
def test_heavy_columns_split_over_workgroups(gpu, values, similarity, monkeypatch):
    """Columns above half of a workgroup's fair share are accumulated by several workgroups (parts of the column's
    users) and summed by the last one to arrive: same result as the oracle, and -- integer counts -- bit-identical to
    the unsplit build on all-ones data."""
    X = named_urm("ml1m", values, scale=0.25)
    kw = dict(topK=50, shrink=3, normalize=True, similarity=similarity)
    monkeypatch.setenv("MI355REC_SIM_MIN_PART_USERS", "1000000")
    plain = Compute_Similarity_MI355X(X, **kw)
    idx0, val0, _ = plain.compute_slabs()
    assert plain.schedule_info()[1] == 0
    monkeypatch.setenv("MI355REC_SIM_MIN_PART_USERS", "64")
    split = Compute_Similarity_MI355X(X, **kw)
    idx1, val1, _ = split.compute_slabs()
    n_items, n_split, n_parts = split.schedule_info()
    assert n_split > 10 and n_parts >= 2 * n_split and n_items == X.shape[1] - n_split + n_parts
    if values == "binary":
        assert (idx0 == idx1).all() and (val0 == val1).all()
    orc = O.OracleSimilarity(X, **dict(kw, topK=0))
    for c in range(X.shape[1]):
        check_topk_against_dense(idx1[c], val1[c], orc.column(c)[0], 50, RTOL)
    # column ranges and the dense (topK = 0) output go through the same split schedule
    s, e = 3, X.shape[1] // 5
    idx2, val2, _ = split.compute_slabs(s, e)
    assert split.schedule_info()[1] > 0
    assert (idx2 == idx1[s:e]).all() and np.allclose(val2, val1[s:e], rtol=1e-6, atol=0)
    dense = Compute_Similarity_MI355X(X, **dict(kw, topK=0))
    W = dense.compute_similarity(0, 40)
    assert dense.schedule_info()[1] > 0
    for c in range(0, 40, 7):
        assert rel_err(W[:, c], orc.column(c)[0]) < RTOL
    for o in (plain, split, dense):
        o.close()


# ------------------------------------------------------------------------------------------------------------
#   Euclidean similarity (Compute_Similarity_Euclidean.py): every pair of columns has a
#   value, float32 arithmetic as in the reference
# ------------------------------------------------------------------------------------------------------------

def test_euclidean_golden_fixture(gpu):
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_Euclidean_MI355X
    z, cases = load_golden("euclidean")
    X = unpack_csr(z, "X")                      # integer ratings: the reference's float32 sums are exact
    n = X.shape[1]
    for k, kw in enumerate(cases):
        want = z["dense_%d" % k]
        dev = Compute_Similarity_Euclidean_MI355X(X, topK=n, **kw)
        W = dev.compute_similarity().toarray()
        assert (np.diag(W) == 0).all() and ((W != 0) == (want != 0)).all(), kw
        assert rel_err(W, want) < RTOL, (kw, rel_err(W, want))
        top = Compute_Similarity_Euclidean_MI355X(X, topK=6, **kw)
        idx, val, _ = top.compute_slabs()
        for c in range(n):
            check_topk_against_dense(idx[c], val[c], want[:, c].astype(np.float64), 6, RTOL)
        # through the dispatcher, as the KNN recommenders call it
        W2 = Compute_Similarity(X, similarity="euclidean", topK=n, **kw).compute_similarity().toarray()
        assert np.array_equal(W2, W)
    # jittered ratings: the reference's own float32 accumulation noise passes through a^2 + b^2 - 2ab
    Xj = unpack_csr(z, "Xj")
    Wj = Compute_Similarity_Euclidean_MI355X(Xj, topK=Xj.shape[1], **cases[0]).compute_similarity().toarray()
    assert rel_err(Wj, z["densej_0"]) < 1e-4


def test_euclidean_row_weights_golden_fixture_and_oracle(gpu):
    """row_weights (Compute_Similarity_Euclidean.py:62-72): weighted dot products, and the distances to the columns times the weights
    of the rows (:174-175; square inputs).  The reference's own outputs for float32 and float64 weights (the device works in
    float32: 1e-5), then a larger square matrix against the oracle incl. a column range."""
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_Euclidean_MI355X
    z, cases = load_golden("euclidean_row_weights")
    X = unpack_csr(z, "X")
    n = X.shape[1]
    for k, case in enumerate(cases):
        kw = dict(case["kw"], row_weights=z[case["weights"]])
        want = z["dense_%d" % k]
        W = Compute_Similarity_Euclidean_MI355X(X, topK=n, **kw).compute_similarity().toarray()
        assert (np.diag(W) == 0).all() and ((W != 0) == (want != 0)).all(), case
        assert rel_err(W, want) < RTOL, (case, rel_err(W, want))
        idx, val, _ = Compute_Similarity_Euclidean_MI355X(X, topK=5, **kw).compute_slabs()
        for c in range(n):
            check_topk_against_dense(idx[c], val[c], want[:, c].astype(np.float64), 5, RTOL)
    X = synthetic_urm(700, 700, 30000, seed=12, values="real")
    X.data = np.round(X.data)
    w = np.random.default_rng(12).uniform(0.2, 2.5, 700).astype(np.float32)
    kw = dict(shrink=1, normalize=True, normalize_avg_row=True, similarity_from_distance_mode="log", row_weights=w)
    want = O.OracleSimilarityEuclidean(X, topK=25, **kw).dense().astype(np.float64)
    idx, val, _ = Compute_Similarity_Euclidean_MI355X(X, topK=25, **kw).compute_slabs()
    for c in range(700):
        check_topk_against_dense(idx[c], val[c], want[:, c], 25, RTOL)
    idx2, val2, _ = Compute_Similarity_Euclidean_MI355X(X, topK=25, **kw).compute_slabs(100, 333)
    assert (idx2 == idx[100:333]).all() and (val2 == val[100:333]).all()
    # all-ones data with weights: still the wide accumulator (the count kernel has no weights)
    Xb = X.copy(); Xb.data[:] = 1.0
    wantb = O.OracleSimilarityEuclidean(Xb, topK=25, **kw).dense().astype(np.float64)
    idxb, valb, _ = Compute_Similarity_Euclidean_MI355X(Xb, topK=25, **kw).compute_slabs()
    # (two all-ones columns with the same support are at distance 0; the reference's float32 a^2 + b^2 - 2ab can round below zero
    # there and its sqrt is nan -- documented deviation: the device clamps to 0 -- so columns holding a nan are left out)
    clean = [c for c in range(0, 700, 7) if not np.isnan(wantb[:, c]).any()]
    assert len(clean) > 50
    for c in clean:
        check_topk_against_dense(idxb[c], valb[c], wantb[:, c], 25, RTOL)


@pytest.mark.parametrize("values,mode", [("binary", "lin"), ("real", "log"), ("binary", "exp")])
def test_euclidean_against_the_oracle(gpu, values, mode):
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_Euclidean_MI355X
    X = named_urm("ml1m", values, scale=0.15)
    if values == "real":
        X.data = np.round(X.data)
    kw = dict(shrink=1, normalize=True, normalize_avg_row=(mode != "lin"), similarity_from_distance_mode=mode)
    orc = O.OracleSimilarityEuclidean(X, topK=20, **kw)
    want = orc.dense().astype(np.float64)
    idx, val, _ = Compute_Similarity_Euclidean_MI355X(X, topK=20, **kw).compute_slabs()
    assert (idx >= 0).all()                      # every column has n - 1 non-zero neighbours
    for c in range(X.shape[1]):
        check_topk_against_dense(idx[c], val[c], want[:, c], 20, RTOL)
    s, e = 5, 60                                 # column ranges
    idx2, val2, _ = Compute_Similarity_Euclidean_MI355X(X, topK=20, **kw).compute_slabs(s, e)
    assert (idx2 == idx[s:e]).all() and (val2 == val[s:e]).all()


def test_euclidean_wide_matrix_uses_accumulator_tiles(gpu):
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_Euclidean_MI355X
    X = synthetic_urm(300, 17000, 60000, seed=3, values="real")     # > 16128 float64 cells: two tiles
    X.data = np.round(X.data)
    kw = dict(shrink=0, normalize=False, normalize_avg_row=True, similarity_from_distance_mode="lin")
    idx, val, _ = Compute_Similarity_Euclidean_MI355X(X, topK=30, **kw).compute_slabs()
    orc = O.OracleSimilarityEuclidean(X, topK=30, **kw)
    for c in (0, 1, 777, 16127, 16128, 16999):
        check_topk_against_dense(idx[c], val[c], orc.columns(c, c + 1)[:, 0].astype(np.float64), 30, RTOL)


def test_euclidean_errors_and_knn_recommender(gpu):
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_Euclidean_MI355X
    X = named_urm("ml1m", "real", scale=0.1)
    X.data = np.round(X.data)
    with pytest.raises(ValueError):
        Compute_Similarity_Euclidean_MI355X(X, similarity_from_distance_mode="sqrt")
    with pytest.raises(ValueError):              # not square: the reference's `item_distance * row_weights` cannot broadcast either
        Compute_Similarity_Euclidean_MI355X(X, row_weights=np.ones(X.shape[0], np.float32))
    with pytest.raises(ValueError):
        Compute_Similarity_Euclidean_MI355X(X, row_weights=np.ones(X.shape[0] + 1, np.float32))
    rec = ItemKNNCFRecommender(X.copy(), verbose=False)
    rec.fit(topK=15, shrink=0, similarity="euclidean", normalize=True, normalize_avg_row=False, similarity_from_distance_mode="exp")
    want = O.OracleSimilarityEuclidean(rec.URM_train, topK=15, shrink=0, normalize=True, similarity_from_distance_mode="exp").dense()
    W = rec.W_sparse.toarray()
    assert ((W != 0).sum(axis=0) == 15).all()
    assert np.abs(W[W != 0] - want[W != 0]).max() < RTOL * np.abs(want).max()


def test_csr_assembled_on_the_device_equals_the_host_assembly(gpu):
    """compute_similarity() returns the CSR matrix built on the device (sort by neighbour id); it must be the very
    matrix the COO assembly of the slabs gives (Compute_Similarity_Cython.pyx:603-605), sorted indices included."""
    from recsys2019_deeplearning_evaluation_amd.similarity import slabs_to_csr
    cases = [(named_urm("ml1m", "real", scale=0.2), dict(topK=30, shrink=2, similarity="cosine"), None),
             (named_urm("ml1m", "binary", scale=0.2), dict(topK=7, shrink=0, similarity="jaccard"), (11, 301)),
             (synthetic_urm(300, 40000, 90000, seed=4, values="binary"), dict(topK=20, shrink=0, similarity="cosine"), None)]
    for X, kw, rng in cases:
        dev = Compute_Similarity_MI355X(X, **kw)
        args = () if rng is None else rng
        W = dev.compute_similarity(*args)
        idx, val, s = dev.compute_slabs(*args)
        want = slabs_to_csr(idx, val, s, X.shape[1])
        want.sort_indices()
        assert W.shape == want.shape and W.nnz == want.nnz and W.dtype == np.float32
        assert np.array_equal(W.indptr, want.indptr) and np.array_equal(W.indices, want.indices)
        assert np.array_equal(W.data, want.data)
        dev.close()


# ---- real-valued data at the BASELINE shapes, on BOTH accumulator types (int64 fixed point / float64) ----------------------
@pytest.mark.parametrize("shape", ["ml20m", "netflix"])
def test_real_valued_full_shape_fixed_point_and_float64_accumulators(gpu, shape, monkeypatch):
    """Ratings (real-valued data) at the ML-20M and Netflix shapes: cosine, one set-based measure and pearson.  The most costly
    columns (the ones the schedule splits) + 200 random ones are checked against the oracle column by column -- once on the
    default accumulator (int64 fixed point wherever the host's bound admits it) and once with MI355REC_SIM_F64_SUMS=1 (float64
    sums, like the reference's double array) -- and the two builds must agree on EVERY column: identical neighbour ids in
    identical order, values within 2e-6 relative."""
    X = named_urm(shape, "real")
    n = X.shape[1]
    for similarity, extra in (("cosine", {}), ("dice", {}), ("pearson", dict(shrink=3))):
        kw = dict(topK=100, similarity=similarity, **extra)
        builds = {}
        for forced in (False, True):
            if forced:
                monkeypatch.setenv("MI355REC_SIM_F64_SUMS", "1")
            else:
                monkeypatch.delenv("MI355REC_SIM_F64_SUMS", raising=False)
            dev = Compute_Similarity_MI355X(X, **kw)
            kind, scale = dev.accumulator_info()
            if similarity == "dice":
                assert kind == "uint32"                    # set-based measures binarise the data (.pyx:219-230)
            elif similarity == "cosine":
                assert kind == ("float64" if forced else "int64-fixed"), (similarity, kind, scale)
            else:
                # pearson: a sparse column of nearly equal ratings has a tiny centred norm, for which the worst-case bound
                # may refuse fixed point -- either way the forced build is float64
                assert kind == "float64" if forced else kind in ("int64-fixed", "float64")
            idx, val, _ = dev.compute_slabs()
            cost = dev.column_costs()
            dev.close()
            builds[forced] = (idx, val)
        monkeypatch.delenv("MI355REC_SIM_F64_SUMS", raising=False)
        (idx_a, val_a), (idx_b, val_b) = builds[False], builds[True]
        np.testing.assert_array_equal(idx_a, idx_b)
        assert np.abs(val_a - val_b).max() <= 2e-6 * np.abs(val_b).max()
        orc = O.OracleSimilarity(X, **dict(kw, topK=0))
        heavy = np.argsort(-cost)[:40 if shape == "ml20m" else 12]
        sample = np.unique(np.concatenate([heavy, np.random.default_rng(2).choice(n, 200, replace=False)]))
        for c in sample:
            col = orc.column(int(c))[0]
            check_topk_against_dense(idx_a[c], val_a[c], col, 100, RTOL)
            check_topk_against_dense(idx_b[c], val_b[c], col, 100, RTOL)


def test_fixed_point_bound_rejects_wide_dynamic_range(gpu, monkeypatch):
    """Data on which no power-of-two scale keeps the worst-case rounding below the bar (values over eight decades, a column whose
    norm is tiny, row_weights): the constructor must fall back to float64 sums -- this keeps the ds_add_f64 branch covered -- and
    the result must match the oracle; the same matrix without its tiny column is admitted to fixed point again."""
    monkeypatch.delenv("MI355REC_SIM_F64_SUMS", raising=False)
    rng = np.random.default_rng(4)
    X = synthetic_urm(3000, 700, 90000, 5, 300, seed=21, values="real").tolil()
    X[:, 5] = 0.0
    X[7, 5] = 1e-4                                    # one cell: column norm 1e-4
    X[11, 5] = 2e-4
    X = X.tocsr()
    X.data[rng.random(X.nnz) < 0.01] *= 1e4           # and a few very large ratings
    X.eliminate_zeros(); X.sort_indices()
    w = (0.5 + rng.random(X.shape[0]) * 40).astype(np.float64)
    for kw in (dict(similarity="cosine"), dict(similarity="cosine", row_weights=w), dict(similarity="asymmetric", asymmetric_alpha=0.3)):
        dev = Compute_Similarity_MI355X(X, topK=30, **kw)
        assert dev.accumulator_info()[0] == "float64", kw
        idx, val, _ = dev.compute_slabs()
        orc = O.OracleSimilarity(X, topK=0, **kw)
        for c in range(X.shape[1]):
            check_topk_against_dense(idx[c], val[c], orc.column(c)[0], 30, RTOL)
        dev.close()
    Y = X.tolil(); Y[:, 5] = 0.0; Y[0, 5] = 3.0; Y = Y.tocsr()
    Y.data = np.minimum(Y.data, 6.0)
    dev = Compute_Similarity_MI355X(Y, topK=30, similarity="cosine")
    kind, scale = dev.accumulator_info()
    assert kind == "int64-fixed" and scale > 1.0
    idx, val, _ = dev.compute_slabs()
    orc = O.OracleSimilarity(Y, topK=0, similarity="cosine")
    for c in range(0, Y.shape[1], 3):
        check_topk_against_dense(idx[c], val[c], orc.column(c)[0], 30, RTOL)
    dev.close()


@pytest.mark.parametrize("weighting", ["BM25", "TF-IDF"])
def test_knn_euclidean_with_feature_weighting(gpu, weighting):
    """similarity="euclidean" must accept the KNN recommenders' feature weighting like every other similarity
    (run_parameter_search.py:219-239 searches the combination): the weighted matrix replaces URM_train and the build runs on it."""
    from oracle.feature_weighting import apply_feature_weighting
    X = named_urm("ml1m", "real", scale=0.1)
    X.data = np.round(X.data)
    rec = ItemKNNCFRecommender(X.copy(), verbose=False)
    rec.fit(topK=12, shrink=1, similarity="euclidean", normalize=False, feature_weighting=weighting, similarity_from_distance_mode="lin")
    want_urm = apply_feature_weighting(X, weighting, False)
    np.testing.assert_allclose(rec.URM_train.toarray(), want_urm.toarray(), rtol=RTOL, atol=1e-7)
    want = O.OracleSimilarityEuclidean(rec.URM_train, topK=12, shrink=1, normalize=False, similarity_from_distance_mode="lin").dense()
    W = rec.W_sparse.toarray()
    assert ((W != 0).sum(axis=0) == 12).all()
    assert np.abs(W[W != 0] - want[W != 0]).max() < 1e-4 * np.abs(want).max()


@pytest.mark.parametrize("step,similarity", [(1.0, "cosine"), (0.5, "cosine"), (0.25, "asymmetric"), (1.0, "euclidean")])
def test_quantised_ratings_use_exact_int32_sums(gpu, step, similarity, monkeypatch):
    """Star / half-star ratings: every product is an integer multiple of step^2, the column sums are exact in int32 cells
    (accumulator "int32-exact", one LDS tile where 8-byte cells need two at ML-20M width) -- same result as the 8-byte-cell
    kernel and as the oracle; jittered ratings, row_weights and mean-centred similarities stay on the wide cells."""
    monkeypatch.delenv("MI355REC_SIM_F64_SUMS", raising=False)
    X = synthetic_urm(2500, 18000, 260000, 5, 700, seed=31, values="real")       # 18 000 columns: one 4-byte tile, two 8-byte tiles
    X.data = (np.random.default_rng(8).integers(1, int(5 / step) + 1, X.nnz) * step).astype(np.float32)       # step, 2 step, ..., 5.0
    assert (X.data * (1.0 / step) == np.round(X.data * (1.0 / step))).all() and (step == 1.0 or (X.data != np.round(X.data)).any())
    kw = dict(topK=40, shrink=2, similarity=similarity)
    if similarity == "asymmetric":
        kw["asymmetric_alpha"] = 0.3
    if similarity == "euclidean":
        from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_Euclidean_MI355X
        make = lambda M: Compute_Similarity_Euclidean_MI355X(M, topK=40, shrink=2, normalize=True, similarity_from_distance_mode="log")
        orc_cols = O.OracleSimilarityEuclidean(X, topK=40, shrink=2, normalize=True, similarity_from_distance_mode="log")
        column = lambda c: orc_cols.columns(c, c + 1)[:, 0].astype(np.float64)
    else:
        make = lambda M: Compute_Similarity_MI355X(M, **kw)
        orc = O.OracleSimilarity(X, **dict(kw, topK=0))
        column = lambda c: orc.column(c)[0]
    dev = make(X)
    kind, scale = dev.accumulator_info()
    assert kind == "int32-exact" and scale == (1.0 / step) ** 2
    idx, val, _ = dev.compute_slabs()
    dev.close()
    monkeypatch.setenv("MI355REC_SIM_NO_INT32", "1")
    wide = make(X)
    assert wide.accumulator_info()[0] == "int64-fixed"
    idx_w, val_w, _ = wide.compute_slabs()
    wide.close()
    monkeypatch.delenv("MI355REC_SIM_NO_INT32")
    np.testing.assert_array_equal(idx, idx_w)
    assert np.abs(val - val_w).max() <= 2e-6 * np.abs(val_w).max()
    for c in np.random.default_rng(3).choice(X.shape[1], 150, replace=False):
        check_topk_against_dense(idx[c], val[c], column(int(c)), 40, RTOL)
    if similarity == "cosine" and step == 1.0:
        Xj = X.copy(); Xj.data = Xj.data + np.float32(1e-3) * np.random.default_rng(5).random(X.nnz).astype(np.float32)
        for M, extra in ((Xj, {}), (X, dict(row_weights=np.arange(1, X.shape[0] + 1, dtype=np.float64) % 7 + 1)), (X, dict(similarity="pearson"))):
            other = Compute_Similarity_MI355X(M, **dict(kw, **extra))
            assert other.accumulator_info()[0] in ("int64-fixed", "float64")
            other.close()


def _slabs_both_selections(X, monkeypatch, **kw):
    """(idx, val) of the threshold-first selection and of the full normalise + radix select, and the first one's counters."""
    out = []
    for switch in ("1", "0"):
        monkeypatch.setenv("MI355REC_SIM_FAST_TOPK", switch)
        dev = Compute_Similarity_MI355X(X, **kw)
        idx, val, _ = dev.compute_slabs()
        out.append((idx.copy(), val.copy(), dev.selection_info()))
        dev.close()
    monkeypatch.delenv("MI355REC_SIM_FAST_TOPK")
    assert out[1][2] == (0, 0, 0)                # switched off: no column is counted
    return out[0], out[1]


@pytest.mark.parametrize("similarity,extra", [("cosine", dict(shrink=0)), ("cosine", dict(shrink=10)), ("asymmetric", dict(shrink=5, asymmetric_alpha=0.3)),
                                              ("jaccard", dict(shrink=0)), ("dice", dict(shrink=3)), ("tversky", dict(shrink=1, tversky_alpha=0.7, tversky_beta=1.5)),
                                              ("cosine", dict(shrink=20, normalize=False)), ("cosine", dict(shrink=0, normalize=False))])
@pytest.mark.parametrize("values", ["binary", "stars"])
def test_threshold_first_topk_equals_full_selection(gpu, similarity, extra, values, monkeypatch):
    """The threshold-first selection (thread maxima of approximate values -> K-th largest maximum -> exact division of the survivors)
    emits what the full path emits, bit for bit -- indices, order and values -- for every denominator, on counts and on exact
    int32 sums, for dense columns (1024-thread instance: 12 000 columns) and K from 1 to THREADS / 4."""
    X = synthetic_urm(9000, 12000, 1_500_000, min_len=5, max_len=3000, seed=77, values="binary")
    if values == "stars":
        rng = np.random.default_rng(5)
        X.data[:] = rng.integers(1, 11, X.nnz) * 0.5                   # half stars: exact int32 sums
    for topK in (1, 100, 256):
        (idx, val, info), (idx0, val0, _) = _slabs_both_selections(X, monkeypatch, topK=topK, similarity=similarity, **extra)
        np.testing.assert_array_equal(idx, idx0)
        np.testing.assert_array_equal(val, val0)
        assert info[0] > 0.9 * X.shape[1], info                  # nearly every column of this matrix has K positive thread maxima
        assert info[2] == 0, info
        assert info[1] < 4 * max(topK, 16) * info[0], info       # the bound is tight: a few candidates beyond K per column
    orc = O.OracleSimilarity(X, topK=0, similarity=similarity, **extra)
    for c in range(0, X.shape[1], 997):
        check_topk_against_dense(idx[c], val[c], orc.column(c)[0], 256, RTOL)


def test_threshold_first_topk_small_instance_and_sparse_columns(gpu, monkeypatch):
    """512-thread instance (narrow matrix), columns with fewer than K positive cells (they take the full path), K at the limit."""
    X = named_urm("ml1m", "binary", scale=0.5)
    for topK in (5, 128):
        (idx, val, info), (idx0, val0, _) = _slabs_both_selections(X, monkeypatch, topK=topK, shrink=2)
        np.testing.assert_array_equal(idx, idx0)
        np.testing.assert_array_equal(val, val0)
        assert info[0] > 0, info
    Xs = synthetic_urm(400, 3000, 6000, min_len=1, max_len=40, seed=3, values="binary")      # every column has < K neighbours
    (idx, val, info), (idx0, val0, _) = _slabs_both_selections(Xs, monkeypatch, topK=50, shrink=0)
    np.testing.assert_array_equal(idx, idx0)
    np.testing.assert_array_equal(val, val0)
    assert info[0] < 0.02 * Xs.shape[1] and info[2] == 0, info     # light columns (< 16 K pair-adds) are scheduled straight onto the full path
    Xm = synthetic_urm(2500, 3000, 60000, min_len=2, max_len=300, seed=4, values="binary")   # in between: some columns reach K positive maxima
    (idx, val, info), (idx0, val0, _) = _slabs_both_selections(Xm, monkeypatch, topK=50, shrink=0)
    np.testing.assert_array_equal(idx, idx0)
    np.testing.assert_array_equal(val, val0)
    assert 0 < info[0] < Xm.shape[1], info


def test_threshold_first_topk_falls_back_on_masses_of_equal_values(gpu, monkeypatch):
    """Every user holds every item: all cells of a column are equal, every one of them passes the bar -- more survivors than the
    candidate buffer holds (4 096), so the column is selected by the full path (lowest ids win the tie) and counted as a fall-back."""
    n_items = 12000
    X = sps.csr_matrix(np.ones((40, n_items), dtype=np.float32))
    (idx, val, info), (idx0, val0, _) = _slabs_both_selections(X, monkeypatch, topK=10, shrink=0)
    np.testing.assert_array_equal(idx, idx0)
    np.testing.assert_array_equal(val, val0)
    assert info[2] == n_items and info[0] == 0, info
    assert (idx[5] == [0, 1, 2, 3, 4, 6, 7, 8, 9, 10]).all()


def test_resident_urm_build_equals_host_build(gpu):
    """mi355rec_sim_create_resident: the constructor that copies a URM already in HBM gives the same handle as the one that
    uploads it -- identical slabs, with feature weighting too (the handle re-weights its own copy, the resident one stays as it
    was and serves the next fit) -- and refuses a resident copy of a different matrix."""
    from recsys2019_deeplearning_evaluation_amd import ResidentURM
    X = named_urm("ml1m", "real", scale=0.4)
    res = ResidentURM(X)
    for kw in (dict(topK=20, shrink=3), dict(topK=20, shrink=3, feature_weighting="BM25"), dict(topK=0, similarity="jaccard")):
        out = []
        for resident in (None, res):
            dev = Compute_Similarity_MI355X(X, resident=resident, **kw)
            out.append(dev.compute_similarity())
            dev.close()
        assert (out[0] != out[1]).nnz == 0 if sps.issparse(out[0]) else np.array_equal(out[0], out[1])
    other = X.copy()
    other.data[0] += 1.0                                    # (entry 0 is always part of the sample)
    with pytest.raises(ValueError, match="does not hold this dataMatrix"):
        Compute_Similarity_MI355X(other, topK=5, resident=res)
    other = X.copy()
    other.data[7] += 1.0                                    # not in the fixed sample: the full comparison of a first presentation sees it
    assert not res.matches(other) and not res.matches(other, thorough=True) and res.matches(X.copy(), thorough=True)
    same = X.copy()
    assert res.matches(same) and res._verified[res._buffers_of(same)] is True      # ... and the verdict is kept for these buffers
    assert res.matches(same)
    rec, rec_res = ItemKNNCFRecommender(X, verbose=False), ItemKNNCFRecommender(X, verbose=False)
    rec.fit(topK=10, shrink=1)
    rec_res.fit(topK=10, shrink=1, resident_urm=res)
    assert (rec.W_sparse != rec_res.W_sparse).nnz == 0
    res.close()


def test_device_cache_trim(gpu):
    """mi355rec_device_trim: the blocks a closed handle left in the library's cache go back to the driver (a process that shares the
    device with another allocator calls it between phases); builds afterwards work as before."""
    from recsys2019_deeplearning_evaluation_amd import _native as N
    X = named_urm("ml1m", "binary", scale=0.5)
    dev = Compute_Similarity_MI355X(X, topK=10)
    before = dev.compute_similarity()
    dev.close()
    assert N.trim_device_cache() > 0                     # the constructor's temporaries and the handle's arrays were cached
    assert N.trim_device_cache() == 0
    dev = Compute_Similarity_MI355X(X, topK=10)
    assert (dev.compute_similarity() != before).nnz == 0
    dev.close()


def test_lds_atomic_rate_is_measured(gpu):
    """mi355rec_lds_atomic_rate: the peak bench.py prices the column kernel against, measured on the device at hand -- within a factor of
    two of round 1's microbenchmark (21.6 lane-adds per ns and CU)."""
    from recsys2019_deeplearning_evaluation_amd import _native as N
    rate = N.lds_atomic_rate()
    assert 0.5 * 21.6e9 * 256 < rate < 2.0 * 21.6e9 * 256, rate


def test_closing_a_handle_does_not_wait_for_another_handles_kernels(gpu):
    """Blocks that go back to the library's cache wait for the streams of the handle that owned them (csrc/common.h ReleaseScope), not for
    the device: while one thread trains an IALS model (hundreds of milliseconds of kernels on its own stream), another thread builds
    and closes small similarity handles -- each constructor + build + close stays far below the time the IALS call has left, and the
    builds are as correct as ever."""
    import threading
    import time
    from recsys2019_deeplearning_evaluation_amd import IALS_MI355X_Epoch
    big = named_urm("ml20m", "binary", scale=0.35)
    conf = big.copy()
    conf.data = (1.0 + conf.data).astype(np.float32)
    k = 128
    ia = IALS_MI355X_Epoch(conf, k, 1e-3, k ** -0.5 * np.random.default_rng(0).random((big.shape[1], k)))
    ia.run_epochs(1)
    t0 = time.perf_counter()
    ia.run_epochs(1)
    epoch_s = time.perf_counter() - t0
    n_epochs = max(4, int(np.ceil(1.0 / max(epoch_s, 1e-3))))           # about a second of kernels
    X = named_urm("ml1m", "binary", scale=0.3)
    want = Compute_Similarity_MI355X(X, topK=10).compute_similarity()
    spans, running = [], threading.Event()

    def train():
        running.set()
        ia.run_epochs(n_epochs)
        running.clear()

    th = threading.Thread(target=train)
    th.start()
    running.wait(10)
    time.sleep(0.05)
    while running.is_set() and len(spans) < 200:
        t1 = time.perf_counter()
        dev = Compute_Similarity_MI355X(X, topK=10)
        got = dev.compute_similarity()
        dev.close()
        spans.append(time.perf_counter() - t1)
        assert (got != want).nnz == 0
    th.join(120)
    ia.close()
    total = n_epochs * epoch_s
    assert len(spans) >= 3, (len(spans), total)
    # with the device-wide wait every close() lasted until the IALS call ended: one or two handles per call
    assert np.median(spans) < 0.2 * total, (np.median(spans), total, len(spans))


def test_packed_counts_kernel_equals_the_one_launch_build(gpu, monkeypatch):
    """sim_packed_kernel (all-ones data: two 16-bit counts per LDS word, two workgroups per CU, the 32-bit kernel's launch behind it for
    what it hands over) against the one-launch build, bit for bit -- on a URM whose most popular items have MORE than 65 535 users (their
    columns are accumulated in parts and added up by the second launch), with split columns and light columns, three denominator forms."""
    X = synthetic_urm(200000, 12000, 12000000, 5, 400, seed=7)
    assert X[:, 0].nnz >= 65536 and X[:, 11999].nnz < 2000
    for kw in (dict(topK=50, shrink=0), dict(topK=100, shrink=5, similarity="jaccard"), dict(topK=120, shrink=0, similarity="asymmetric", asymmetric_alpha=0.3),
               dict(topK=1, shrink=2, similarity="dice"), dict(topK=128, shrink=0, similarity="tversky", tversky_alpha=0.7, tversky_beta=1.3),
               dict(topK=10, shrink=3, normalize=False), dict(topK=64, shrink=0, normalize=False)):
        out = {}
        for packed in ("1", "0"):
            monkeypatch.setenv("MI355REC_SIM_PACKED", packed)
            dev = Compute_Similarity_MI355X(X, **kw)
            idx, val, _ = dev.compute_slabs()
            out[packed] = (idx, val, dev.stats()["n_launches"])
            dev.close()
        assert out["1"][2] == 2 and out["0"][2] == 1, (out["1"][2], out["0"][2])
        np.testing.assert_array_equal(out["1"][0], out["0"][0])
        np.testing.assert_array_equal(out["1"][1], out["0"][1])
    monkeypatch.delenv("MI355REC_SIM_PACKED")


def test_non_finite_values_are_refused_like_the_reference_dispatcher(gpu):
    """Compute_Similarity.py:34-36 asserts np.isfinite over the data array; here the library's constructor finds non-finite values in its
    own pass over the uploaded values (the host no longer scans 80 MB twice per fit) and the dispatcher raises the reference's message."""
    X = named_urm("ml1m", "real", scale=0.2)
    for bad in (np.nan, np.inf, -np.inf):
        Y = X.copy()
        Y.data[[3, 77]] = bad
        with pytest.raises(AssertionError, match="Data matrix contains 2 non finite values"):
            Compute_Similarity(Y, topK=5, shrink=0)
        with pytest.raises(ValueError, match="non finite"):
            Compute_Similarity_MI355X(Y, topK=5, shrink=0, similarity="jaccard")
    Compute_Similarity(X, topK=5, shrink=0).compute_similarity()
