"""Host-side logic that needs no GPU: recommender surface, slab assembly, column sharding, synthetic URMs."""
import numpy as np
import pytest
import scipy.sparse as sps

from oracle import oracle as O
from recsys2019_deeplearning_evaluation_amd import recommender_base as RB
from recsys2019_deeplearning_evaluation_amd.sharding import balanced_column_ranges
from recsys2019_deeplearning_evaluation_amd.similarity import slabs_to_csr
from recsys2019_deeplearning_evaluation_amd.slim_bpr import rows_slabs_to_csr
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm, synthetic_urm
from _util import csr_columns_as_slabs


def test_slabs_to_csr_equals_reference_coo_assembly():
    X = synthetic_urm(200, 80, 3000, 3, 40, seed=1, values="real")
    orc = O.OracleSimilarity(X, topK=7, shrink=1)
    idx, val = orc.build_slabs(0, 80)
    W = slabs_to_csr(idx, val, 0, 80)
    Wo = O.slabs_to_csr(idx, val, 0, 80)
    assert sps.isspmatrix_csr(W) and W.dtype == np.float32 and abs(W - Wo).max() == 0
    part = slabs_to_csr(idx[10:30], val[10:30], 10, 80)
    assert abs(part - Wo.multiply(sps.csr_matrix(([1.0] * 20, (range(20), range(10, 30))), shape=(20, 80)).sum(axis=0) > 0)).max() < 1e-7
    back_idx, back_val = csr_columns_as_slabs(W, 7)
    np.testing.assert_array_equal(back_idx, idx)


def test_rows_slabs_to_csr():
    idx = np.array([[2, 0, -1], [-1, -1, -1], [1, -1, -1]], np.int32)
    val = np.array([[0.5, 0.25, 0], [0, 0, 0], [-1.0, 0, 0]], np.float32)
    W = rows_slabs_to_csr(idx, val, 3)
    np.testing.assert_array_equal(W.toarray(), [[0.25, 0, 0.5], [0, 0, 0], [0, -1.0, 0]])


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_balanced_column_ranges(world):
    rng = np.random.default_rng(0)
    cost = (1e6 / np.arange(1, 2001) ** 0.8).astype(np.int64) + rng.integers(0, 50, 2000)     # Zipf-like skew
    ranges = balanced_column_ranges(cost, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == 2000 and len(ranges) == world
    assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:])) and all(e > s for s, e in ranges)
    loads = np.array([cost[s:e].sum() for s, e in ranges], dtype=float)
    assert loads.max() <= 1.25 * loads.mean() + cost.max()
    # equal-count ranges would be far worse on this skew
    if world >= 4:
        naive = np.array([c.sum() for c in np.array_split(cost, world)], dtype=float)
        assert loads.max() < naive.max()


def test_balanced_ranges_degenerate():
    assert balanced_column_ranges([5, 5], 8) == [(0, 1), (1, 2)]
    assert balanced_column_ranges([0, 0, 0, 0], 2) == [(0, 2), (2, 4)]
    assert balanced_column_ranges(np.ones(10), 1) == [(0, 10)]


def test_recommend_and_persistence(tmp_path):
    X = synthetic_urm(50, 30, 400, 2, 20, seed=2, values="binary")

    class Fixed(RB.BaseMatrixFactorizationRecommender):
        RECOMMENDER_NAME = "Fixed"

    rec = Fixed(X, verbose=False)
    rng = np.random.default_rng(0)
    rec.USER_factors = rng.normal(size=(50, 4)); rec.ITEM_factors = rng.normal(size=(30, 4))
    ranked, scores = rec.recommend(np.arange(5), cutoff=6, return_scores=True)
    for u in range(5):
        seen = set(X[u].indices.tolist())
        assert not seen & set(ranked[u]) and len(ranked[u]) == 6
        s = rec.USER_factors[u] @ rec.ITEM_factors.T
        s[list(seen)] = -np.inf
        assert ranked[u] == np.argsort(-s)[:6].tolist()
    single = rec.recommend(3, cutoff=4)
    assert single == ranked[3][:4]
    only = rec._compute_item_score(np.arange(2), items_to_compute=[1, 2])
    assert np.isinf(only[:, 0]).all() and np.isfinite(only[:, 1:3]).all()
    rec.set_items_to_ignore([0, 1])
    assert not {0, 1} & set(rec.recommend(0, cutoff=10, remove_custom_items_flag=True))
    rec.save_model(str(tmp_path) + "/", "m")
    other = Fixed(X, verbose=False)
    other.load_model(str(tmp_path) + "/", "m")
    np.testing.assert_array_equal(other.USER_factors, rec.USER_factors)
    assert other.use_bias is False


def test_similarity_topk_helper_matches_reference_semantics():
    rng = np.random.default_rng(1)
    S = rng.normal(size=(12, 12)); S[rng.random((12, 12)) < 0.5] = 0
    W = RB.similarityMatrixTopK(S, k=3)
    for c in range(12):
        col = S[:, c]; nz = np.flatnonzero(col)
        want = nz[np.argsort(col[nz])[-3:]]
        np.testing.assert_array_equal(np.sort(W[:, c].nonzero()[0]), np.sort(want))
    assert abs(RB.similarityMatrixTopK(sps.csr_matrix(S), k=3) - W).max() < 1e-7


def test_early_stopping_loop():
    class Toy(RB.Incremental_Training_Early_Stopping):
        verbose = False
        RECOMMENDER_NAME = "toy"

        def __init__(self):
            self.epochs_run = 0; self.best = None

        def _run_epoch(self, n):
            self.epochs_run += 1

        def _prepare_model_for_validation(self):
            pass

        def _update_best_model(self):
            self.best = self.epochs_run

    class Evaluator:
        def __init__(self):
            self.values = iter([0.1, 0.3, 0.2, 0.25, 0.1, 0.0])

        def evaluateRecommender(self, rec):
            return {10: {"MAP": next(self.values)}}, "str"

    t = Toy()
    t._train_with_early_stopping(7)
    assert t.epochs_run == 7 and t.epochs_best == 6 and t.best == 7
    t = Toy()
    t._train_with_early_stopping(100, validation_every_n=2, stop_on_validation=True, validation_metric="MAP",
                                 lower_validations_allowed=2, evaluator_object=Evaluator())
    assert t.epochs_best == 4 and t.best == 4 and t.epochs_run == 8
    with pytest.raises(AssertionError):
        Toy()._train_with_early_stopping(5, evaluator_object=Evaluator())


def test_synthetic_family_is_seeded_and_has_no_cold_rows():
    a = named_urm("ml1m", "real", scale=0.1); b = named_urm("ml1m", "real", scale=0.1)
    assert (a != b).nnz == 0 and a.has_sorted_indices
    assert (np.diff(a.indptr) > 0).all() and (np.diff(a.tocsc().indptr) > 0).all()
    assert len(np.unique(a.data)) > 1000 and (a.data != np.rint(a.data)).mean() > 0.99   # jittered ratings
    c = named_urm("ml1m", "binary", scale=0.1)
    assert (c.data == 1).all()


def test_feature_weighting_matches_reference_fixture():
    from _util import load_golden, unpack_csr
    from oracle.feature_weighting import apply_feature_weighting
    z, _ = load_golden("feature_weighting")
    X = unpack_csr(z, "X")
    np.testing.assert_allclose(apply_feature_weighting(X, "BM25", False).toarray(), z["bm25_T"].astype(np.float32), rtol=1e-6)
    np.testing.assert_allclose(apply_feature_weighting(X, "TF-IDF", True).toarray(), z["tfidf_T"].astype(np.float32), rtol=1e-6)
    assert apply_feature_weighting(X, "none", False) is X


def test_similarity_column_ranges_add_the_fixed_per_column_part():
    """Shard ranges for the similarity build balance pairs + a fixed per-column cost: with a head-heavy catalogue the
    pure pair count would give the tail rank nearly all the columns (and all of their clearing / ranking work)."""
    from recsys2019_deeplearning_evaluation_amd.sharding import FIXED_PAIRS_PER_CELL, similarity_column_ranges

    class Stub:
        n_columns = 1000

        def column_costs(self):
            c = np.full(1000, 10, dtype=np.int64)
            c[:10] = 1_000_000
            return c

    plain = balanced_column_ranges(Stub().column_costs(), 2)
    fixed = similarity_column_ranges(Stub(), 2)
    assert plain[0][1] <= 6                                   # half of the pairs = five head columns
    per_col = FIXED_PAIRS_PER_CELL * 1000
    est = lambda r: sum(Stub().column_costs()[r[0]:r[1]]) + per_col * (r[1] - r[0])
    assert abs(est(fixed[0]) - est(fixed[1])) <= 1_000_000 + per_col      # within one head column
    assert fixed[0][1] > plain[0][1]
    assert fixed[0][0] == 0 and fixed[-1][1] == 1000 and fixed[0][1] == fixed[1][0]


@pytest.mark.parametrize("n,density,k", [(60, 0.3, 5), (80, 0.05, 10), (50, 1.0, 7), (40, 0.5, 100)])
def test_similarity_matrix_topk_equals_the_column_loop(n, density, k):
    """similarityMatrixTopK (Base/Recommender_utils.py:55-122): per column the k largest NON-ZERO cells, negative values
    included, zeros never competing -- vectorised here, checked against the plain column loop on sparse and dense input."""
    from recsys2019_deeplearning_evaluation_amd.recommender_base import similarityMatrixTopK
    rng = np.random.default_rng(n)
    A = sps.random(n, n, density, random_state=np.random.RandomState(n), format="csr", dtype=np.float32)
    A.data = rng.standard_normal(A.nnz).astype(np.float32)
    D = A.toarray()
    want = np.zeros((n, n), np.float32)
    for c in range(n):
        nz = np.flatnonzero(D[:, c] != 0)
        top = nz[np.argsort(D[nz, c])[-min(k, n):]]
        want[top, c] = D[top, c]
    assert np.array_equal(similarityMatrixTopK(A, k).toarray(), want)
    assert np.array_equal(similarityMatrixTopK(D, k).toarray(), want)


def test_similarity_matrix_topk_reference_unit_tests():
    """The two cases of the reference's own Base/Recommender_utils_Test.py:18-50, on this package's similarityMatrixTopK."""
    from recsys2019_deeplearning_evaluation_amd.recommender_base import similarityMatrixTopK
    rng = np.random.RandomState(0)
    dense_input = rng.random_sample((100, 100))
    dense_output = similarityMatrixTopK(dense_input, k=20)
    assert (dense_output.toarray() != 0).sum() == 20 * 100                   # DenseToDense (:18-31)
    dense_input = rng.random_sample((20, 20))
    on_dense = similarityMatrixTopK(dense_input, k=5).toarray()
    on_sparse = similarityMatrixTopK(sps.csc_matrix(dense_input), k=5).toarray()
    assert np.allclose(on_dense, on_sparse)                                   # sparseToSparse (:34-50)


def test_interleaved_parts_are_equal_in_count_and_cost():
    """The serpentine deal of the cost order (the multi-GPU partition of the similarity build, restated on the host): a partition,
    counts within one column of each other, cost within the heaviest column of each other -- against contiguous ranges, which
    balance cost but not counts (that imbalance is what pads the all-gather slabs)."""
    from recsys2019_deeplearning_evaluation_amd.sharding import balanced_column_ranges, interleaved_parts
    rng = np.random.default_rng(4)
    cost = np.sort((rng.pareto(1.2, 5000) * 1000).astype(np.int64))[::-1].copy()        # popularity skew, heavy head first
    for n_parts in (1, 2, 3, 8):
        parts = interleaved_parts(cost, n_parts)
        assert sorted(np.concatenate(parts).tolist()) == list(range(len(cost)))
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
        sums = [int(cost[p].sum()) for p in parts]
        assert max(sums) - min(sums) <= int(cost.max())
    ranges = balanced_column_ranges(cost, 8)
    widths = [e - s for s, e in ranges]
    assert max(widths) > 2 * (len(cost) // 8)                       # equal-cost contiguous ranges are very unequal in width
    # ties in the cost keep the column order (stable), as the device's cost order does
    np.testing.assert_array_equal(interleaved_parts(np.ones(10, np.int64), 2)[0], [0, 3, 4, 7, 8])


def test_resident_urm_fingerprint_tells_matrices_apart():
    """ResidentURM.matches (the check behind `resident=`): same matrix -> same fingerprint; a changed value, index or shape -> not."""
    from recsys2019_deeplearning_evaluation_amd._native import ResidentURM
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    X = named_urm("ml1m", "real", scale=0.2)
    f = ResidentURM.fingerprint_of(X)
    assert ResidentURM.fingerprint_of(X.copy()) == f
    Y = X.copy(); Y.data[0] += 1.0
    assert ResidentURM.fingerprint_of(Y) != f
    Z = X.copy(); Z.indices[0], Z.indices[1] = Z.indices[1], Z.indices[0]
    assert ResidentURM.fingerprint_of(Z) != f
    assert ResidentURM.fingerprint_of(X[:-1]) != f


def test_ials_row_ranges_price_a_row_at_its_solve_as_well_as_its_profile():
    """sharding.ials_row_ranges: cost(row) = (L + 0.9 k) k^2 (fitted to measured ranges, profiles/r6_ials_ranges.txt).  The ranges
    cover every row once, and on a popularity-skewed matrix the tail range (many short rows) carries FEWER stored values than the
    head range (few long rows) by about the rows' solve cost -- with the flop-count model (k / 3 entries per row) it carried as
    many and ran 36 % longer."""
    import scipy.sparse as sps
    from recsys2019_deeplearning_evaluation_amd.sharding import ials_row_ranges, IALS_ROW_ENTRIES_PER_FACTOR
    from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
    X = synthetic_urm(3000, 2000, 90000, 2, 400, seed=11)
    k, world = 40, 4
    ur, ir = ials_row_ranges(X, world, k)
    for ranges, n in ((ur, X.shape[0]), (ir, X.shape[1])):
        assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    li = np.diff(sps.csc_matrix(X).indptr).astype(np.float64)
    cost = [(li[s:e].sum() + IALS_ROW_ENTRIES_PER_FACTOR * k * (e - s)) for s, e in ir]
    assert max(cost) / min(cost) < 1.15, cost
    nnz = [li[s:e].sum() for s, e in ir]
    assert nnz[-1] < nnz[0]                      # the tail range holds more rows and therefore fewer values
