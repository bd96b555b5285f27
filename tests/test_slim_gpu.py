"""Parity of the HIP SLIM-BPR epoch (through the C ABI) against the CPU oracle.

Replay mode: the oracle draws the (u, i, j) stream with glibc rand() exactly like the reference and runs its
strictly sequential SGD in float64; the device executes the same stream (level-scheduled on the dense store, in
order on the symmetric store) on a float32 S.  Tolerances: see test_mf_gpu.py -- max-norm 1e-5 for sgd; for the
adaptive optimisers the per-item 1/sqrt(cache) scaling amplifies float32 storage rounding, so the error
distribution is checked instead."""
import numpy as np
import pytest
import scipy.sparse as sps

from oracle import oracle as O
from recsys2019_deeplearning_evaluation_amd import SLIM_BPR_MI355X, SLIM_BPR_MI355X_Epoch
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm, synthetic_urm
from _util import load_golden, unpack_csr
from test_mf_gpu import assert_factor_parity

pytestmark = pytest.mark.gpu
MODES = ["sgd", "adagrad", "rmsprop", "adam"]


def _replay(X, epochs, **kw):
    orc = O.OracleSLIM(X, topK=False, final_model_sparse_weights=False, **kw)
    orc.record_samples(10 ** 7)
    for _ in range(epochs):
        orc.epochIteration_Cython()
    u, i, j = orc.recorded()
    dev = SLIM_BPR_MI355X_Epoch(X, topK=False, final_model_sparse_weights=False, **kw)
    dev.replay_samples(u, i, j)
    return orc, dev, (u, i, j)


def test_golden_fixture_replay(gpu):
    z, cases = load_golden("slim_bpr")
    X = unpack_csr(z, "X")
    for n, case in enumerate(cases):
        orc, dev, _ = _replay(X, case["epochs"], **case["kw"])
        assert_factor_parity(dev.get_S_dense(), z["S_%d" % n], case["kw"]["sgd_mode"], "S")
        dev.close()


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("mode", MODES)
def test_replay_parity(gpu, symmetric, mode):
    X = named_urm("ml1m", "binary", scale=0.1)           # 604 x 370
    orc, dev, (u, i, j) = _replay(X, 6, symmetric=symmetric, random_seed=13, sgd_mode=mode, learning_rate=0.01,
                                  li_reg=0.003, lj_reg=0.005)
    S = dev.get_S_dense()
    assert_factor_parity(S, orc.get_S_dense(), mode, "S")
    assert (np.diag(S) == 0).all()
    if symmetric:
        np.testing.assert_array_equal(S, S.T)
    st = dev.stats()
    assert st["n_units"] == len(u) and st["algorithmic_bytes"] > 0
    dev.close()


@pytest.mark.parametrize("symmetric", [False, True])
def test_topk_extraction_matches_get_S(gpu, symmetric):
    X = named_urm("ml1m", "binary", scale=0.1)
    orc, dev, _ = _replay(X, 4, symmetric=symmetric, random_seed=3, sgd_mode="sgd", learning_rate=0.05, li_reg=0.01, lj_reg=0.02)
    S = dev.get_S_dense().astype(np.float64)
    for topK in [1, 10, 150, 5000]:
        idx, val = dev.get_S_slabs(topK)
        want = O.oracle_similarity_topk_rows(S, topK, zeros_compete=symmetric)      # same S, reference selection rule
        k = idx.shape[1]
        for r in range(S.shape[0]):
            got = idx[r][idx[r] >= 0]
            row = want[r]
            order = np.lexsort((row.indices, -row.data))
            np.testing.assert_array_equal(got, row.indices[order])
            np.testing.assert_array_equal(val[r][:len(got)], row.data[order].astype(np.float32))
            assert (idx[r][len(got):] == -1).all() and len(got) <= k
    dev.topK = 10
    dev.final_model_sparse_weights = True
    W = dev.get_S()
    assert sps.isspmatrix_csr(W) and W.shape == S.shape and (np.diff(W.indptr) <= 10).all()
    dev.close()


def test_native_epoch_is_a_valid_stream_and_matches_oracle(gpu):
    X = named_urm("ml1m", "binary", scale=0.12)
    kw = dict(symmetric=False, random_seed=21, sgd_mode="sgd", learning_rate=0.02, li_reg=0.001, lj_reg=0.001)
    a = SLIM_BPR_MI355X_Epoch(X, topK=False, **kw)
    a.epochIteration_Cython()
    assert a.stats()["n_units"] == X.shape[0] + 1                  # .pyx:215 with batch_size 1
    b = SLIM_BPR_MI355X_Epoch(X, topK=False, **kw)
    b.epochIteration_Cython()
    np.testing.assert_array_equal(a.get_S_dense(), b.get_S_dense())   # same seed -> same stream -> same model
    S = a.get_S_dense()
    assert np.isfinite(S).all() and (S != 0).sum() > 0 and (np.diag(S) == 0).all()
    # BPR pushes S[i, seen] up and S[j, seen] down: column sums over a user's seen items end up positive on average
    assert S[X[0].indices][:, X[0].indices].mean() > 0


def test_recommender_fit_surface(gpu):
    X = named_urm("ml1m", "binary", scale=0.1)
    for symmetric in (True, False):
        rec = SLIM_BPR_MI355X(X, verbose=False)
        rec.fit(epochs=15, symmetric=symmetric, learning_rate=0.05, topK=20, sgd_mode="adagrad", random_seed=4)
        assert sps.isspmatrix_csr(rec.W_sparse) and rec.W_sparse.shape == (X.shape[1], X.shape[1])
        assert (np.diff(rec.W_sparse.tocsc().indptr) <= 20).all()
        scores = rec._compute_item_score(np.arange(60))
        dense = X[:60].toarray() > 0
        assert np.mean([scores[r][dense[r]].mean() > scores[r][~dense[r]].mean() for r in range(60)]) > 0.9
    with pytest.raises(NotImplementedError):
        SLIM_BPR_MI355X(X, verbose=False).fit(epochs=1, train_with_sparse_weights=True)
    with pytest.raises(ValueError):
        SLIM_BPR_MI355X(X, verbose=False).fit(epochs=1, topK=0)


def test_baseline_config_3_ml20m_shape_properties(gpu):
    """BASELINE.json configs[2]: SLIM-BPR top-k=100 on the ML-20M-shaped URM (dense S = 2.86 GB in HBM).  Too big for the
    float64 oracle in a test, so size-independent properties of one native epoch on the dense store: only sampled
    (i, j) rows of S are touched, S[i, seen] went up and S[j, seen] went down, the diagonal is zero, and get_S returns
    sorted non-zero rows."""
    X = named_urm("ml20m", "binary")
    dev = SLIM_BPR_MI355X_Epoch(X, topK=100, symmetric=False, sgd_mode="sgd", learning_rate=0.05, random_seed=5)
    dev.epochIteration_Cython()
    st = dev.stats()
    assert st["n_units"] == X.shape[0] + 1 and st["n_launches"] > 100        # level schedule, not one step per launch
    idx, val = dev.get_S_slabs(100)
    n = X.shape[1]
    assert idx.shape == (n, 100)
    valid = idx >= 0
    assert (val[valid] != 0).all() and (np.diff(np.where(valid, val, -1e30), axis=1) <= 0).all()
    assert (idx != np.arange(n)[:, None]).all()
    touched_rows = valid.any(axis=1).sum()
    assert 0.2 * n < touched_rows <= n                                        # ~139k steps over 26.7k item rows
    assert (val[valid] > 0).any() and (val[valid] < 0).any()
    dev.close()
