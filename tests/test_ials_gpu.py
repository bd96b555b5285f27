"""Parity of the HIP IALS solve step (through the C ABI) against the CPU oracle / golden fixtures.
float64 on both sides: tolerance 1e-8 relative (the only differences are summation order and Gauss-Jordan
vs. LAPACK inverse)."""
import numpy as np
import pytest
import scipy.sparse as sps

from oracle import oracle as O
from recsys2019_deeplearning_evaluation_amd import IALS_MI355X_Epoch, IALSRecommender
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
from _util import load_golden, rel_err, unpack_csr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def test_golden_fixture(gpu):
    z, cases = load_golden("ials")
    X = unpack_csr(z, "X")
    for n, case in enumerate(cases):
        kw = case["kw"]
        Cm = O.oracle_ials_confidence(X, kw["confidence_scaling"], kw["alpha"], kw.get("epsilon", 1.0))
        dev = IALS_MI355X_Epoch(Cm, kw["num_factors"], kw["reg"], z["V0_%d" % n])
        dev.run_epochs(case["epochs"])
        U, V = dev.get_factors()
        assert rel_err(U, z["U_%d" % n]) < RTOL
        assert rel_err(V, z["V_%d" % n]) < RTOL
        dev.close()
        # the same case END TO END through the package's recommender: its own confidence scaling (ials.py) and its own draw of
        # the initial item factors (same NumPy stream position as the reference's _init_factors, IALSRecommender.py:204-210)
        from recsys2019_deeplearning_evaluation_amd import IALSRecommender
        np.random.seed(404 + n)
        rec = IALSRecommender(X, verbose=False)
        rec.fit(epochs=case["epochs"], **kw)
        assert rel_err(rec.USER_factors, z["U_%d" % n]) < RTOL
        assert rel_err(rec.ITEM_factors, z["V_%d" % n]) < RTOL


@pytest.mark.parametrize("k", [1, 5, 32, 33, 64, 100, 160, 161, 200, 224, 225, 240, 255])
def test_factor_counts(gpu, k):
    X = named_urm("ml1m", "real", scale=0.08)
    Cm = O.oracle_ials_confidence(X, "linear", 2.0)
    Cc = sps.csc_matrix(Cm)
    rng = np.random.default_rng(k)
    V0 = k ** -0.5 * rng.random((X.shape[1], k))
    U = np.zeros((X.shape[0], k)); V = V0.copy()
    O.oracle_ials_epoch(Cm, Cc, U, V, 1e-3)
    dev = IALS_MI355X_Epoch(Cm, k, 1e-3, V0)
    dev.run_epochs(1)
    Ud, Vd = dev.get_factors()
    assert rel_err(Ud, U) < RTOL and rel_err(Vd, V) < RTOL
    # every solved row satisfies its normal equations (independent of the oracle)
    VV = V0.T @ V0
    for u in [0, 7, X.shape[0] - 1]:
        s, e = Cm.indptr[u], Cm.indptr[u + 1]
        Yi = V0[Cm.indices[s:e]]; c = Cm.data[s:e].astype(np.float64)
        Bm = VV + Yi.T @ ((c - 1)[:, None] * Yi) + 1e-3 * np.eye(k)
        assert np.abs(Bm @ Ud[u] - Yi.T @ c).max() < 1e-8 * max(1.0, np.abs(Yi.T @ c).max())
    st = dev.stats()
    assert st["n_units"] == X.shape[0] + X.shape[1] and st["algorithmic_flops"] > 0
    dev.close()


def test_small_regularisation_and_log_scaling(gpu):
    X = named_urm("ml1m", "real", scale=0.1)
    Cm = O.oracle_ials_confidence(X, "log", 5.0, 0.3)
    Cc = sps.csc_matrix(Cm)
    k = 48
    V0 = k ** -0.5 * np.random.default_rng(1).random((X.shape[1], k))
    U = np.zeros((X.shape[0], k)); V = V0.copy()
    for _ in range(3):
        O.oracle_ials_epoch(Cm, Cc, U, V, 1e-5)       # the hyper-parameter search goes down to reg = 1e-5
    dev = IALS_MI355X_Epoch(Cm, k, 1e-5, V0)
    dev.run_epochs(3)
    Ud, Vd = dev.get_factors()
    assert rel_err(Ud, U) < 1e-6 and rel_err(Vd, V) < 1e-6


def test_cold_rows_are_left_alone_and_halves_compose(gpu):
    X = named_urm("ml1m", "binary", scale=0.08).tolil()
    X[3, :] = 0; X[:, 5] = 0
    X = X.tocsr()
    Cm = O.oracle_ials_confidence(X, "linear", 1.0)
    k = 16
    rng = np.random.default_rng(2)
    V0 = rng.random((X.shape[1], k)); U0 = rng.random((X.shape[0], k))
    a = IALS_MI355X_Epoch(Cm, k, 1e-2, V0, U0); a.run_epochs(1)
    Ua, Va = a.get_factors()
    np.testing.assert_array_equal(Ua[3], U0[3]); np.testing.assert_array_equal(Va[5], V0[5])
    # the multi-GPU entry points: two user ranges + two item ranges == one epoch
    b = IALS_MI355X_Epoch(Cm, k, 1e-2, V0, U0)
    nu, ni = X.shape
    b.user_half(0, nu // 2); b.user_half(nu // 2, nu); b.item_half(0, ni // 3); b.item_half(ni // 3, ni); b.synchronize()
    Ub, Vb = b.get_factors()
    assert rel_err(Ub, Ua) < 1e-12 and rel_err(Vb, Va) < 1e-12


def test_too_many_factors_is_refused_not_miscomputed(gpu):
    X = named_urm("ml1m", "binary", scale=0.05)
    with pytest.raises(NotImplementedError):
        IALS_MI355X_Epoch(O.oracle_ials_confidence(X), 256, 1e-3, np.zeros((X.shape[1], 256)))


def test_recommender_surface(gpu):
    X = named_urm("ml1m", "binary", scale=0.12)
    np.random.seed(5)
    rec = IALSRecommender(X, verbose=False)
    rec.fit(epochs=3, num_factors=24, alpha=5.0, reg=1e-2)
    assert rec.USER_factors.shape == (X.shape[0], 24)
    scores = rec._compute_item_score(np.arange(50))
    dense = X[:50].toarray() > 0
    assert np.mean([scores[r][dense[r]].mean() > scores[r][~dense[r]].mean() for r in range(50)]) > 0.95
    with pytest.raises(ValueError):
        rec.fit(confidence_scaling="nope")


def test_baseline_config_5_ml20m_k200_properties(gpu):
    """BASELINE.json configs[4]: IALS k=200 on the ML-20M-shaped URM.  Oracle-free check at full size: after the user
    half-step every sampled user row solves its own normal equations (YtY + Y_I^T (C-1) Y_I + reg I) x = Y_I^T c against
    the ITEM factors it was solved with, and after the item half-step sampled item rows do against the updated users."""
    X = named_urm("ml20m", "binary")
    Cm = O.oracle_ials_confidence(X, "linear", 1.0)
    k, reg = 200, 1e-3
    V0 = k ** -0.5 * np.random.default_rng(0).random((X.shape[1], k))
    dev = IALS_MI355X_Epoch(Cm, k, reg, V0)
    nu, ni = X.shape
    dev.user_half(0, nu); dev.synchronize()
    U, V_same = dev.get_factors()
    np.testing.assert_array_equal(V_same, V0)
    VV = V0.T @ V0
    reg_diag = np.diag(reg * np.ones(k))
    rng = np.random.default_rng(1)

    def against_update_row(rows, C, Y, YtY, X_dev, what):
        """the device rows against the reference's _update_row (IALSRecommender.py:170-201, restated in oracle.py): 1e-8 of the row's scale"""
        for r in rows:
            s, e = C.indptr[r], C.indptr[r + 1]
            ref = O._ials_update_row(C.indices[s:e], C.data[s:e].astype(np.float64), Y, YtY, reg_diag)
            err = np.abs(X_dev[r] - ref).max() / max(np.abs(ref).max(), 1e-300)
            assert err < 1e-8, (what, int(r), e - s, err)

    for u in [0, 1, nu // 2, nu - 1, int(np.argmax(np.diff(Cm.indptr)))]:
        s, e = Cm.indptr[u], Cm.indptr[u + 1]
        Yi = V0[Cm.indices[s:e]]; c = Cm.data[s:e].astype(np.float64)
        Bm = VV + Yi.T @ ((c - 1)[:, None] * Yi) + reg * np.eye(k)
        rhs = Yi.T @ c
        assert np.abs(Bm @ U[u] - rhs).max() < 1e-7 * np.abs(rhs).max()
    Lu = np.diff(Cm.indptr)
    users = np.unique(np.concatenate([rng.choice(nu, 500, replace=False), np.argsort(-Lu)[:8]]))
    against_update_row(users, Cm, V0, VV, U, "user row")
    dev.item_half(0, ni); dev.synchronize()
    U2, V = dev.get_factors()
    np.testing.assert_array_equal(U2, U)
    UU = U.T @ U
    Cc = sps.csc_matrix(Cm)
    for i in [0, ni // 2, ni - 1]:
        s, e = Cc.indptr[i], Cc.indptr[i + 1]
        Yi = U[Cc.indices[s:e]]; c = Cc.data[s:e].astype(np.float64)
        Bm = UU + Yi.T @ ((c - 1)[:, None] * Yi) + reg * np.eye(k)
        rhs = Yi.T @ c
        assert np.abs(Bm @ V[i] - rhs).max() < 1e-7 * np.abs(rhs).max()
    n_split = dev.schedule_info()[0]
    assert n_split > 0, "the popular items' rows (profiles beyond 8192 entries) must have been split over workgroups"
    # every row whose profile was split over workgroups (parts published, summed by the last arriver), and 300 random ones
    Li = np.diff(Cc.indptr)
    split_rows = np.flatnonzero(Li > 2 * 4096)
    assert len(split_rows) == n_split, (len(split_rows), n_split)
    items = np.unique(np.concatenate([split_rows, rng.choice(ni, 300, replace=False)]))
    against_update_row(items, Cc, U, UU, V, "item row")
    st = dev.stats()
    assert st["algorithmic_flops"] > 0
    dev.close()


@pytest.mark.parametrize("k", [8, 40, 200])
def test_rows_split_over_workgroups_equal_unsplit_rows(gpu, k, monkeypatch):
    """A row with a long profile is accumulated by several workgroups (parts of its profile, published, added up in part order by
    the last arriver, which solves).  Forced on a small matrix (parts of 32 profile entries): against the oracle, and against the
    same epochs with every row on one workgroup."""
    # k = 200: a smaller matrix and one epoch (the oracle's 200 x 200 solves dominate the test's time on the GPU box's host)
    X = named_urm("ml1m", "real", scale=0.08 if k == 200 else 0.15)
    epochs = 1 if k == 200 else 2
    Cm = O.oracle_ials_confidence(X, "linear", 3.0)
    Cc = sps.csc_matrix(Cm)
    V0 = k ** -0.5 * np.random.default_rng(3).random((X.shape[1], k))
    U = np.zeros((X.shape[0], k)); V = V0.copy()
    for _ in range(epochs):
        O.oracle_ials_epoch(Cm, Cc, U, V, 1e-2)
    monkeypatch.setenv("MI355REC_IALS_PART_ROWS", "32")
    split = IALS_MI355X_Epoch(Cm, k, 1e-2, V0)
    split.run_epochs(epochs)
    n_split, n_parts = split.schedule_info()
    assert n_split > 20 and n_parts >= 3 * n_split, (n_split, n_parts)
    Us, Vs = split.get_factors()
    monkeypatch.delenv("MI355REC_IALS_PART_ROWS")
    monkeypatch.setenv("MI355REC_IALS_NO_SPLIT", "1")
    whole = IALS_MI355X_Epoch(Cm, k, 1e-2, V0)
    whole.run_epochs(epochs)
    assert whole.schedule_info() == (0, 0)
    Uw, Vw = whole.get_factors()
    assert rel_err(Us, U) < 1e-8 and rel_err(Vs, V) < 1e-8
    assert rel_err(Us, Uw) < 1e-10 and rel_err(Vs, Vw) < 1e-10


@pytest.mark.parametrize("k", [8, 48, 200, 207, 208])
def test_two_stage_epochs_equal_one_kernel_epochs(gpu, k, monkeypatch):
    """The default epoch builds the augmented systems of a batch of rows into HBM (ials_row_kernel, STAGE 1) and solves them with two
    workgroups per CU (ials_solve_kernel: a panel wavefront + seven tile wavefronts); MI355REC_IALS_TWO_STAGE=0 keeps the whole row in
    one workgroup.  The same operations on the same systems: the factors agree to 1e-12 (not bit for bit -- two compilations of the
    expressions, and Y^T Y itself is summed with atomics in arrival order in every epoch) -- with rows split over workgroups, with batches of a few rows
    (MI355REC_IALS_SYSTEM_GIB); where the solve stage does not apply (k > 207: 14 tiles per tile wavefront) the switch changes nothing."""
    X = named_urm("ml1m", "real", scale=0.12)
    Cm = O.oracle_ials_confidence(X, "linear", 3.0)
    V0 = k ** -0.5 * np.random.default_rng(k).random((X.shape[1], k))
    monkeypatch.setenv("MI355REC_IALS_PART_ROWS", "64")              # long rows split into parts of 64 profile entries
    out = {}
    for label, two_stage, gib in (("one", "0", None), ("two", "1", None), ("two-small-batches", "1", "0.004")):
        monkeypatch.setenv("MI355REC_IALS_TWO_STAGE", two_stage)
        if gib:
            monkeypatch.setenv("MI355REC_IALS_SYSTEM_GIB", gib)
        else:
            monkeypatch.delenv("MI355REC_IALS_SYSTEM_GIB", raising=False)
        dev = IALS_MI355X_Epoch(Cm, k, 1e-2, V0)
        dev.run_epochs(2)
        out[label] = dev.get_factors()
        assert dev.schedule_info()[0] > 0                               # split rows took part
        dev.close()
    assert rel_err(out["two"][0], out["one"][0]) < 1e-12 and rel_err(out["two"][1], out["one"][1]) < 1e-12
    assert rel_err(out["two-small-batches"][0], out["two"][0]) < 1e-12 and rel_err(out["two-small-batches"][1], out["two"][1]) < 1e-12
    U = np.zeros((X.shape[0], k)); V = V0.copy()
    Cc = sps.csc_matrix(Cm)
    for _ in range(2):
        O.oracle_ials_epoch(Cm, Cc, U, V, 1e-2)
    assert rel_err(out["two"][0], U) < RTOL and rel_err(out["two"][1], V) < RTOL
