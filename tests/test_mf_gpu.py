"""Parity of the HIP BPR-MF / FunkSVD epochs (through the C ABI) against the CPU oracle.

Replay mode: the oracle draws the sample stream with glibc rand() exactly like the reference; the device runs
the same stream from the same initial factors; float32 factors must agree within 1e-5 relative (north_star).
Native mode: the device draws its own samples; they must be valid samples of the reference's sampler, and
replaying THAT stream through the oracle must reproduce the device factors.
"""
import numpy as np
import pytest

from oracle import oracle as O
from recsys2019_deeplearning_evaluation_amd import (MatrixFactorization_BPR_MI355X, MatrixFactorization_FunkSVD_MI355X,
                                                    MatrixFactorization_MI355X_Epoch, MatrixFactorization_MI355X_Group)
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm, synthetic_urm
from _util import load_golden, rel_err, unpack_csr

pytestmark = pytest.mark.gpu
RTOL = 1e-5
MODES = ["sgd", "adagrad", "rmsprop", "adam"]
# Tolerance: element-wise |dev - ref| <= 1e-5 |ref| + 1e-6 max|ref| on the float32 outputs (north_star), for EVERY optimiser.  Plain sgd keeps
# float32 factors on the device.  adagrad / rmsprop / adam divide every gradient component by (sqrt(running g^2) + 1e-8):
# a component whose mini-batch gradient nearly cancels turns float32 rounding into an O(lr) error, so for these modes the
# device keeps factors, biases and moments in float64 (precision="auto"), exactly like the reference's `double` arrays.


def assert_factor_parity(dev, ref, mode, what):
    """north_star: "within 1e-5 relative on float32 factor matrices" -- read ELEMENT-WISE: |dev - ref| <= 1e-5 |ref| + 1e-6 max|ref|
    for every entry (the absolute term is the float32 resolution floor for entries near zero), not only relative to the largest."""
    dev = np.asarray(dev, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    tol = RTOL * np.abs(ref) + 1e-6 * max(np.abs(ref).max(), 1e-30)
    excess = np.abs(dev - ref) - tol
    assert excess.max() <= 0, (what, mode, "worst excess over the element-wise tolerance", excess.max(),
                               "norm-wise error", np.abs(dev - ref).max() / max(np.abs(ref).max(), 1e-30))


def _replay_case(X, kw, epochs):
    mode = kw.get("sgd_mode", "sgd")
    orc = O.OracleMF(X, **kw)
    orc.record_samples(10 ** 7)
    for _ in range(epochs):
        orc.epochIteration_Cython()
    u, i, j, r = orc.recorded()
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    if kw["algorithm_name"] == "MF_BPR":
        dev.replay_samples(u, i, neg_item=j)
    else:
        dev.replay_samples(u, i, rating=r)
    assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), mode, "U")
    assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), mode, "V")
    if kw.get("use_bias"):
        assert_factor_parity(dev.get_USER_bias(), orc.get_USER_bias(), mode, "bu")
        assert_factor_parity(dev.get_ITEM_bias(), orc.get_ITEM_bias(), mode, "bi")
        assert abs(float(dev.get_GLOBAL_bias()) - float(orc.get_GLOBAL_bias())) < 1e-4 * max(1e-3, abs(float(orc.get_GLOBAL_bias())))
    st = dev.stats()
    assert st["n_units"] == len(u)
    assert abs(st["loss"] - orc.cumulative_loss()) <= 1e-3 * max(1.0, orc.cumulative_loss()) or epochs > 1
    dev.close()
    return dev


def test_golden_fixture_replay(gpu):
    """The committed reference outputs: oracle stream replayed on the device lands on the reference's factors."""
    z, cases = load_golden("matrix_factorization")
    mats = {"Xb": unpack_csr(z, "Xb"), "Xr": unpack_csr(z, "Xr")}
    for n, case in enumerate(cases):
        X, kw = mats[case["matrix"]], case["kw"]
        orc = O.OracleMF(X, **kw)
        orc.record_samples(10 ** 6)
        for _ in range(case["epochs"]):
            orc.epochIteration_Cython()
        u, i, j, r = orc.recorded()
        dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                               initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
        dev.replay_samples(u, i, neg_item=j) if kw["algorithm_name"] == "MF_BPR" else dev.replay_samples(u, i, rating=r)
        mode = kw["sgd_mode"]
        assert_factor_parity(dev.get_USER_factors(), z["U_%d" % n], mode, "U")
        assert_factor_parity(dev.get_ITEM_factors(), z["V_%d" % n], mode, "V")
        if kw.get("use_bias"):
            assert_factor_parity(dev.get_ITEM_bias(), z["bi_%d" % n], mode, "bi")
            assert_factor_parity(dev.get_USER_bias(), z["bu_%d" % n], mode, "bu")
        dev.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [1, 10, 64, 128, 200, 300])
def test_bpr_replay_parity(gpu, mode, k):
    X = named_urm("ml1m", "binary", scale=0.2)
    kw = dict(n_factors=k, algorithm_name="MF_BPR", batch_size=100, random_seed=11, sgd_mode=mode, learning_rate=0.05,
              user_reg=0.002, positive_reg=0.003, negative_reg=0.004)
    _replay_case(X, kw, epochs=4)


@pytest.mark.parametrize("batch_size", [1, 7, 1000, 5000])
def test_bpr_batch_sizes(gpu, batch_size):
    X = named_urm("ml1m", "binary", scale=0.2)
    kw = dict(n_factors=64, algorithm_name="MF_BPR", batch_size=batch_size, random_seed=5, sgd_mode="sgd", learning_rate=0.05)
    _replay_case(X, kw, epochs=2 if batch_size > 1 else 1)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("use_bias", [False, True])
def test_funk_replay_parity(gpu, mode, use_bias):
    X = named_urm("ml1m", "real", scale=0.12)
    kw = dict(n_factors=32, algorithm_name="FUNK_SVD", batch_size=256, random_seed=21, sgd_mode=mode, learning_rate=0.01,
              user_reg=0.01, item_reg=0.5, positive_reg=0.0, bias_reg=0.02, use_bias=use_bias, negative_interactions_quota=0.35)
    _replay_case(X, kw, epochs=1)


@pytest.mark.parametrize("mode", ["sgd", "adam"])
def test_long_streams_in_several_calls_and_schedule_parts(gpu, mode):
    """A stream longer than the in-LDS schedule holds (256 mini-batches) is scheduled part by part, each part a stream of its own
    for the global mini-batch index (Adam's t) and the global-bias ring; so is every call.  600 mini-batches of FunkSVD with
    biases, in one call and handed over in three calls of 300 + 1 + 299 mini-batches, must land where the oracle lands."""
    X = named_urm("ml1m", "real", scale=0.08)
    kw = dict(n_factors=16, algorithm_name="FUNK_SVD", batch_size=16, random_seed=5, sgd_mode=mode, learning_rate=0.01,
              user_reg=0.01, item_reg=0.3, positive_reg=0.0, bias_reg=0.02, use_bias=True, negative_interactions_quota=0.2)
    orc = O.OracleMF(X, **kw)
    orc.record_samples(10 ** 7)
    orc.epochIteration_Cython()
    u, i, j, r = orc.recorded()
    n = 600 * 16
    assert len(u) >= n
    orc = O.OracleMF(X, **kw)
    orc.replay(u[:n], i[:n], rating=r[:n])
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors, initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    one = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors, initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    for a, b in ((0, 300 * 16), (300 * 16, 301 * 16), (301 * 16, n)):
        dev.replay_samples(u[a:b], i[a:b], rating=r[a:b])
    one.replay_samples(u[:n], i[:n], rating=r[:n])
    for m in (dev, one):
        assert_factor_parity(m.get_USER_factors(), orc.get_USER_factors(), mode, "U")
        assert_factor_parity(m.get_ITEM_factors(), orc.get_ITEM_factors(), mode, "V")
        assert_factor_parity(m.get_USER_bias(), orc.get_USER_bias(), mode, "bu")
        assert_factor_parity(m.get_ITEM_bias(), orc.get_ITEM_bias(), mode, "bi")
        assert abs(float(m.get_GLOBAL_bias()) - float(orc.get_GLOBAL_bias())) < 1e-4 * max(1e-3, abs(float(orc.get_GLOBAL_bias())))
    dev.close(); one.close()


def test_baseline_config_2_ml1m_k64_replay(gpu):
    """BASELINE config 2 at full ML-1M shape: k=64, batch 1000, sgd; 5 replayed epochs."""
    X = named_urm("ml1m", "binary")
    kw = dict(n_factors=64, algorithm_name="MF_BPR", batch_size=1000, random_seed=42, sgd_mode="sgd", learning_rate=1e-3,
              init_std_dev=0.1)
    _replay_case(X, kw, epochs=5)


def _assert_valid_bpr_stream(X, u, i, j):
    indptr, indices = X.indptr, X.indices
    assert ((u >= 0) & (u < X.shape[0])).all() and ((j >= 0) & (j < X.shape[1])).all()
    seen = set(zip(np.repeat(np.arange(X.shape[0]), np.diff(indptr)).tolist(), indices.tolist()))
    pairs_pos = list(zip(u.tolist(), i.tolist())); pairs_neg = list(zip(u.tolist(), j.tolist()))
    assert all(p in seen for p in pairs_pos), "positive item not in the user's profile"
    assert not any(p in seen for p in pairs_neg), "negative item is in the user's profile"


def test_native_sampler_is_a_valid_reference_sampler(gpu):
    X = synthetic_urm(2000, 300, 60000, 1, 299, seed=3, values="binary")
    # a user with every item and a user with none must never be drawn (.pyx:950-958)
    X = X.tolil(); X[0, :] = 1.0; X[1, :] = 0.0; X = X.tocsr(); X.sort_indices()
    dev = MatrixFactorization_MI355X_Epoch(X, n_factors=16, algorithm_name="MF_BPR", batch_size=500, random_seed=77, learning_rate=0.05)
    dev.epochIteration_Cython()
    u, i, j = dev.last_epoch_samples()
    assert len(u) == (X.shape[0] // 500 + 1) * 500                 # .pyx:583
    _assert_valid_bpr_stream(X, u, i, j)
    assert 0 not in set(u.tolist()) and 1 not in set(u.tolist())
    # users uniform among eligible ones: chi-square on 20 buckets
    counts = np.bincount(u // 100, minlength=20).astype(float)
    expected = len(u) / 20.0
    assert ((counts - expected) ** 2 / expected).sum() < 60
    # different epochs draw different streams; same seed reproduces the stream
    dev.epochIteration_Cython()
    u2, _, _ = dev.last_epoch_samples()
    assert (u2 != u).any()
    again = MatrixFactorization_MI355X_Epoch(X, n_factors=16, algorithm_name="MF_BPR", batch_size=500, random_seed=77, learning_rate=0.05)
    again.epochIteration_Cython()
    np.testing.assert_array_equal(again.last_epoch_samples()[0], u)


def test_sampler_grid_larger_than_the_device_draws_one_epoch(gpu):
    """the sampler read the epoch counter that its own launch advanced -- workgroups dispatched after workgroup 0 had
    retired drew from the NEXT epoch (a launch of 78 k workgroups is not resident at once).  The counter is advanced by a kernel of
    its own now.  FunkSVD at the ML-20M shape (20 M samples per epoch): two handles with one seed draw the same stream, the stream
    of epoch 2 differs from epoch 1 everywhere it should, and a handle that runs epochs 1 and 2 in ONE call ends on epoch 2's
    stream."""
    X = named_urm("ml20m", "real")
    kw = dict(n_factors=8, algorithm_name="FUNK_SVD", batch_size=1000, random_seed=123, sgd_mode="sgd", learning_rate=1e-3,
              use_bias=False, negative_interactions_quota=0.0)
    a = MatrixFactorization_MI355X_Epoch(X, **kw)
    b = MatrixFactorization_MI355X_Epoch(X, **kw)
    a.epochIteration_Cython()
    u1, i1, r1 = a.last_epoch_samples()
    b.epochIteration_Cython()
    for x, y in zip((u1, i1, r1), b.last_epoch_samples()):
        np.testing.assert_array_equal(x, y)
    a.epochIteration_Cython()
    u2, i2, r2 = a.last_epoch_samples()
    assert (u2 != u1).mean() > 0.99                                     # (two independent uniform draws of 138 493 users)
    c = MatrixFactorization_MI355X_Epoch(X, **kw)
    c.epochIteration_Cython(2)
    for x, y in zip((u2, i2, r2), c.last_epoch_samples()):
        np.testing.assert_array_equal(x, y)
    for m in (a, b, c):
        m.close()


@pytest.mark.parametrize("algorithm", ["MF_BPR", "FUNK_SVD"])
def test_native_epoch_equals_oracle_on_the_device_stream(gpu, algorithm):
    X = named_urm("ml1m", "real" if algorithm == "FUNK_SVD" else "binary", scale=0.2)
    kw = dict(n_factors=48, algorithm_name=algorithm, batch_size=200, random_seed=31, sgd_mode="adagrad", learning_rate=0.05,
              user_reg=0.01, positive_reg=0.01, negative_reg=0.01)
    if algorithm == "FUNK_SVD":
        kw.update(use_bias=True, bias_reg=0.01, negative_interactions_quota=0.4, batch_size=2000)
    orc = O.OracleMF(X, **kw)
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    dev.epochIteration_Cython()
    u, i, third = dev.last_epoch_samples()
    if algorithm == "MF_BPR":
        orc.replay(u, i, j=third)
    else:
        # positives carry the stored rating, negatives 0 and are absent from the profile (.pyx:900-931)
        dense = X.toarray()
        np.testing.assert_array_equal(dense[u, i], third)
        frac_pos = (third != 0).mean()
        assert abs(frac_pos - 0.4) < 0.03           # quota is the probability of a POSITIVE (sic, .pyx:898)
        orc.replay(u, i, rating=third.astype(np.float64))
    assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), "adagrad", "U")
    assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), "adagrad", "V")


def test_recommender_fit_surface(gpu):
    X = named_urm("ml1m", "binary", scale=0.15)
    rec = MatrixFactorization_BPR_MI355X(X, verbose=False)
    rec.fit(epochs=30, batch_size=200, num_factors=16, learning_rate=0.1, sgd_mode="adagrad", random_seed=1)
    assert rec.USER_factors.shape == (X.shape[0], 16) and rec.ITEM_factors.shape == (X.shape[1], 16)
    assert rec.use_bias is False
    # training must have learned to rank seen items above unseen ones (AUC-like check on the training data)
    scores = rec._compute_item_score(np.arange(100))
    dense = X[:100].toarray() > 0
    pos = np.array([scores[r][dense[r]].mean() for r in range(100)]); neg = np.array([scores[r][~dense[r]].mean() for r in range(100)])
    assert (pos > neg).mean() > 0.9
    ranked = rec.recommend(np.arange(10), cutoff=5)
    assert all(len(r) == 5 for r in ranked)
    f = MatrixFactorization_FunkSVD_MI355X(named_urm("ml1m", "real", scale=0.1), verbose=False)
    f.fit(epochs=2, batch_size=500, num_factors=8, learning_rate=0.01, use_bias=True, random_seed=2)
    assert f.USER_bias.shape == (f.n_users,) and np.isfinite(f.ITEM_factors).all()


def test_headline_ml20m_k128_replay_vs_oracle(gpu):
    """BASELINE headline config, exactly: 138 493 x 26 744, k=128, batch 1000, sgd.  One full reference epoch (139 mini-batches,
    glibc rand() stream of the oracle) replayed on the device; then two more NATIVE epochs (graph replay, device sampler) whose
    streams are replayed through the oracle."""
    X = named_urm("ml20m", "binary")
    kw = dict(n_factors=128, algorithm_name="MF_BPR", batch_size=1000, random_seed=42, sgd_mode="sgd", learning_rate=0.05)
    orc = O.OracleMF(X, **kw)
    orc.record_samples(10 ** 6)
    orc.epochIteration_Cython()
    u, i, j, _ = orc.recorded()
    assert len(u) == 139000
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    dev.replay_samples(u, i, neg_item=j)
    assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), "sgd", "U")
    assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), "sgd", "V")
    for _ in range(2):
        dev.epochIteration_Cython()
        du, di, dj = dev.last_epoch_samples()
        orc.replay(du, di, j=dj)
    assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), "sgd", "U")
    assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), "sgd", "V")
    dev.close()


@pytest.mark.parametrize("mode", ["sgd", "adam"])
def test_funksvd_ml20m_k128_replay_vs_oracle(gpu, mode):
    """FunkSVD at the ML-20M shape, k=128, batch 1000, biases on: the first 300 mini-batches of a reference epoch."""
    X = named_urm("ml20m", "real")
    kw = dict(n_factors=128, algorithm_name="FUNK_SVD", batch_size=1000, random_seed=7, sgd_mode=mode, learning_rate=0.005,
              user_reg=0.01, item_reg=0.01, bias_reg=0.01, use_bias=True, negative_interactions_quota=0.3)
    orc = O.OracleMF(X, **kw)
    n = 300 * 1000
    rng = np.random.default_rng(5)
    # a valid FunkSVD stream without running the 20 M-sample reference epoch: stored positives and zero-rated other items
    rows = np.repeat(np.arange(X.shape[0]), np.diff(X.indptr))
    pick = rng.integers(0, X.nnz, n)
    u = rows[pick].astype(np.int32); i = X.indices[pick].astype(np.int32); r = X.data[pick].astype(np.float64)
    neg = rng.random(n) < 0.3
    i[neg] = rng.integers(0, X.shape[1], neg.sum()); r[neg] = 0.0
    orc.replay(u, i, rating=r)
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    dev.replay_samples(u[:100000], i[:100000], rating=r[:100000])      # two calls: state carries over
    dev.replay_samples(u[100000:], i[100000:], rating=r[100000:])
    assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), mode, "U")
    assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), mode, "V")
    assert_factor_parity(dev.get_USER_bias(), orc.get_USER_bias(), mode, "bu")
    assert_factor_parity(dev.get_ITEM_bias(), orc.get_ITEM_bias(), mode, "bi")
    assert abs(float(dev.get_GLOBAL_bias()) - float(orc.get_GLOBAL_bias())) < 1e-5 * max(1.0, abs(float(orc.get_GLOBAL_bias())))
    dev.close()


@pytest.mark.parametrize("mode", MODES)
def test_native_graph_epochs_equal_oracle(gpu, mode):
    """Several native epochs in ONE call (hipGraph replays): row versions, Adam's t and the global-bias ring carry over."""
    X = named_urm("ml1m", "real", scale=0.2)
    kw = dict(n_factors=40, algorithm_name="FUNK_SVD", batch_size=128, random_seed=3, sgd_mode=mode, learning_rate=0.01,
              user_reg=0.01, bias_reg=0.01, use_bias=True, negative_interactions_quota=0.25)
    orc = O.OracleMF(X, **kw)
    a = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                         initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    b = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                         initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    a.epochIteration_Cython(3)                       # one call, three epochs
    for _ in range(3):                               # three calls: the same streams (same seed, same epoch counter)
        b.epochIteration_Cython()
        u, i, r = b.last_epoch_samples()
        orc.replay(u, i, rating=r.astype(np.float64))
    for dev in (a, b):
        assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), mode, "U")
        assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), mode, "V")
        assert_factor_parity(dev.get_ITEM_bias(), orc.get_ITEM_bias(), mode, "bi")
        assert abs(float(dev.get_GLOBAL_bias()) - float(orc.get_GLOBAL_bias())) < 1e-5 * max(1.0, abs(float(orc.get_GLOBAL_bias())))


@pytest.mark.parametrize("algorithm", ["MF_BPR", "FUNK_SVD"])
def test_schedule_paths_agree(gpu, algorithm, monkeypatch):
    """The in-LDS schedule (with wide tasks for long lists) and the general radix-sort schedule build the same tasks."""
    X = named_urm("ml1m", "real", scale=0.3)
    kw = dict(n_factors=64, algorithm_name=algorithm, batch_size=1500, random_seed=8, sgd_mode="sgd", learning_rate=0.02,
              user_reg=0.01, positive_reg=0.01, negative_reg=0.02, use_bias=algorithm == "FUNK_SVD", bias_reg=0.01)
    out = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("MI355REC_MF_GENERAL_SCHEDULE", "1")
        dev = MatrixFactorization_MI355X_Epoch(X, **kw)
        dev.epochIteration_Cython(3)
        out.append((dev.get_USER_factors(), dev.get_ITEM_factors(), dev.last_epoch_samples()))
        dev.close()
    np.testing.assert_array_equal(out[0][2][0], out[1][2][0])
    for a, b in zip(out[0][:2], out[1][:2]):
        assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max()
    counts = np.bincount(out[0][2][1][:1500])
    assert counts.max() > 8, "the case must contain a list long enough to be split over a workgroup (more than 2 rounds of 4)"


def test_fp32_state_is_available_for_adaptive_modes(gpu):
    """precision="fp32" keeps float32 factors and moments for the adaptive optimisers (half the memory); it then only
    tracks the float64 reference statistically -- this is the round-1 behaviour, kept as an option, not the default."""
    X = named_urm("ml1m", "binary", scale=0.2)
    kw = dict(n_factors=64, algorithm_name="MF_BPR", batch_size=100, random_seed=11, sgd_mode="adagrad", learning_rate=0.05)
    orc = O.OracleMF(X, **kw)
    orc.record_samples(10 ** 6); orc.epochIteration_Cython()
    u, i, j, _ = orc.recorded()
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, precision="fp32", **kw)
    dev.replay_samples(u, i, neg_item=j)
    err = np.abs(dev.get_ITEM_factors() - orc.get_ITEM_factors()) / np.abs(orc.get_ITEM_factors()).max()
    assert np.median(err) < 1e-6 and err.max() < 0.1
    with pytest.raises(ValueError):
        MatrixFactorization_MI355X_Epoch(X, precision="fp16", **kw)


def test_full_size_ml20m_k128_properties(gpu):
    """BASELINE headline shape (138k x 27k, k=128, batch 1000).  Size-independent properties: with lr = 0 nothing
    moves; with lr > 0 only sampled rows move, and the batch update is linear in the learning rate for sgd."""
    X = named_urm("ml20m", "binary")
    base = dict(n_factors=128, algorithm_name="MF_BPR", batch_size=1000, random_seed=9, sgd_mode="sgd")
    frozen = MatrixFactorization_MI355X_Epoch(X, learning_rate=0.0, **base)
    U0, V0 = frozen.get_factors()
    frozen.epochIteration_Cython()
    U1, V1 = frozen.get_factors()
    np.testing.assert_array_equal(U0, U1); np.testing.assert_array_equal(V0, V1)
    u, i, j = frozen.last_epoch_samples()
    # first mini-batch only: update is linear in lr (same seed => same samples, same start-of-batch factors)
    # (learning rates large enough for the step to dwarf one float32 ulp of the factors)
    a = MatrixFactorization_MI355X_Epoch(X, learning_rate=20.0, **base); a.replay_samples(u[:1000], i[:1000], neg_item=j[:1000])
    b = MatrixFactorization_MI355X_Epoch(X, learning_rate=40.0, **base); b.replay_samples(u[:1000], i[:1000], neg_item=j[:1000])
    Ua, Va = a.get_factors(); Ub, Vb = b.get_factors()
    moved = np.unique(u[:1000])
    untouched = np.setdiff1d(np.arange(X.shape[0]), moved)
    np.testing.assert_array_equal(Ua[untouched], U0[untouched])
    dA, dB = (Ua - U0)[moved], (Ub - U0)[moved]
    assert np.abs(dB - 2 * dA).max() <= 1e-4 * np.abs(dB).max()
    st = a.stats()
    assert st["n_units"] == 1000


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [5, 64, 130])
def test_asysvd_replay_parity(gpu, mode, k):
    """AsySVD (MatrixFactorization_Cython_Epoch.pyx:393-541): nnz + 1 strictly ordered single-sample steps."""
    X = named_urm("ml1m", "real", scale=0.05)                 # 302 x 185, ~3 k interactions per epoch
    kw = dict(n_factors=k, algorithm_name="ASY_SVD", batch_size=1, random_seed=17, sgd_mode=mode, learning_rate=0.005,
              user_reg=0.01, item_reg=0.02, bias_reg=0.02, use_bias=True, negative_interactions_quota=0.4)
    orc = O.OracleMF(X, **kw)
    orc.record_samples(10 ** 6)
    orc.epochIteration_Cython(); orc.epochIteration_Cython()
    u, i, _, r = orc.recorded()
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    assert dev.get_USER_factors().shape == (X.shape[1], k)       # the "user" matrix is item-sized
    dev.replay_samples(u, i, rating=r)
    assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), mode, "Y")
    assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), mode, "X")
    assert_factor_parity(dev.get_ITEM_bias(), orc.get_ITEM_bias(), mode, "bi")
    assert dev.stats()["n_units"] == len(u) == 2 * (X.nnz + 1)
    dev.close()


def test_asysvd_native_epoch_and_recommender(gpu):
    from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_AsySVD_MI355X
    X = named_urm("ml1m", "real", scale=0.05)
    kw = dict(n_factors=12, algorithm_name="ASY_SVD", batch_size=1, random_seed=4, sgd_mode="sgd", learning_rate=0.005,
              negative_interactions_quota=0.0)
    orc = O.OracleMF(X, **kw)
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    dev.epochIteration_Cython()
    u, i, r = dev.last_epoch_samples()
    assert len(u) == X.nnz + 1
    np.testing.assert_array_equal(X.toarray()[u, i], r)           # quota 0 => every sample is a stored interaction
    orc.replay(u, i, rating=r.astype(np.float64))
    assert rel_err(dev.get_USER_factors(), orc.get_USER_factors()) < RTOL
    assert rel_err(dev.get_ITEM_factors(), orc.get_ITEM_factors()) < RTOL
    rec = MatrixFactorization_AsySVD_MI355X(X, verbose=False)
    rec.fit(epochs=2, num_factors=8, learning_rate=0.005, use_bias=True, random_seed=1, batch_size=64)
    assert rec.ITEM_factors_Y.shape == (X.shape[1], 8) and rec.USER_factors.shape == (X.shape[0], 8)
    assert len(rec.recommend(3, cutoff=5)) == 5
    with pytest.raises(AssertionError):
        MatrixFactorization_MI355X_Epoch(X, n_factors=4, algorithm_name="ASY_SVD", batch_size=2)



# ---- fused sample tasks and replica-batched launches ------------------------------------------------------------------------
@pytest.mark.parametrize("mode,k", [("sgd", 64), ("sgd", 128), ("adam", 32), ("sgd", 256)])
def test_fused_sample_tasks_equal_one_task_per_row(gpu, mode, k, monkeypatch):
    """A single-sample user task that also updates the sample's once-touched item rows (mf_sched_sort_kernel) must leave
    exactly the factors the one-task-per-row schedule leaves: same arithmetic per row, bit for bit."""
    X = named_urm("ml1m", "binary", scale=0.3)
    kw = dict(n_factors=k, algorithm_name="MF_BPR", batch_size=300, random_seed=19, sgd_mode=mode, learning_rate=0.05,
              user_reg=0.01, positive_reg=0.02, negative_reg=0.03)
    out = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("MI355REC_MF_NO_FUSE", "1")
        dev = MatrixFactorization_MI355X_Epoch(X, **kw)
        dev.epochIteration_Cython(3)
        out.append((dev.get_USER_factors(), dev.get_ITEM_factors(), dev.last_epoch_samples()))
        dev.close()
    np.testing.assert_array_equal(out[0][2][0], out[1][2][0])
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    # the case must contain all kinds of samples: both item rows taken over, one of them, none (shared rows)
    u, i, j = out[0][2]
    b0 = slice(0, 300)
    ci, cj = np.bincount(np.concatenate([i[b0], j[b0]]), minlength=X.shape[1])[i[b0]], np.bincount(np.concatenate([i[b0], j[b0]]), minlength=X.shape[1])[j[b0]]
    assert ((ci == 1) & (cj == 1)).any() and ((ci > 1) & (cj == 1)).any() and ((ci > 1) | (cj > 1)).any()


def _group_case(X, kws, epochs, exact=True):
    solo = [MatrixFactorization_MI355X_Epoch(X, **kw) for kw in kws]
    members = [MatrixFactorization_MI355X_Epoch(X, **kw) for kw in kws]
    group = MatrixFactorization_MI355X_Group(members)
    group.epochIteration_Cython(1)
    group.epochIteration_Cython(epochs - 1)          # two calls: the second replays the captured graph
    gst = group.stats()
    total = 0
    for a, b, kw in zip(solo, members, kws):
        a.epochIteration_Cython(epochs)
        for x, y in zip(a.last_epoch_samples(), b.last_epoch_samples()):
            np.testing.assert_array_equal(x, y)
        Ua, Va = a.get_factors(); Ub, Vb = b.get_factors()
        if exact:
            np.testing.assert_array_equal(Ua, Ub); np.testing.assert_array_equal(Va, Vb)
        else:
            assert rel_err(Ub, Ua) < 1e-6 and rel_err(Vb, Va) < 1e-6
        if kw.get("use_bias"):
            assert rel_err(b.get_ITEM_bias(), a.get_ITEM_bias()) < 1e-6
            assert abs(float(a.get_GLOBAL_bias()) - float(b.get_GLOBAL_bias())) < 1e-6 * max(1.0, abs(float(a.get_GLOBAL_bias())))
        assert b.stats()["n_units"] == (epochs - 1) * len(b.last_epoch_samples()[0])
        total += b.stats()["n_units"]
    assert gst["n_units"] == total
    # members remain ordinary handles: one more epoch on their own equals one more epoch of the solo twin
    solo[0].epochIteration_Cython(); members[0].epochIteration_Cython()
    np.testing.assert_array_equal(solo[0].get_ITEM_factors(), members[0].get_ITEM_factors()) if exact else None
    group.close()
    for m in solo + members:
        m.close()


@pytest.mark.parametrize("mode", ["sgd", "adagrad", "adam"])
def test_group_members_end_bit_identical_to_training_alone(gpu, mode):
    """mi355rec_mf_group_*: mini-batch b of R models in one launch; models differ in k (inside one kernel instance), learning
    rate, regularisation, seed -- what run_parameter_search.py varies -- and each must end exactly where it ends alone."""
    X = named_urm("ml1m", "binary", scale=0.25)
    ks = [36, 48, 64, 40, 64] if mode == "sgd" else [20, 32, 24, 32, 18]            # float32: 4 | k <= 64; float64: 2 | k <= 32
    kws = [dict(n_factors=k, algorithm_name="MF_BPR", batch_size=200, random_seed=50 + n, sgd_mode=mode,
                learning_rate=0.01 * (n + 1), user_reg=0.001 * n, positive_reg=0.002, negative_reg=0.001 * (5 - n))
           for n, k in enumerate(ks)]
    _group_case(X, kws, epochs=3)


def test_group_funksvd_with_biases(gpu):
    X = named_urm("ml1m", "real", scale=0.12)
    kws = [dict(n_factors=32, algorithm_name="FUNK_SVD", batch_size=256, random_seed=70 + n, sgd_mode="sgd", learning_rate=0.005 * (n + 1),
                user_reg=0.01, item_reg=0.01, bias_reg=0.01 * n, use_bias=True, negative_interactions_quota=0.3) for n in range(3)]
    _group_case(X, kws, epochs=2, exact=False)         # the global bias is summed with atomics: last bits depend on arrival order


def test_group_at_headline_shape_and_rejections(gpu):
    X = named_urm("ml20m", "binary")
    rng = np.random.default_rng(1)
    U0 = rng.normal(0, 0.1, (X.shape[0], 128)).astype(np.float32); V0 = rng.normal(0, 0.1, (X.shape[1], 128)).astype(np.float32)
    kws = [dict(n_factors=128, algorithm_name="MF_BPR", batch_size=1000, random_seed=42 + n, sgd_mode="sgd", learning_rate=0.05,
                initial_USER_factors=U0, initial_ITEM_factors=V0) for n in range(4)]
    _group_case(X, kws, epochs=2)
    a = MatrixFactorization_MI355X_Epoch(X, n_factors=128, algorithm_name="MF_BPR", batch_size=1000, random_seed=1,
                                         initial_USER_factors=U0, initial_ITEM_factors=V0)
    b = MatrixFactorization_MI355X_Epoch(X, n_factors=128, algorithm_name="MF_BPR", batch_size=500, random_seed=1,
                                         initial_USER_factors=U0, initial_ITEM_factors=V0)
    c = MatrixFactorization_MI355X_Epoch(X, n_factors=12, algorithm_name="MF_BPR", batch_size=1000, random_seed=1)
    with pytest.raises(ValueError):
        MatrixFactorization_MI355X_Group([a, b])            # different batch size
    with pytest.raises(ValueError):
        MatrixFactorization_MI355X_Group([a, c])            # different kernel instance
    with pytest.raises(ValueError):
        MatrixFactorization_MI355X_Group([a, a])
    a2 = MatrixFactorization_MI355X_Epoch(X, n_factors=128, algorithm_name="MF_BPR", batch_size=1000, random_seed=2,
                                          initial_USER_factors=U0, initial_ITEM_factors=V0)
    g = MatrixFactorization_MI355X_Group([a, a2])
    a2.close()
    with pytest.raises(RuntimeError, match="member 1 was closed"):
        g.epochIteration_Cython()                       # (a dangling native handle otherwise)
    g.close()


def test_asysvd_full_ml1m_shape_k64_replay(gpu):
    """AsySVD at the full ML-1M shape (6 040 x 3 706, 1 000 209 interactions), k = 64, biases on: one whole reference epoch
    (nnz + 1 strictly ordered steps, each rewriting every Y row of the sampled user's profile) replayed against the oracle."""
    X = named_urm("ml1m", "real")
    kw = dict(n_factors=64, algorithm_name="ASY_SVD", batch_size=1, random_seed=23, sgd_mode="sgd", learning_rate=0.002,
              user_reg=0.01, item_reg=0.01, bias_reg=0.01, use_bias=True, negative_interactions_quota=0.2)
    orc = O.OracleMF(X, **kw)
    orc.record_samples(2 * 10 ** 6)
    orc.epochIteration_Cython()
    u, i, _, r = orc.recorded()
    assert len(u) == X.nnz + 1
    dev = MatrixFactorization_MI355X_Epoch(X, initial_USER_factors=orc.initial_USER_factors,
                                           initial_ITEM_factors=orc.initial_ITEM_factors, **kw)
    dev.replay_samples(u, i, rating=r)
    assert_factor_parity(dev.get_USER_factors(), orc.get_USER_factors(), "sgd", "Y")
    assert_factor_parity(dev.get_ITEM_factors(), orc.get_ITEM_factors(), "sgd", "X")
    assert_factor_parity(dev.get_ITEM_bias(), orc.get_ITEM_bias(), "sgd", "bi")
    assert_factor_parity(dev.get_USER_bias(), orc.get_USER_bias(), "sgd", "bu")
    print("asysvd ml1m k64: %d steps in %.2f s on the device" % (len(u), dev.stats()["call_ms"] * 1e-3))
    dev.close()
