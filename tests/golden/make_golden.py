#!/usr/bin/env python3
"""Generates the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container (where /root/reference exists):   python tests/golden/make_golden.py
It imports the reference's compiled Cython kernels (built by oracle/build_ref.py into oracle/_ref/) and its
pure-Python IALSRecommender, runs them on small seeded inputs and stores inputs + outputs as .npz files.
The fixtures are what pins the oracle (tests/test_oracle_golden.py) and, through it, the HIP kernels on
machines where /root/reference does not exist (the GPU box).

Stored per case: the input matrix (CSR triplet), the keyword arguments (JSON) and the reference outputs.
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref_loader  # noqa: E402


def small_urm(n_users, n_items, density, seed, real):
    rng = np.random.default_rng(seed)
    X = sps.random(n_users, n_items, density, format="csr", random_state=np.random.RandomState(seed), dtype=np.float32)
    if real:
        X.data = (rng.integers(1, 6, X.nnz) + 1e-3 * rng.random(X.nnz)).astype(np.float32)
    else:
        X.data[:] = 1.0
    # no empty rows / columns
    X = X.tolil()
    for u in range(n_users):
        if X[u].nnz == 0:
            X[u, rng.integers(0, n_items)] = 1.0
    X = X.tocsc().tolil()
    Xc = X.tocsc()
    for i in np.flatnonzero(np.diff(Xc.indptr) == 0):
        X[rng.integers(0, n_users), i] = 1.0
    X = sps.csr_matrix(X, dtype=np.float32)
    X.sort_indices()
    return X


def pack_csr(prefix, X):
    return {prefix + "_indptr": X.indptr.astype(np.int32), prefix + "_indices": X.indices.astype(np.int32),
            prefix + "_data": X.data.astype(np.float32), prefix + "_shape": np.array(X.shape, dtype=np.int64)}


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def euclidean_row_weights():
    """Compute_Similarity_Euclidean with row_weights (Compute_Similarity_Euclidean.py:62-72, :153, :174-175): square inputs only
    (the distances to the n_cols columns are multiplied by the n_rows weights).  float32 and float64 weights (the latter promote
    the distance vector to float64).  `python make_golden.py euclidean_row_weights` writes this fixture alone."""
    EUC = ref_loader.load_python_reference("Base.Similarity.Compute_Similarity_Euclidean", "Compute_Similarity_Euclidean")
    X = small_urm(48, 48, 0.25, 23, real=True)
    X.data = np.round(X.data)
    out = pack_csr("X", X)
    rng = np.random.default_rng(23)
    out["w64"] = rng.uniform(0.25, 2.0, X.shape[0])
    out["w32"] = out["w64"].astype(np.float32)
    cases = []
    for mode in ("lin", "log", "exp"):
        for normalize, avg_row, shrink in ((False, False, 0), (True, True, 3)):
            for w in ("w32", "w64"):
                cases.append(dict(weights=w, kw=dict(similarity_from_distance_mode=mode, normalize=normalize, normalize_avg_row=avg_row,
                                                     shrink=shrink)))
    for n, case in enumerate(cases):
        kw = dict(case["kw"], row_weights=out[case["weights"]])
        out["dense_%d" % n] = quiet(lambda: EUC(X, topK=X.shape[1], **kw).compute_similarity()).toarray().astype(np.float32)
        out["top_%d" % n] = quiet(lambda: EUC(X, topK=5, **kw).compute_similarity()).toarray().astype(np.float32)
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "euclidean_row_weights.npz"), **out)


def main():
    assert build_ref.build(), "reference sources not available: run this where /root/reference exists"
    MF, SLIM, SIM = ref_loader.load("mf"), ref_loader.load("slim"), ref_loader.load("sim")

    # ---------------- similarity ----------------
    X = small_urm(90, 40, 0.12, 11, real=True)
    out = pack_csr("X", X)
    cases = []
    rw = np.random.default_rng(5).random(X.shape[0])
    for sim in ["cosine", "adjusted", "asymmetric", "pearson", "jaccard", "dice", "tversky"]:
        for shrink, normalize in [(0, True), (7, True), (3, False)]:
            kw = dict(shrink=shrink, normalize=normalize, similarity=sim, asymmetric_alpha=0.3, tversky_alpha=0.7, tversky_beta=1.3)
            cases.append(kw)
    for n, kw in enumerate(cases):
        dense = quiet(lambda: SIM(X, topK=0, **kw).compute_similarity())
        top = quiet(lambda: SIM(X, topK=6, **kw).compute_similarity())
        out["dense_%d" % n] = np.asarray(dense, dtype=np.float64)
        out["top6_%d" % n] = top.toarray().astype(np.float32)
    out["dense_rw"] = np.asarray(quiet(lambda: SIM(X, topK=0, shrink=2, row_weights=rw).compute_similarity()))
    out["row_weights"] = rw
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "similarity.npz"), **out)

    # ---------------- top-K selection rule: Compute_Similarity_Python (pure-Python reference) vs the Cython class ----------------
    # adjusted / pearson produce negative similarities.  Compute_Similarity_Python takes the K largest cells of the FULL column
    # (zeros compete, then are dropped, Compute_Similarity_Python.py:346-355); the Cython class partitions a zero-padded array
    # of the touched cells (.pyx:523-545), which picks padding entries (stale ids) as soon as fewer than K touched cells are
    # non-negative.  Both outputs are stored, on tie-free data, so the tests can state exactly where they differ.
    PYSIM = ref_loader.load_python_reference("Base.Similarity.Compute_Similarity_Python", "Compute_Similarity_Python")
    Xk = small_urm(120, 45, 0.1, 23, real=True)
    out = pack_csr("X", Xk)
    cases = [dict(topK=12, shrink=2, normalize=True, similarity=sim) for sim in ("cosine", "adjusted", "pearson", "asymmetric")]
    cases += [dict(topK=5, shrink=0, normalize=True, similarity="adjusted"), dict(topK=40, shrink=0, normalize=False, similarity="pearson")]
    for n, kw in enumerate(cases):
        out["python_%d" % n] = quiet(lambda: PYSIM(Xk, **kw).compute_similarity()).toarray().astype(np.float32)
        out["cython_%d" % n] = quiet(lambda: SIM(Xk, **kw).compute_similarity()).toarray().astype(np.float32)
        out["dense_%d" % n] = np.asarray(quiet(lambda: SIM(Xk, **dict(kw, topK=0)).compute_similarity()), dtype=np.float64)
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "similarity_topk_rules.npz"), **out)

    # ---------------- BPR-MF / FunkSVD ----------------
    Xb = small_urm(70, 50, 0.15, 12, real=False)
    Xr = small_urm(70, 50, 0.15, 13, real=True)
    out = dict(pack_csr("Xb", Xb), **pack_csr("Xr", Xr))
    cases = []
    for mode in ["sgd", "adagrad", "rmsprop", "adam"]:
        cases.append(dict(matrix="Xb", epochs=3, kw=dict(n_factors=12, algorithm_name="MF_BPR", batch_size=8, random_seed=101,
                          sgd_mode=mode, learning_rate=0.05, user_reg=0.01, positive_reg=0.02, negative_reg=0.03)))
        cases.append(dict(matrix="Xr", epochs=2, kw=dict(n_factors=12, algorithm_name="FUNK_SVD", batch_size=16, random_seed=202,
                          sgd_mode=mode, learning_rate=0.02, user_reg=0.01, item_reg=0.4, bias_reg=0.05, use_bias=True,
                          negative_interactions_quota=0.3)))
    cases.append(dict(matrix="Xr", epochs=2, kw=dict(n_factors=5, algorithm_name="FUNK_SVD", batch_size=1, random_seed=7,
                      sgd_mode="sgd", learning_rate=0.02, use_bias=False, negative_interactions_quota=0.0)))
    for mode in ["sgd", "adagrad", "rmsprop", "adam"]:      # AsySVD: batch_size 1, two item-sized matrices
        cases.append(dict(matrix="Xr", epochs=2, kw=dict(n_factors=7, algorithm_name="ASY_SVD", batch_size=1, random_seed=303,
                          sgd_mode=mode, learning_rate=0.01, user_reg=0.01, item_reg=0.02, bias_reg=0.03, use_bias=True,
                          negative_interactions_quota=0.3)))
    cases.append(dict(matrix="Xr", epochs=1, kw=dict(n_factors=70, algorithm_name="ASY_SVD", batch_size=1, random_seed=304,
                      sgd_mode="sgd", learning_rate=0.005, use_bias=False, negative_interactions_quota=0.0)))
    for n, case in enumerate(cases):
        Xc = Xb if case["matrix"] == "Xb" else Xr
        m = MF(Xc, **case["kw"])
        for _ in range(case["epochs"]):
            quiet(m.epochIteration_Cython)
        out["U_%d" % n] = m.get_USER_factors()
        out["V_%d" % n] = m.get_ITEM_factors()
        if case["kw"].get("use_bias"):
            out["bu_%d" % n] = m.get_USER_bias(); out["bi_%d" % n] = m.get_ITEM_bias(); out["mu_%d" % n] = m.get_GLOBAL_bias()
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "matrix_factorization.npz"), **out)

    # ---------------- SLIM-BPR ----------------
    Xs = small_urm(80, 30, 0.2, 14, real=False)
    out = pack_csr("X", Xs)
    cases = []
    for symmetric in [False, True]:
        for mode in ["sgd", "adagrad", "rmsprop", "adam"]:
            cases.append(dict(epochs=3, kw=dict(symmetric=symmetric, random_seed=303, sgd_mode=mode, learning_rate=0.05,
                                                li_reg=0.01, lj_reg=0.02)))
    for n, case in enumerate(cases):
        e = SLIM(Xs, topK=False, final_model_sparse_weights=False, **case["kw"])
        for _ in range(case["epochs"]):
            quiet(e.epochIteration_Cython)
        S = quiet(e.get_S)
        out["S_%d" % n] = S.toarray() if sps.issparse(S) else np.asarray(S)
        quiet(e._dealloc)
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "slim_bpr.npz"), **out)

    # ---------------- SLIM-BPR, sparse-tree store (train_with_sparse_weights=True) ----------------
    # 79 users -> 80 steps per epoch (a multiple of 5: four rebalance points), 81 users -> 82 steps (five: 16, 32, 48, 64, 80);
    # get_S is called in the middle of training too, because it prunes the model (SLIM_BPR_Cython_Epoch.pyx:381-382, 740-780)
    out = {}
    cases = []
    for n_users, topK, mode, regs in [(79, 4, "sgd", (0.0, 0.0)), (81, 4, "sgd", (0.0, 0.0)), (79, 6, "adam", (0.01, 0.02)),
                                      (81, 3, "adagrad", (0.0, 0.03)), (79, False, "rmsprop", (0.01, 0.0)), (81, 25, "sgd", (0.02, 0.0))]:
        cases.append(dict(n_users=n_users, epochs=[2, 1], kw=dict(topK=topK, random_seed=404, sgd_mode=mode, learning_rate=0.05,
                                                                 li_reg=regs[0], lj_reg=regs[1], train_with_sparse_weights=True)))
    urms = {nu: small_urm(nu, 30, 0.2, 16, real=False) for nu in (79, 81)}
    for nu, Xs in urms.items():
        out.update(pack_csr("X%d" % nu, Xs))
    for n, case in enumerate(cases):
        e = SLIM(urms[case["n_users"]], **case["kw"])
        for m, epochs in enumerate(case["epochs"]):
            for _ in range(epochs):
                quiet(e.epochIteration_Cython)
            S = quiet(e.get_S)
            out["indptr_%d_%d" % (n, m)] = S.indptr; out["indices_%d_%d" % (n, m)] = S.indices; out["data_%d_%d" % (n, m)] = S.data
        quiet(e._dealloc)
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "slim_bpr_sparse.npz"), **out)

    # ---------------- IALS (pure-Python reference) ----------------
    IALS = ref_loader.load_python_reference("MatrixFactorization.IALSRecommender", "IALSRecommender")
    Xi = small_urm(60, 45, 0.15, 15, real=True)
    out = pack_csr("X", Xi)
    cases = [dict(epochs=2, kw=dict(num_factors=8, confidence_scaling="linear", alpha=1.0, reg=1e-3)),
             dict(epochs=2, kw=dict(num_factors=16, confidence_scaling="log", alpha=3.0, epsilon=0.5, reg=0.05))]
    for n, case in enumerate(cases):
        np.random.seed(404 + n)
        rec = quiet(lambda: IALS(Xi, verbose=False))
        state = np.random.get_state()
        quiet(lambda: rec.fit(epochs=case["epochs"], **case["kw"]))
        np.random.set_state(state)
        k = case["kw"]["num_factors"]
        out["V0_%d" % n] = k ** -0.5 * np.random.random_sample((Xi.shape[1], k))   # same draw as _init_factors (:204-210)
        out["U_%d" % n] = rec.USER_factors
        out["V_%d" % n] = rec.ITEM_factors
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "ials.npz"), **out)
    # ---------------- BM25 / TF-IDF pre-weighting (pure-Python reference) ----------------
    bm25 = ref_loader.load_python_reference("Base.IR_feature_weighting", "okapi_BM_25")
    tfidf = ref_loader.load_python_reference("Base.IR_feature_weighting", "TF_IDF")
    Xw = small_urm(50, 35, 0.2, 16, real=True)
    out = pack_csr("X", Xw)
    out["bm25_T"] = bm25(Xw.astype(np.float32).T).T.toarray()
    out["tfidf_T"] = tfidf(Xw.astype(np.float32).T).T.toarray()
    out["cases"] = np.array(json.dumps([]))
    np.savez_compressed(os.path.join(HERE, "feature_weighting.npz"), **out)
    # ---------------- P3alpha / RP3beta (pure-Python reference) ----------------
    P3 = ref_loader.load_python_reference("GraphBased.P3alphaRecommender", "P3alphaRecommender")
    RP3 = ref_loader.load_python_reference("GraphBased.RP3betaRecommender", "RP3betaRecommender")
    Xg = small_urm(70, 45, 0.18, 17, real=True)
    out = pack_csr("X", Xg)
    cases = [dict(cls="P3", kw=dict(topK=8, alpha=1.0, normalize_similarity=False)),
             dict(cls="P3", kw=dict(topK=8, alpha=0.7, normalize_similarity=True, min_rating=2, implicit=False)),
             dict(cls="RP3", kw=dict(topK=8, alpha=1.0, beta=0.6, normalize_similarity=True)),
             dict(cls="RP3", kw=dict(topK=6, alpha=1.3, beta=0.3, normalize_similarity=False))]
    for n, case in enumerate(cases):
        rec = quiet(lambda: (P3 if case["cls"] == "P3" else RP3)(Xg.copy(), verbose=False))
        quiet(lambda: rec.fit(**case["kw"]))
        out["W_%d" % n] = rec.W_sparse.toarray().astype(np.float64)
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "graph_based.npz"), **out)
    # ---------------- EASE_R (pure-Python reference over the compiled similarity) ----------------
    EASE = ref_loader.load_python_reference("EASE_R.EASE_R_Recommender", "EASE_R_Recommender")
    Xs = small_urm(70, 45, 0.18, 21, real=True)
    out = pack_csr("X", Xs)
    cases = [dict(topK=None, l2_norm=50.0, normalize_matrix=False), dict(topK=7, l2_norm=10.0, normalize_matrix=True),
             dict(topK=None, l2_norm=1e3, normalize_matrix=True)]
    for n, kw in enumerate(cases):
        rec = quiet(lambda: EASE(Xs.copy()))
        quiet(lambda: rec.fit(verbose=False, **kw))
        W = rec.W_sparse
        out["W_%d" % n] = (W.toarray() if sps.issparse(W) else np.asarray(W)).astype(np.float32)
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "ease_r.npz"), **out)
    # ---------------- Euclidean similarity (pure-Python reference) ----------------
    EUC = ref_loader.load_python_reference("Base.Similarity.Compute_Similarity_Euclidean", "Compute_Similarity_Euclidean")
    Xe = small_urm(60, 40, 0.2, 18, real=True)
    Xe.data = np.round(Xe.data)          # integer ratings: the reference's float32 sums are exact, parity is 1e-6
    out = pack_csr("X", Xe)
    Xj = small_urm(60, 40, 0.2, 19, real=True)     # jittered ratings: float32 cancellation noise in the reference
    out.update(pack_csr("Xj", Xj))
    cases = []
    for mode in ("lin", "log", "exp"):
        for normalize, avg_row, shrink in ((False, False, 0), (True, False, 2), (False, True, 0), (True, True, 5)):
            cases.append(dict(similarity_from_distance_mode=mode, normalize=normalize, normalize_avg_row=avg_row, shrink=shrink))
    for n, kw in enumerate(cases):
        out["dense_%d" % n] = quiet(lambda: EUC(Xe, topK=Xe.shape[1], **kw).compute_similarity()).toarray().astype(np.float32)
        out["top_%d" % n] = quiet(lambda: EUC(Xe, topK=6, **kw).compute_similarity()).toarray().astype(np.float32)
    out["densej_0"] = quiet(lambda: EUC(Xj, topK=Xj.shape[1], **cases[0]).compute_similarity()).toarray().astype(np.float32)
    out["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "euclidean.npz"), **out)
    euclidean_row_weights()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    if sys.argv[1:] == ["euclidean_row_weights"]:
        euclidean_row_weights()
    else:
        main()
