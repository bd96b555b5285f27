#!/usr/bin/env python3
"""One short invocation of ONE hot path at the ML-20M shape, for rocprofv3 (scripts/pmc_round.sh).  Usage: run_path.py <path>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355REC_NO_GRAPH", "1")       # rocprofv3 (ROCm 7.2) crashes while tracing hipGraph replays
import numpy as np  # noqa: E402

from bench import load_urm, K_FACTORS, BATCH, TOPK  # noqa: E402
from recsys2019_deeplearning_evaluation_amd import (Compute_Similarity_MI355X, IALS_MI355X_Epoch, MatrixFactorization_MI355X_Epoch,  # noqa: E402
                                                    MI355XScorer, SLIM_BPR_MI355X_Epoch)

path = sys.argv[1]
urm = load_urm("ml20m")
if path == "mf":
    m = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3, sgd_mode="sgd", random_seed=1)
    m.epochIteration_Cython(2)
elif path == "mf_group":
    from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Group
    rng = np.random.default_rng(0)
    U0 = rng.normal(0, 0.1, (urm.shape[0], K_FACTORS)).astype(np.float32); V0 = rng.normal(0, 0.1, (urm.shape[1], K_FACTORS)).astype(np.float32)
    members = [MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3, sgd_mode="sgd",
                                                random_seed=200 + r, initial_USER_factors=U0, initial_ITEM_factors=V0) for r in range(32)]
    g = MatrixFactorization_MI355X_Group(members)
    g.epochIteration_Cython(2)
elif path == "asy":
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    x1m = named_urm("ml1m", "real")
    m = MatrixFactorization_MI355X_Epoch(x1m, n_factors=64, algorithm_name="ASY_SVD", batch_size=1, learning_rate=1e-3, sgd_mode="sgd", use_bias=True,
                                         negative_interactions_quota=0.0, random_seed=1)
    rng = np.random.default_rng(0)
    rows = np.repeat(np.arange(x1m.shape[0]), np.diff(x1m.indptr))
    pick = rng.integers(0, x1m.nnz, 131072)
    m.replay_samples(rows[pick].astype(np.int32), x1m.indices[pick].astype(np.int32), rating=x1m.data[pick].astype(np.float32))
elif path == "funk":
    # the bench workload: ONE native epoch (20 001 mini-batches of on-device samples, in-LDS schedule 256 at a time, global-bias ring)
    m = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="FUNK_SVD", batch_size=BATCH, learning_rate=1e-3, sgd_mode="sgd", use_bias=True,
                                         negative_interactions_quota=0.0, random_seed=1)
    m.epochIteration_Cython(1)
elif path == "sim":
    s = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
    s.compute_slabs()
elif path in ("slim_dense", "slim_symmetric"):
    s = SLIM_BPR_MI355X_Epoch(urm, symmetric=path == "slim_symmetric", sgd_mode="adagrad", learning_rate=1e-4, topK=TOPK, random_seed=7)
    s.epochIteration_Cython(1)
elif path == "ials":
    conf = urm.copy(); conf.data = (1.0 + conf.data).astype(np.float32)
    k = 200
    ia = IALS_MI355X_Epoch(conf, k, 1e-3, k ** -0.5 * np.random.default_rng(0).random((urm.shape[1], k)))
    ia.run_epochs(1)
elif path == "score":
    rng = np.random.default_rng(0)
    sc = MI355XScorer(rng.normal(0, 0.1, (urm.shape[0], K_FACTORS)).astype(np.float32), rng.normal(0, 0.1, (urm.shape[1], K_FACTORS)).astype(np.float32), urm)
    sc.recommend(rng.choice(urm.shape[0], 1000, replace=False).astype(np.int32), 20)
else:
    raise SystemExit("unknown path " + path)
# the path's own stream time (events around the call), for scripts/summarize_pmc.py: launches x average duration of a kernel
# cannot exceed it
for obj in ("m", "g", "s", "ia", "sc"):
    if obj in dir() and hasattr(globals()[obj], "stats"):
        st = globals()[obj].stats()
        print("path_call_ms=%.6f kernel_ms=%.6f n_launches=%d n_units=%d" % (st.get("call_ms", 0.0), st.get("kernel_ms", 0.0), st.get("n_launches", 0), st.get("n_units", 0)))
        break
print("done", path)
