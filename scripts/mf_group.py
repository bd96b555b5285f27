"""Replica-batched BPR-MF on ONE GPU: R independent models, mini-batch b of all of them in one launch (mi355rec_mf_group_*).
Usage: mf_group.py 1 8 32 [--k 128] [--epochs 30]     prints samples/s aggregate, per-launch duration, fraction of HBM peak."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import load_urm, BATCH  # noqa: E402
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch, MatrixFactorization_MI355X_Group  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sizes", nargs="+", type=int)
ap.add_argument("--k", type=int, default=128)
ap.add_argument("--epochs", type=int, default=30)
ap.add_argument("--mode", default="sgd")
ap.add_argument("--workload", default="ml20m")
args = ap.parse_args()

urm = load_urm(args.workload)
k = args.k
rng = np.random.default_rng(0)
U0 = rng.normal(0, 0.1, (urm.shape[0], k)).astype(np.float32)
V0 = rng.normal(0, 0.1, (urm.shape[1], k)).astype(np.float32)
per_epoch = (urm.shape[0] // BATCH + 1) * BATCH
nb = per_epoch // BATCH
for R in args.sizes:
    members = [MatrixFactorization_MI355X_Epoch(urm, n_factors=k, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3,
                                                sgd_mode=args.mode, random_seed=100 + r, initial_USER_factors=U0, initial_ITEM_factors=V0)
               for r in range(R)]
    g = MatrixFactorization_MI355X_Group(members)
    g.epochIteration_Cython(2)
    g.epochIteration_Cython(args.epochs)
    st = g.stats()
    rate = st["n_units"] / (st["call_ms"] * 1e-3)
    g.set_profiling(2 * nb)
    g.epochIteration_Cython(2)
    pst = g.stats()
    g.set_profiling(0)
    us = pst["kernel_ms"] / max(1, pst["n_timed"]) * 1e3
    alg = R * BATCH * 24.0 * k
    print("R=%3d  %8.1f M samples/s aggregate  whole-epoch %.3f of HBM peak | launch %.2f us -> %.0f GB/s algorithmic = %.3f of 8 TB/s | epoch %.3f ms" % (
        R, rate / 1e6, rate * 24 * k / 8e12, us, alg / (us * 1e-6) / 1e9, alg / (us * 1e-6) / 8e12, st["call_ms"] / args.epochs), flush=True)
    g.close()
    for m in members:
        m.close()
