"""BPR-MF k=128 B=1000 at ML-20M shape: samples/s of the plain epoch loop against MI355REC_MF_OVERLAP=1 (schedule of epoch e + 1 on a
second stream while the mini-batches of epoch e run), one model and a group of R models.  Usage: mf_overlap.py [R] [epochs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import load_urm, BATCH  # noqa: E402
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch, MatrixFactorization_MI355X_Group  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 0
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
urm = load_urm("ml20m")
k = 128
rng = np.random.default_rng(0)
U0 = rng.normal(0, 0.1, (urm.shape[0], k)).astype(np.float32)
V0 = rng.normal(0, 0.1, (urm.shape[1], k)).astype(np.float32)


def model(seed):
    return MatrixFactorization_MI355X_Epoch(urm, n_factors=k, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3, sgd_mode="sgd",
                                            random_seed=seed, initial_USER_factors=U0, initial_ITEM_factors=V0)


for overlap in ("", "require"):
    if overlap:
        os.environ["MI355REC_MF_OVERLAP"] = overlap
    else:
        os.environ.pop("MI355REC_MF_OVERLAP", None)
    m = model(7)
    m.epochIteration_Cython(8)
    best = 0.0
    for _ in range(3):
        m.epochIteration_Cython(epochs)
        st = m.stats()
        best = max(best, st["n_units"] / (st["call_ms"] * 1e-3))
    print("one model   overlap=%-8s %8.1f M samples/s  (%.3f ms per epoch)" % (overlap or "off", best / 1e6, (urm.shape[0] // BATCH + 1) * BATCH / best * 1e3), flush=True)
    m.close()
    if R:
        members = [model(100 + r) for r in range(R)]
        g = MatrixFactorization_MI355X_Group(members)
        g.epochIteration_Cython(8)
        best = 0.0
        for _ in range(2):
            g.epochIteration_Cython(epochs // 2)
            st = g.stats()
            best = max(best, st["n_units"] / (st["call_ms"] * 1e-3))
        print("%2d models   overlap=%-8s %8.1f M samples/s aggregate" % (R, overlap or "off", best / 1e6), flush=True)
        g.close()
        for mm in members:
            mm.close()
