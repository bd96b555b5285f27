"""32 BPR-MF models in one launch per mini-batch at the headline shape: epoch time (diagnostics; run on the GPU box).
Environment switches of the library apply (MI355REC_MF_GROUP_OCC8=1, ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_urm, K_FACTORS, BATCH
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch, MatrixFactorization_MI355X_Group
urm = load_urm("ml20m")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)
U0 = rng.normal(0, 0.1, (urm.shape[0], K_FACTORS)).astype(np.float32)
V0 = rng.normal(0, 0.1, (urm.shape[1], K_FACTORS)).astype(np.float32)
members = [MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3, sgd_mode="sgd",
                                            random_seed=200 + r, initial_USER_factors=U0, initial_ITEM_factors=V0) for r in range(R)]
g = MatrixFactorization_MI355X_Group(members)
g.epochIteration_Cython(2)
n = 20
g.epochIteration_Cython(n)
st = g.stats()
samples = R * n * (urm.shape[0] // BATCH + 1) * BATCH
print("%d models: %.3f ms per epoch, %.1f M samples/s (%s)" % (R, st["call_ms"] / n, samples / st["call_ms"] / 1e3,
      " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("MI355REC_MF"))), flush=True)
