"""FunkSVD epoch at the ML-20M shape (20 001 mini-batches of 1000): graph segments against plain launches (MI355REC_NO_GRAPH=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_urm, K_FACTORS, BATCH
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch
urm = load_urm("ml20m")
for no_graph in ("", "1"):
    if no_graph:
        os.environ["MI355REC_NO_GRAPH"] = "1"
    m = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, learning_rate=1e-3, init_std_dev=0.1, random_seed=7, algorithm_name="FUNK_SVD",
                                         batch_size=BATCH, sgd_mode="sgd", use_bias=True, negative_interactions_quota=0.0)
    for _ in range(4):
        t = time.perf_counter(); m.epochIteration_Cython(1); dt = time.perf_counter() - t
        st = m.stats()
        print("%s epoch %.1f ms wall, stream %.1f ms, %.1f M samples/s" % ("plain launches" if no_graph else "graph segments", dt * 1e3, st["call_ms"],
                                                                            st["n_units"] / st["call_ms"] / 1e3), flush=True)
    m.close()
