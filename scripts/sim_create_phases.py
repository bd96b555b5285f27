"""ItemKNN constructor at the ML-20M shape: Python wall time of Compute_Similarity_MI355X(...) and, on stderr, the native
constructor's phases (MI355REC_SIM_CREATE_PHASES=1).  Usage: sim_create_phases.py [binary|real]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI355REC_SIM_CREATE_PHASES"] = "1"
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
kind = sys.argv[1] if len(sys.argv) > 1 else "binary"
X = named_urm("ml20m", kind)
for rep in range(3):
    print("--- constructor %d" % rep, file=sys.stderr, flush=True)
    t = time.perf_counter()
    sim = Compute_Similarity_MI355X(X, topK=100, shrink=0, normalize=True, similarity="cosine")
    sim.synchronize()
    dt = time.perf_counter() - t
    t = time.perf_counter()
    W = sim.compute_similarity()
    print("constructor %.3f ms (Python wall), build + CSR assembly on the host %.3f ms, kernel %.3f ms" % (dt * 1e3, (time.perf_counter() - t) * 1e3,
          sim.stats()["kernel_ms"]), flush=True)
    sim.close()
