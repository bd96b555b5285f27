"""Aggregate SLIM-BPR throughput of R independent models training side by side on ONE GPU (one handle, stream and host thread each;
BASELINE config 3: ML-20M shape, adagrad).  The reference parallelises every SGD path by independent models
(ParameterTuning/run_parameter_search.py:498-503).  Dense store: the compute units are leased R ways (MI355REC_SLIM_CUS), so that
every model's owned rows stay resident.  Usage: slim_replicas.py [dense|symmetric] 1 2 4 8"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys2019_deeplearning_evaluation_amd import SLIM_BPR_MI355X_Epoch
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm

store = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("dense", "symmetric") else "dense"
counts = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8]
X = named_urm("ml20m", "binary")
epochs = 6
for R in counts:
    os.environ["MI355REC_SLIM_CUS"] = str(max(32, 256 // R))
    reps = [SLIM_BPR_MI355X_Epoch(X, symmetric=store == "symmetric", topK=100, learning_rate=1e-4, sgd_mode="adagrad", random_seed=100 + r)
            for r in range(R)]
    for m in reps:
        m.epochIteration_Cython(1)
    threads = [threading.Thread(target=m.epochIteration_Cython, args=(epochs,)) for m in reps]
    t0 = time.perf_counter()
    for t in threads: t.start()
    for t in threads: t.join()
    wall = time.perf_counter() - t0
    rate = R * epochs * (X.shape[0] + 1) / wall
    info = [m.schedule_info()[0] for m in reps] if store == "dense" else []
    print("%s store, %2d models: %.2f M samples/s aggregate (%.2f ms per model epoch incl. its schedule) owned rows %s" % (
        store, R, rate / 1e6, wall / epochs * 1e3, info), flush=True)
    for m in reps: m.close()
