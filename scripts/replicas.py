"""Aggregate BPR-MF throughput of N concurrent replicas on ONE GPU (one stream each).  Usage: replicas.py 1 8 16   (more than 16 streams stall on this stack)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_urm, K_FACTORS, BATCH
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch
urm = load_urm("ml20m")
per_epoch = (urm.shape[0] // BATCH + 1) * BATCH
for n_rep in [int(a) for a in sys.argv[1:]]:
    epochs = 100
    reps = [MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3,
                                             sgd_mode="sgd", random_seed=100 + r) for r in range(n_rep)]
    for m in reps:
        m.epochIteration_Cython(2)
    threads = [threading.Thread(target=m.epochIteration_Cython, args=(epochs,)) for m in reps]
    t0 = time.perf_counter()
    for t in threads: t.start()
    for t in threads: t.join()
    wall = time.perf_counter() - t0
    rate = n_rep * epochs * per_epoch / wall
    print("%2d replicas: %.1f M samples/s aggregate, %.3f of HBM peak (24 k B/sample)" % (n_rep, rate / 1e6, rate * 24 * K_FACTORS / 8e12), flush=True)
    for m in reps: m.close()
