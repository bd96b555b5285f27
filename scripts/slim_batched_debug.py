"""Where does the batched-load symmetric flow kernel (MI355REC_SLIM_BATCHED=1) start to stall?  Runs one symmetric
SLIM-BPR epoch per size, each in its own process under a timeout (a stalled hand-off aborts after 20 s of device wall clock and the
call fails; a wedged process is killed), plain kernel first as the yardstick.  Run on the GPU box:
    python scripts/slim_batched_debug.py [ml1m|ml20m] [scales ...]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import sys, time
sys.path.insert(0, %r)
from recsys2019_deeplearning_evaluation_amd import SLIM_BPR_MI355X_Epoch
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
X = named_urm(%r, "binary", scale=%r)
ep = SLIM_BPR_MI355X_Epoch(X, symmetric=True, learning_rate=1e-3, sgd_mode=%r, random_seed=3)
t = time.perf_counter()
ep.epochIteration_Cython(2)
print("%%d x %%d nnz %%d longest profile %%d: %%.2f ms per epoch" %% (X.shape[0], X.shape[1], X.nnz, max(X.indptr[1:] - X.indptr[:-1]),
      (time.perf_counter() - t) * 500), flush=True)
"""

shape = sys.argv[1] if len(sys.argv) > 1 else "ml1m"
scales = [float(a) for a in sys.argv[2:]] or [0.1, 0.2, 0.4, 0.7, 1.0]
for mode in ("sgd", "adagrad"):
    for scale in scales:
        for batched in (False, True):
            env = dict(os.environ)
            env.pop("MI355REC_SLIM_BATCHED", None)
            if batched:
                env["MI355REC_SLIM_BATCHED"] = "1"
            t0 = time.time()
            try:
                r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, shape, scale, mode)], env=env, capture_output=True, text=True, timeout=75)
                out = (r.stdout.strip() or r.stderr.strip().splitlines()[-1] if (r.stdout or r.stderr) else "no output") + (" [rc %d]" % r.returncode)
            except subprocess.TimeoutExpired:
                out = "KILLED after 75 s"
            print("%-5s scale %.2f %-7s %-8s %s  (%.1f s)" % (shape, scale, mode, "batched" if batched else "plain", out, time.time() - t0), flush=True)
