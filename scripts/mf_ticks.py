"""Phase stamps of the mini-batch kernel at the headline shape (diagnostics; run on the GPU box).  Usage: mf_ticks.py [bpr|funk]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI355REC_MF_TICKS"] = "1"
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch, _native as N
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
X = named_urm("ml20m", "binary")
funk = len(sys.argv) > 1 and sys.argv[1] == "funk"
if funk:
    m = MatrixFactorization_MI355X_Epoch(X, n_factors=128, algorithm_name="FUNK_SVD", batch_size=1000, learning_rate=1e-3, sgd_mode="sgd", random_seed=1,
                                         use_bias=True, negative_interactions_quota=0.0)
else:
    m = MatrixFactorization_MI355X_Epoch(X, n_factors=128, algorithm_name="MF_BPR", batch_size=1000, learning_rate=1e-3, sgd_mode="sgd", random_seed=1)
m.epochIteration_Cython(1)
for tag in ("graph", ):
    n_ep = 1 if funk else 50
    m.epochIteration_Cython(n_ep); st = m.stats()
    print(tag, "epoch ms", st["call_ms"] / n_ep, "samples/s %.1fM" % (st["n_units"] / st["call_ms"] / 1e3), "us per mini-batch %.2f" % (st["call_ms"] * 1e3 / st["n_launches"]))
n = C.c_int64(0)
N.check(m._lib.mi355rec_mf_get_phase_ticks(m._h, None, 0, C.byref(n)))
t = np.zeros(n.value, np.uint64)
N.check(m._lib.mi355rec_mf_get_phase_ticks(m._h, N.ptr(t), n.value, C.byref(n)))
t = t.reshape(-1, 8).astype(np.int64)
act = t[:, 5] > 0
print("active waves", act.sum(), "of", len(t))
t0 = t[:, 0].min()
a = t[act]
for name, col in (("entry", 0), ("header", 1), ("rows", 2), ("list", 3), ("exit", 4)):
    v = (a[:, col] - t0)
    print("%-7s since first entry: min %6d  p50 %6d  p90 %6d  max %6d cycles" % (name, v.min(), np.median(v), np.quantile(v, 0.9), v.max()))
d = a[:, 4] - a[:, 0]
print("per-wave entry->exit: p50 %d p90 %d max %d" % (np.median(d), np.quantile(d, 0.9), d.max()))
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 1000)):
    sel = (a[:, 5] >= lo) & (a[:, 5] <= hi)
    if sel.any():
        print("len %3d-%3d: n %5d  header %5d rows %5d list %6d tail %5d  (median cycles per phase)" % (
            lo, hi, sel.sum(), np.median(a[sel, 1] - a[sel, 0]), np.median(a[sel, 2] - a[sel, 1]), np.median(a[sel, 3] - a[sel, 2]), np.median(a[sel, 4] - a[sel, 3])))
for lo, hi in ((1, 1), (2, 2), (3, 1000)):
    sel = (a[:, 5] >= lo) & (a[:, 5] <= hi) & (a[:, 6] > 0)
    if sel.any():
        print("len %3d-%3d: tail split: list done -> own row written %5d, -> taken-over item rows written %5d, -> exit %5d (median cycles)" % (
            lo, hi, np.median(a[sel, 6] - a[sel, 3]), np.median(a[sel, 7] - a[sel, 6]), np.median(a[sel, 4] - a[sel, 7])))
print("inactive waves exit after (median)", np.median(t[~act][:, 4] - t[~act][:, 0]) if (~act).any() else None)
print("max len", a[:, 5].max())
