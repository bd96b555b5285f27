"""Packed-counts launch against the one-launch build on random all-ones URMs (shapes, densities, similarity modes, topK, column ranges and
interleaved parts drawn at random): every slab must be identical bit for bit.  Usage: sim_packed_fuzz.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    n_items = int(rng.integers(10500, 32000))
    n_users = int(rng.choice([3000, 20000, 70000, 150000, 260000]))
    mean_len = float(rng.choice([8, 30, 90]))
    nnz = int(min(n_users * mean_len, 2.5e7))
    X = synthetic_urm(n_users, n_items, nnz, 1, min(2000, n_items - 1), seed=int(rng.integers(1 << 30)), zipf_exponent=float(rng.choice([0.5, 0.8, 1.1])))
    sim = str(rng.choice(["cosine", "jaccard", "dice", "tversky", "asymmetric", "tanimoto"]))
    kw = dict(topK=int(rng.choice([1, 5, 50, 100, 128])), shrink=int(rng.choice([0, 0, 3, 50])), similarity=sim,
              normalize=bool(rng.integers(2)) or sim != "cosine")
    lo = int(rng.integers(0, n_items // 2)); hi = int(rng.integers(lo + 1, n_items + 1))
    rng_cols = (None, None) if rng.integers(2) else (lo, hi)
    out = {}
    for packed in ("1", "0"):
        os.environ["MI355REC_SIM_PACKED"] = packed
        dev = Compute_Similarity_MI355X(X, **kw)
        idx, val, _ = dev.compute_slabs(*rng_cols)
        out[packed] = (idx, val, dev.stats()["n_launches"], dev.selection_info())
        dev.close()
    same = np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1])
    n += 1
    print("%3d %s users %6d items %5d nnz %8d %s cols %s: launches %d/%d selection %s %s" % (
        n, "OK " if same else "MISMATCH", n_users, n_items, X.nnz, kw, rng_cols, out["1"][2], out["0"][2], out["1"][3], "" if same else "<<<<<<"), flush=True)
    if not same:
        sys.exit(1)
print("fuzz: %d cases identical" % n)
