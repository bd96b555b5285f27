#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: BPR-MF SGD samples/sec (+ ItemKNN cosine build
seconds) on an ML-20M-shaped synthetic URM (138 493 x 26 744, ~20 M interactions), k=128, batch 1000.

  python bench.py --gpus N --steps K --warmup W [--workload ml20m|ml1m|netflix]
  N > 1:  either as above -- bench.py then starts the N ranks itself (launch_ranks: one process per GPU, refuses to run when fewer
          than N devices are visible) -- or under python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
          127.0.0.1 ... bench.py --gpus N ... (WORLD_SIZE must then equal N).

A "step" is one reference epoch of MatrixFactorization_BPR_Cython (n_users // batch_size + 1 = 139 mini-batches of
1000 samples drawn on the device, MatrixFactorization_Cython_Epoch.pyx:583).  Inputs (URM, factors) are resident
in HBM before the timed region, which is pure hipGraph replay (no instrumentation inside it).  BPR-MF does not shard
(every sample reads and writes shared factor rows): with N GPUs every rank trains an independent replica with its own
seed -- how the reference's hyper-parameter search parallelises (run_parameter_search.py:498) -- so `value` is the
aggregate samples/s of N replicas ("weak").  The ItemKNN cosine build IS sharded (item columns, cost-balanced, one RCCL
all-gather); its wall seconds at this N are in `extra.itemknn` (strong scaling of a fixed build; "built" = full result
resident on every rank's device, the same definition at N = 1).  `--workload netflix` runs both on BASELINE.json's
configs[3] shape (480 189 x 17 770, 100 M interactions).

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel of the headline metric (mf_batch_kernel):
algorithmic bytes of one launch (batch_size x 24*k B) over the average launch duration measured with the
dispatch's own HIP start/stop events on the handle's stream in a separate, untimed call of the same run.  `extra.paths`
carries one roofline block per other hot path (SLIM-BPR, FunkSVD, IALS, scoring).  `cpu_baseline` is the reference's own
compiled Cython kernel (oracle/_ref) on one host core, same URM / k / batch.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TF = 78.6             # FP64 vector = matrix peak
F32_MFMA_PEAK_TF = 157.3
LDS_ATOMIC_PEAK = 21.6e9 * 256  # ds_add_u32 lane-adds per second, measured (profiles/r1_lds_atomics_microbench.txt) x 256 CUs
K_FACTORS = 128
BATCH = 1000
TOPK = 100


_T0 = time.perf_counter()


def note(msg):
    """Progress on stderr (the JSON line on stdout stays alone): a run that is cut short still says how far it got."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ml20m", choices=["ml20m", "ml1m", "netflix"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sim", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of each CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the other hot paths (FunkSVD, SLIM-BPR, IALS, scoring, replicas)")
    ap.add_argument("--no-paths", action="store_true", help="skip the other hot paths but keep the emulated 8-way ItemKNN build")
    ap.add_argument("--no-ials", action="store_true", help="skip the row-sharded IALS epoch (BASELINE config 5)")
    ap.add_argument("--no-netflix", action="store_true", help="skip the Netflix-shape ItemKNN build (BASELINE config 4; part of every N > 1 run and of the default N = 1 run)")
    return ap.parse_args()


def load_urm(name):
    """Synthetic URM of the named shape, cached as .npz under /tmp (generation is not part of any timing)."""
    import numpy as np
    import scipy.sparse as sps
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mi355rec_urm_%s_binary.npz" % name)
    if os.path.isfile(cache):
        try:
            return sps.load_npz(cache).tocsr().astype(np.float32)
        except Exception:
            pass
    urm = named_urm(name, "binary")
    try:
        tmp = "%s.%d.tmp.npz" % (cache, os.getpid())          # (ranks may race: write aside, then rename)
        sps.save_npz(tmp, urm, compressed=False)
        os.replace(tmp, cache)
    except Exception:
        pass
    return urm


def load_urm_once_per_node(name, net):
    """N > 1: rank 0 generates (or finds) the cached URM, the other ranks load its file after a barrier."""
    if net.world == 1 or net.rank == 0:
        urm = load_urm(name)
    net.barrier()
    if net.world > 1 and net.rank != 0:
        urm = load_urm(name)
    return urm


def pmc_traffic(kernel):
    """HBM bytes per launch from the latest committed PMC collection (profiles/*_pmc_traffic.json, produced by
    scripts/pmc_round.sh: separate FETCH_SIZE / WRITE_SIZE passes; FETCH doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  bench.py cannot run the counters itself, so the figure is NOT a measurement of this run: `source` says which
    file and when.  (None, None) if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                doc = json.load(f)
            for name, row in doc["kernels"].items():          # (rocprofv3 reports template instances: "mf_batch_kernel<0, float, 4, 32, 1>")
                if name == kernel or name.split("<")[0] == kernel:
                    return row["hbm_bytes_per_launch"], "%s: %s (collected %s)" % (os.path.basename(path), name, doc.get("collected", "?"))
        except Exception:
            continue
    return None, None


def cpu_baseline_bpr(urm, seconds):
    """The reference's compiled Cython BPR epoch (oracle/_ref) on ONE host core; falls back to the C restatement."""
    import io
    from contextlib import redirect_stdout
    from oracle import ref_loader
    kw = dict(n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3, sgd_mode="sgd", random_seed=42)
    MF = ref_loader.load("mf")
    kind = "reference"
    if MF is None:
        from oracle.oracle import OracleMF as MF
        kind = "port"
    m = MF(urm, **kw)
    per_epoch = (urm.shape[0] // BATCH + 1) * BATCH
    epochs, t0 = 0, time.perf_counter()
    with redirect_stdout(io.StringIO()):
        while True:
            m.epochIteration_Cython()
            epochs += 1
            if time.perf_counter() - t0 >= seconds:
                break
    dt = time.perf_counter() - t0
    return {"value": epochs * per_epoch / dt, "unit": "samples/s", "cores": 1, "kind": kind,
            "sample": "%d epochs (%d samples) of MF_BPR k=%d batch=%d on the same URM, %.1f s, %s" % (
                epochs, epochs * per_epoch, K_FACTORS, BATCH, dt,
                "reference Cython kernel compiled -O2 (oracle/_ref)" if kind == "reference" else "C restatement oracle/oracle.c -O2")}


def cpu_baseline_sim(urm, costs, seconds):
    """Reference Compute_Similarity_Cython on a bounded, cost-measured column range; extrapolated by cost."""
    import io
    from contextlib import redirect_stdout
    from oracle import ref_loader
    SIM = ref_loader.load("sim")
    kind = "reference"
    if SIM is None:
        from oracle.oracle import OracleSimilarity as SIM
        kind = "port"
    n = urm.shape[1]
    total = float(costs.sum())
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        obj = SIM(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
        t_init = time.perf_counter() - t0
        # a range from the middle of the popularity spectrum worth ~1.5 % of the total cost, grown until the budget is used
        start = n // 3
        end = start
        frac = 0.0
        t_cols = 0.0
        step_cost = 0.015 * total
        while t_cols < seconds and end < n - 1:
            s = end
            acc = 0.0
            while end < n - 1 and acc < step_cost:
                acc += costs[end]
                end += 1
            t1 = time.perf_counter()
            obj.compute_similarity(start_col=s, end_col=end)
            t_cols += time.perf_counter() - t1
            frac += acc / total
    est_full = t_init + t_cols / max(frac, 1e-12)
    return {"value": est_full, "unit": "s", "cores": 1, "kind": kind,
            "sample": "columns [%d,%d) = %.2f %% of the build's work ran %.1f s (+ %.1f s constructor); full build extrapolated "
                      "by cost" % (start, end, 100 * frac, t_cols, t_init)}


def _quiet():
    import io
    from contextlib import redirect_stdout
    return redirect_stdout(io.StringIO())


def cpu_baseline_slim(urm, symmetric, seconds):
    """The reference's compiled SLIM_BPR_Cython_Epoch (oracle/_ref; falls back to the C restatement) on ONE host core:
    BASELINE config 3 (adagrad, dense / symmetric store), whole reference epochs of n_users + 1 steps until the budget is used."""
    from oracle import ref_loader
    kw = dict(train_with_sparse_weights=False, final_model_sparse_weights=True, learning_rate=1e-4, li_reg=0.0, lj_reg=0.0,
              batch_size=1, topK=TOPK, symmetric=symmetric, sgd_mode="adagrad", random_seed=7)
    SL = ref_loader.load("slim")
    kind = "reference"
    if SL is None:
        from oracle.oracle import OracleSLIM as SL
        kind = "port"
    with _quiet():
        m = SL(urm, **kw)
        epochs, t0 = 0, time.perf_counter()
        while True:
            m.epochIteration_Cython()
            epochs += 1
            if time.perf_counter() - t0 >= seconds:
                break
        dt = time.perf_counter() - t0
        if hasattr(m, "_dealloc"):
            m._dealloc()
        del m                                  # (the reference's __dealloc__ prints: keep it inside the redirection)
        import gc
        gc.collect()
    n = epochs * (urm.shape[0] + 1)
    return {"value": n / dt, "unit": "samples/s", "cores": 1, "kind": kind,
            "sample": "%d reference epochs (%d steps) of SLIM_BPR adagrad, %s store, on the same URM, %.1f s" % (
                epochs, n, "symmetric (triangular)" if symmetric else "dense", dt)}


def _mf_reference(urm, **kw):
    from oracle import ref_loader
    MF = ref_loader.load("mf")
    kind = "reference"
    if MF is None:
        from oracle.oracle import OracleMF as MF
        kind = "port"
    return MF(urm, **kw), kind


def cpu_baseline_funk(urm, seconds):
    """Reference FunkSVD epoch (k=128, batch 1000, biases) on ONE host core.  A reference epoch is nnz // 1000 + 1 mini-batches
    (20 M samples at this shape: minutes on a CPU) and cannot be cut short, so the sample is the epoch of a URM of the SAME shape
    (same factor matrices, same row gathers) holding a random 1/40 of the interactions."""
    import numpy as np
    rng = np.random.default_rng(11)
    sub = urm.tocoo()
    keep = rng.random(sub.nnz) < 1.0 / 40.0
    import scipy.sparse as sps
    sub = sps.csr_matrix((sub.data[keep], (sub.row[keep], sub.col[keep])), shape=urm.shape, dtype=np.float32)
    sub.sort_indices()
    with _quiet():
        m, kind = _mf_reference(sub, n_factors=K_FACTORS, algorithm_name="FUNK_SVD", batch_size=BATCH, learning_rate=1e-3, sgd_mode="sgd",
                                use_bias=True, negative_interactions_quota=0.0, random_seed=42)
        per_epoch = (sub.nnz // BATCH + 1) * BATCH
        epochs, t0 = 0, time.perf_counter()
        while True:
            m.epochIteration_Cython()
            epochs += 1
            if time.perf_counter() - t0 >= seconds:
                break
        dt = time.perf_counter() - t0
    return {"value": epochs * per_epoch / dt, "unit": "samples/s", "cores": 1, "kind": kind,
            "sample": "%d reference epochs (%d samples) of FUNK_SVD k=%d batch=%d with biases on a %dx%d URM holding 1/40 of the interactions "
                      "(%d nnz; same factor matrices), %.1f s" % (epochs, epochs * per_epoch, K_FACTORS, BATCH, urm.shape[0], urm.shape[1], sub.nnz, dt)}


def cpu_baseline_asy(urm, k, seconds):
    """Reference AsySVD epoch (nnz + 1 ordered steps, each O(profile x k)) on ONE host core, on the rows of a random subset of the
    users (same items, same profile-length distribution, same Y and X matrices), sized for the budget."""
    import numpy as np
    rng = np.random.default_rng(12)
    n_users = urm.shape[0]
    order = rng.permutation(n_users)
    take = max(50, n_users // 16)
    total_steps, total_dt, used = 0, 0.0, 0
    kind = "reference"
    with _quiet():
        while total_dt < seconds and used < n_users:
            rows = np.sort(order[used:used + take])
            used += len(rows)
            sub = urm[rows].tocsr()
            sub.sort_indices()
            m, kind = _mf_reference(sub, n_factors=k, algorithm_name="ASY_SVD", batch_size=1, learning_rate=1e-3, sgd_mode="sgd",
                                    use_bias=True, negative_interactions_quota=0.0, random_seed=42)
            t0 = time.perf_counter()
            m.epochIteration_Cython()
            total_dt += time.perf_counter() - t0
            total_steps += sub.nnz + 1
    return {"value": total_steps / total_dt, "unit": "samples/s", "cores": 1, "kind": kind,
            "sample": "reference ASY_SVD epochs (k=%d, biases) on the rows of %d of the %d users (%d steps, mean profile %.0f), %.1f s" % (
                k, used, n_users, total_steps, urm.nnz / n_users, total_dt)}


def cpu_baseline_ials(conf, k, reg, V0, seconds):
    """IALSRecommender._update_row (NumPy, IALSRecommender.py:170-201: gather, k x k Gramian update, np.linalg.inv) on ONE host
    thread, as restated by oracle.oracle._ials_update_row -- the reference class itself is pure Python and does not travel to the
    GPU box.  A random sample of user rows and item rows is timed; a whole epoch (every warm user row, then every warm item row)
    is extrapolated by the flop cost 2 L k^2 + 2 k^3 of a row with L interactions."""
    import numpy as np
    from oracle import oracle as O
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    try:
        rng = np.random.default_rng(13)
        csr = conf.tocsr()
        csc = conf.tocsc()
        V = np.asarray(V0, dtype=np.float64)
        U = rng.normal(0, 0.1, (conf.shape[0], k))
        reg_diag = np.diag(reg * np.ones(k))
        VV, UU = V.T.dot(V), U.T.dot(U)
        cost = lambda L: 2.0 * L * k * k + 2.0 * k ** 3
        full = cost(np.diff(csr.indptr)[np.diff(csr.indptr) > 0]).sum() + cost(np.diff(csc.indptr)[np.diff(csc.indptr) > 0]).sum()
        done, t_used, n_rows = 0.0, 0.0, 0
        users, items = rng.permutation(conf.shape[0]), rng.permutation(conf.shape[1])
        pos = 0
        while t_used < seconds and pos < min(len(users), len(items)):
            t0 = time.perf_counter()
            for u in users[pos:pos + 32]:
                s, e = csr.indptr[u], csr.indptr[u + 1]
                if e > s:
                    O._ials_update_row(csr.indices[s:e], csr.data[s:e], V, VV, reg_diag)
                    done += cost(e - s); n_rows += 1
            for i in items[pos:pos + 8]:
                s, e = csc.indptr[i], csc.indptr[i + 1]
                if e > s:
                    O._ials_update_row(csc.indices[s:e], csc.data[s:e], U, UU, reg_diag)
                    done += cost(e - s); n_rows += 1
            t_used += time.perf_counter() - t0
            pos += 32
    finally:
        if limit is not None:
            limit.unregister() if hasattr(limit, "unregister") else None
    return {"value": t_used * full / max(done, 1.0), "unit": "s/epoch", "cores": 1, "kind": "port",
            "sample": "%d rows (users and items at random) = %.3f %% of an epoch's 2 L k^2 + 2 k^3 flops in %.1f s, NumPy on one BLAS thread; "
                      "epoch extrapolated by that cost.  kind 'port': oracle._ials_update_row, the NumPy restatement of IALSRecommender._update_row, "
                      "timed on THIS host in THIS run; the reference class itself cannot travel to the GPU box -- its own figure is the committed "
                      "fixture `cpu_baseline_reference_fixture`, timed on a DIFFERENT host" % (n_rows, 100.0 * done / full, t_used)}


def hbm_block(kernel, st, seconds, note=None):
    gbps = st["algorithmic_bytes"] / seconds / 1e9
    out = {"bound": "hbm", "kernel": kernel, "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
           "samples_per_s": st["n_units"] / seconds, "seconds": seconds}
    if note:
        out["note"] = note
    return out


def other_paths(urm, args, out=None, cpu_jobs=None):
    """The remaining rows of SURVEY.md section 8 on the same URM shape, one short run each (N = 1 only): one roofline block per
    path.  Throughputs come from the handle's own stream events (call_ms); fractions are ALGORITHMIC work / time against the
    MI355X peaks.  Fills `out` as it goes: what was measured before a failure stays.  The CPU baseline leg of
    a path is NOT run here: it is appended to `cpu_jobs` as a closure that fills the path's block, and main() runs those one after
    the other once every GPU measurement is done and the URM-generating child process has exited (a leg that shares the host with
    another busy process would flatter the GPU)."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import (IALS_MI355X_Epoch, MatrixFactorization_MI355X_Epoch,
                                                        SLIM_BPR_MI355X_Epoch)
    out = {} if out is None else out
    cpu_jobs = [] if cpu_jobs is None else cpu_jobs

    def later(tag, leg, ratio):
        """ratio(block, baseline) -> the GPU / CPU speed-up of that path's own unit"""
        def job():
            blk = out[tag]
            blk["cpu_baseline"] = leg()
            blk["speedup_vs_cpu_baseline"] = ratio(blk, blk["cpu_baseline"]["value"])
        cpu_jobs.append((tag, job))

    def mf_run(tag, epochs, note=None, **kw):
        m = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, learning_rate=1e-3, init_std_dev=0.1, random_seed=7, **kw)
        m.epochIteration_Cython(1)
        m.epochIteration_Cython(epochs)
        st = m.stats()
        out[tag] = hbm_block("mf_batch_kernel", st, st["call_ms"] * 1e-3, note)
        out[tag]["epochs"] = epochs
        m.close()

    cpu = not args.no_cpu_baseline
    note("paths: adagrad, funk")
    mf_run("bpr_mf_k128_batch1000_adagrad", 50, "float64 factors + moments (adaptive optimisers)", algorithm_name="MF_BPR",
           batch_size=BATCH, sgd_mode="adagrad")
    mf_run("funk_svd_k128_batch1000_bias", 1, "20 001 mini-batches per epoch, in-LDS schedule 256 mini-batches at a time", algorithm_name="FUNK_SVD",
           batch_size=BATCH, sgd_mode="sgd", use_bias=True, negative_interactions_quota=0.0)
    if cpu:
        later("funk_svd_k128_batch1000_bias", lambda: cpu_baseline_funk(urm, args.cpu_seconds), lambda b, c: b["samples_per_s"] / c)
    # the reference's own arithmetic type (MatrixFactorization_Cython_Epoch.pyx:65: double): the same headline epoch with float64 factors
    mf_run("bpr_mf_k128_batch1000_fp64", 50, "float64 factors, plain sgd: the reference's arithmetic width; 48 k B algorithmic per sample",
           algorithm_name="MF_BPR", batch_size=BATCH, sgd_mode="sgd", precision="fp64")

    note("paths: 32-model group")
    # REPLICA-BATCHED launches: 32 independent models (own factors, seed, sample stream), mini-batch b of all of them in ONE grid
    # (mi355rec_mf_group_*) -- the device-side form of run_parameter_search.py:498's pool of workers.  Every member ends
    # bit-identical to training alone (tests/test_mf_gpu.py::test_group_*).
    n_grp, epochs = 32, 30
    rng = np.random.default_rng(0)
    U0 = rng.normal(0, 0.1, (urm.shape[0], K_FACTORS)).astype(np.float32)
    V0 = rng.normal(0, 0.1, (urm.shape[1], K_FACTORS)).astype(np.float32)
    from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Group
    members = [MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3,
                                                sgd_mode="sgd", random_seed=200 + r, initial_USER_factors=U0, initial_ITEM_factors=V0)
               for r in range(n_grp)]
    grp = MatrixFactorization_MI355X_Group(members)
    grp.epochIteration_Cython(2)
    grp.epochIteration_Cython(epochs)
    gst = grp.stats()
    nb = urm.shape[0] // BATCH + 1
    grp.set_profiling(2 * nb)
    grp.epochIteration_Cython(2)
    pst = grp.stats()
    sec = gst["call_ms"] * 1e-3
    launch_s = pst["kernel_ms"] / max(1, pst["n_timed"]) * 1e-3
    alg_launch = n_grp * BATCH * 24.0 * K_FACTORS
    traffic, traffic_source = pmc_traffic("mf_group_batch_kernel")
    out["bpr_mf_k128_batch1000_32_models_one_launch"] = {
        "bound": "hbm", "kernel": "mf_group_batch_kernel", "achieved": alg_launch / launch_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": alg_launch / launch_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": alg_launch, "avg_launch_us": launch_s * 1e6, "timed_launches": pst["n_timed"],
        "samples_per_s": gst["n_units"] / sec, "whole_epoch_frac": gst["algorithmic_bytes"] / sec / 1e9 / HBM_PEAK_GBPS,
        "models": n_grp, "epochs": epochs, "seconds": sec,
        "note": "aggregate of 32 independent models; one launch per mini-batch index carries all 32 (grid 750 x 32 workgroups)"}
    grp.close()
    for m in members:
        m.close()
    del U0, V0

    note("paths: 8 replicas on 8 streams")
    # concurrent replicas on ONE GPU: how run_parameter_search.py:498 uses the path (one model per worker); 8 handles, 8 streams
    n_rep, epochs = 8, 100
    reps = [MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH, learning_rate=1e-3,
                                             sgd_mode="sgd", random_seed=100 + r) for r in range(n_rep)]
    for m in reps:
        m.epochIteration_Cython(2)
    threads = [threading.Thread(target=m.epochIteration_Cython, args=(epochs,)) for m in reps]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    per_epoch = (urm.shape[0] // BATCH + 1) * BATCH
    out["bpr_mf_k128_batch1000_8_replicas_one_gpu"] = {
        "bound": "hbm", "kernel": "mf_batch_kernel", "samples_per_s": n_rep * epochs * per_epoch / wall,
        "achieved": n_rep * epochs * per_epoch * 24.0 * K_FACTORS / wall / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": n_rep * epochs * per_epoch * 24.0 * K_FACTORS / wall / 1e9 / HBM_PEAK_GBPS, "replicas": n_rep, "seconds": wall,
        "note": "aggregate of 8 independent models training concurrently on one device (host wall clock)"}
    for m in reps:
        m.close()

    note("paths: exact multi-GPU mode, emulated")
    # SURVEY 8(e)'s exact multi-GPU mode at a batch size where a mini-batch fills the chip (65 536: outside the reference's search
    # space, which stops at 1024): this handle plays rank 0 of 8 -- its share of every mini-batch's row tasks, the packing of the
    # rows it owns and the merge of the other ranks' slabs are MEASURED; the all-gather between them is modelled from its size
    big = 65536
    one = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=big, learning_rate=1e-3,
                                           sgd_mode="sgd", random_seed=7)
    one.epochIteration_Cython(2)
    one.epochIteration_Cython(20)
    st1 = one.stats()
    single_rate = st1["n_units"] / (st1["call_ms"] * 1e-3)
    one.close()
    r0 = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=big, learning_rate=1e-3,
                                          sgd_mode="sgd", random_seed=7)
    t_batch = t_merge = 0.0
    n_done = 0
    for ep in range(3):
        _, _, nbytes, nb = r0.shard_begin_epoch(0, 8)
        for b in range(nb):
            t0 = time.perf_counter(); r0.shard_batch(b); t1 = time.perf_counter(); r0.shard_merge(b); t2 = time.perf_counter()
            if ep > 0:
                t_batch += t1 - t0; t_merge += t2 - t1; n_done += 1
        r0.shard_end_epoch()
    r0.close()
    ring_s = 7 * nbytes / 50e9 + 50e-6
    direct_s = nbytes / 50e9 + 50e-6
    per_batch = t_batch / n_done + t_merge / n_done
    out["bpr_mf_exact_mode_8_gpus_batch65536_emulated"] = {
        "single_gpu_samples_per_s": single_rate, "rank_share_ms_per_batch": t_batch / n_done * 1e3, "merge_ms_per_batch": t_merge / n_done * 1e3,
        "slab_MB_per_rank_per_batch": nbytes / 1e6,
        "modelled_allgather_ms": {"one_ring_50GBps_per_link": ring_s * 1e3, "seven_links_at_once": direct_s * 1e3},
        "predicted_samples_per_s": {"one_ring": big / (per_batch + ring_s), "seven_links": big / (per_batch + direct_s)},
        "note": "rank 0 of 8 measured on one GPU (host-timed, blocking calls), exchange modelled, unmeasured on hardware: every mini-batch moves "
                "the rows it touches (3 x batch x k x 4 B in total) over xGMI, so the exact mode cannot beat one GPU that moves the same bytes "
                "through HBM; replicas (one model per GPU) are the mode that scales"}

    for symmetric in (False, True):
        note("paths: slim symmetric=%s" % symmetric)
        sl = SLIM_BPR_MI355X_Epoch(urm, symmetric=symmetric, sgd_mode="adagrad", learning_rate=1e-4, topK=TOPK, random_seed=7)
        sl.epochIteration_Cython(1)
        n_ep = 3
        t0 = time.perf_counter()
        sl.epochIteration_Cython(n_ep)
        wall = time.perf_counter() - t0
        st = sl.stats()
        # (the next epoch's schedule runs on a second stream behind the kernel: the host's clock around the blocking call is the
        # honest figure, the main stream's events cannot be longer)
        sec = max(st["call_ms"] * 1e-3, wall)
        t0 = time.perf_counter()
        sl.get_S_slabs(TOPK)
        kernel = "slim_sym_flow_kernel" if symmetric else "slim_dense_flow_kernel"
        blk = hbm_block(kernel, st, sec, "BASELINE config 3 (adagrad); one persistent dataflow kernel per epoch: " + (
            "8-byte {value, tag} cells polled in place, long profiles by a whole workgroup" if symmetric else "the busiest rows owned in LDS by turn-taking workgroups, the other steps one wavefront each") + "; the next epoch is sampled and scheduled on a second stream behind the kernel")
        traffic, traffic_source = pmc_traffic(kernel)
        blk.update({"seconds_per_epoch": sec / n_ep, "flow_kernel_ms_per_epoch": st["kernel_ms"] / n_ep,
                    "us_per_step_amortised": sec / st["n_units"] * 1e6, "get_S_topk_s": time.perf_counter() - t0,
                    "traffic": traffic, "traffic_source": traffic_source, "bound_note": "latency: the chain of dependent steps on the busiest "
                    "row / cells (1 214 / 3 895 links at this shape, profiles/r4_slim_critical_paths.txt), not bytes"})
        if not symmetric:
            owned, cold = sl.schedule_info()
            blk.update({"owned_rows": owned, "steps_on_rows_in_hbm": cold})
        out["slim_bpr_%s" % ("symmetric" if symmetric else "dense")] = blk
        sl.close()
        if cpu:
            later("slim_bpr_%s" % ("symmetric" if symmetric else "dense"), lambda symmetric=symmetric: cpu_baseline_slim(urm, symmetric, args.cpu_seconds),
                  lambda b, c: b["samples_per_s"] / c)

    note("paths: slim, 4 models side by side")
    # R independent SLIM models on R streams (how the reference's search runs every SGD path: run_parameter_search.py:498-503); the
    # dense store's compute units are leased R ways so that every model's owned rows stay resident
    R = 4
    os.environ["MI355REC_SLIM_CUS"] = str(256 // R)
    try:
        reps = [SLIM_BPR_MI355X_Epoch(urm, symmetric=False, sgd_mode="adagrad", learning_rate=1e-4, topK=TOPK, random_seed=300 + r) for r in range(R)]
        for m in reps:
            m.epochIteration_Cython(1)
        n_ep = 4
        threads = [threading.Thread(target=m.epochIteration_Cython, args=(n_ep,)) for m in reps]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        wall = time.perf_counter() - t0
        out["slim_bpr_dense_4_models_side_by_side"] = {
            "samples_per_s": R * n_ep * (urm.shape[0] + 1) / wall, "models": R, "seconds": wall, "owned_rows_per_model": [m.schedule_info()[0] for m in reps],
            "note": "aggregate of 4 independent models (4 handles, streams and host threads); one model alone: slim_bpr_dense.  The epoch of one "
                    "model is bound by its chain of dependent steps, and the polling of concurrent persistent kernels lengthens every link: "
                    "models side by side add little (profiles/r4_slim_replicas_first.txt)"}
        for m in reps:
            m.close()
    finally:
        os.environ.pop("MI355REC_SLIM_CUS", None)

    note("paths: scoring")
    # scoring + ranking of 1000 users (the Evaluator's block size, Base/Evaluation/Evaluator.py:406-408), k = 128
    from recsys2019_deeplearning_evaluation_amd import MI355XScorer
    rng = np.random.default_rng(0)
    Uf = rng.normal(0, 0.1, (urm.shape[0], K_FACTORS)).astype(np.float32)
    Vf = rng.normal(0, 0.1, (urm.shape[1], K_FACTORS)).astype(np.float32)
    sc = MI355XScorer(Uf, Vf, urm)
    users = rng.choice(urm.shape[0], 1000, replace=False).astype(np.int32)
    sc.recommend(users, 20)
    t0 = time.perf_counter()
    for _ in range(5):
        sc.recommend(users, 20)
    wall = (time.perf_counter() - t0) / 5
    st = sc.stats()
    t0 = time.perf_counter()                        # the reference's host path for the same block (NumPy, all host cores via BLAS)
    host = Uf[users] @ Vf.T
    for r, u in enumerate(users):
        host[r, urm.indices[urm.indptr[u]:urm.indptr[u + 1]]] = -np.inf
    part = (-host).argpartition(20, axis=1)[:, :20]
    np.argsort(-host[np.arange(1000)[:, None], part], axis=1)
    host_wall = time.perf_counter() - t0
    tf = st["algorithmic_flops"] / (st["kernel_ms"] * 1e-3) / 1e12
    out["mf_scoring_1000_users_cutoff20"] = {"bound": "mfma", "kernel": "score_gemm_kernel", "achieved": tf, "peak": F32_MFMA_PEAK_TF,
                                             "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF, "users_per_s": 1000 / wall,
                                             "device_ms": st["call_ms"], "gemm_ms": st["kernel_ms"],
                                             "host_numpy_users_per_s": 1000 / host_wall}
    sc.close()

    note("paths: ials k=200")
    # BASELINE config 5: IALS k = 200 on the ML-20M shape (one GPU here; the row-sharded epoch is sharding.sharded_ials_epoch)
    k = 200
    conf = urm.copy()
    conf.data = (1.0 + 1.0 * conf.data).astype(np.float32)
    V0 = k ** -0.5 * np.random.default_rng(0).random((urm.shape[1], k))
    ia = IALS_MI355X_Epoch(conf, k, 1e-3, V0)
    ia.run_epochs(1)                                  # warm-up (first touch of the handle's buffers), like every other path here
    ia.run_epochs(1)
    st = ia.stats()
    sec = st["call_ms"] * 1e-3
    tf = st["algorithmic_flops"] / sec / 1e12
    out["ials_k200"] = {"bound": "fp64", "kernel": "ials_row_kernel", "achieved": tf, "peak": FP64_PEAK_TF, "unit": "TFLOP/s",
                        "frac": tf / FP64_PEAK_TF, "seconds_per_epoch": sec, "row_solves_per_s": st["n_units"] / sec,
                        "row_kernel_ms": st["kernel_ms"]}
    ia.close()
    if cpu:
        later("ials_k200", lambda: cpu_baseline_ials(conf, k, 1e-3, V0, args.cpu_seconds), lambda b, c: c / b["seconds_per_epoch"])
    # the REFERENCE's own _update_row timed where /root/reference exists (tests/golden/make_ials_reference_timing.py): a committed fixture,
    # from another host than this run's -- next to, not instead of, the same-run port above
    try:
        with open(os.path.join(ROOT, "tests", "golden", "ials_reference_timing.json")) as f:
            fx = json.load(f)
        out["ials_k200"]["cpu_baseline_reference_fixture"] = {
            "value": fx["seconds_per_epoch_extrapolated"], "unit": "s/epoch", "cores": fx["blas_threads"], "kind": "reference-fixture",
            "sample": "NOT this run and NOT this host: the reference's IALSRecommender._update_row, %d rows = %.2f %% of an epoch's flops in %.1f s on %s (%s); "
                      "committed fixture tests/golden/ials_reference_timing.json" % (fx["rows_timed"], 100 * fx["fraction_of_an_epochs_flops"], fx["seconds"],
                                                                                   fx["cpu"], fx["generated"])}
    except Exception:
        pass

    note("paths: asysvd")
    # AsySVD at the ML-1M shape, k = 64, biases: nnz + 1 strictly ordered steps, each rewriting every Y row of the
    # sampled user's profile -- consecutive steps share the popular items' rows, so the epoch is one dependent chain
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    x1m = named_urm("ml1m", "real")
    ka = 64
    asy = MatrixFactorization_MI355X_Epoch(x1m, n_factors=ka, algorithm_name="ASY_SVD", batch_size=1, learning_rate=1e-3, sgd_mode="sgd",
                                           use_bias=True, negative_interactions_quota=0.0, random_seed=42)
    asy.epochIteration_Cython(1)
    st = asy.stats()
    sec = st["call_ms"] * 1e-3
    steps = st["n_units"]
    mean_len = x1m.nnz / x1m.shape[0]
    alg = steps * (16.0 * ka * mean_len + 8.0 * ka)          # every profile row of Y read and written once (fp32) + X_i
    out["asy_svd_ml1m_k64"] = {"bound": "latency (one dependent chain of steps)", "kernel": "mf_asy_kernel", "samples_per_s": steps / sec,
                               "seconds_per_epoch": sec, "us_per_step": sec / steps * 1e6, "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": alg / sec / 1e9 / HBM_PEAK_GBPS,
                               "note": "ML-1M shape (6 040 x 3 706, 1 000 209 interactions); algorithmic bytes = 16 k L_u + 8 k per step"}
    asy.close()
    if cpu:
        later("asy_svd_ml1m_k64", lambda: cpu_baseline_asy(x1m, ka, args.cpu_seconds), lambda b, c: b["samples_per_s"] / c)
    return out


class HoldoutEvaluator:
    """What Base/Evaluation/Evaluator.py:EvaluatorHoldout does per validation, restated for the bench's end-to-end row: every user
    with a held-out item, in blocks of 1000 (Evaluator.py:406-408), recommender.recommend(block, cutoff, remove_seen_flag=True),
    PRECISION / RECALL / MAP at the cutoff on the host.  Keeps its own wall clock (`seconds`, `calls`)."""

    def __init__(self, URM_test, cutoff=10):
        import numpy as np
        self.URM_test = URM_test.tocsr()
        self.cutoff = cutoff
        self.users = np.flatnonzero(np.diff(self.URM_test.indptr) > 0).astype(np.int64)
        self.seconds, self.calls = 0.0, 0

    def evaluateRecommender(self, recommender):
        import numpy as np
        t0 = time.perf_counter()
        X, c = self.URM_test, self.cutoff
        n_items = X.shape[1]
        hits_sum = recall_sum = map_sum = 0.0
        rank_weight = 1.0 / (1.0 + np.arange(c))
        for at in range(0, len(self.users), 1000):
            block = self.users[at:at + 1000]
            lists = recommender.recommend(block, cutoff=c, remove_seen_flag=True)
            rec = np.full((len(block), c), -1, np.int64)
            for r, row in enumerate(lists):
                rec[r, :min(c, len(row))] = row[:c]
            # is_relevant for the whole block at once: (row, item) pairs as single keys
            n_rel = (X.indptr[block + 1] - X.indptr[block]).astype(np.int64)
            rows = np.repeat(np.arange(len(block), dtype=np.int64), n_rel)
            cols = np.concatenate([X.indices[X.indptr[u]:X.indptr[u + 1]] for u in block]) if len(block) else np.zeros(0, np.int64)
            hit = np.isin(np.arange(len(block), dtype=np.int64)[:, None] * n_items + rec, rows * n_items + cols) & (rec >= 0)
            n_hit = hit.sum(1).astype(np.float64)
            hits_sum += float((n_hit / c).sum())
            recall_sum += float((n_hit / n_rel).sum())
            map_sum += float(((hit * np.cumsum(hit, 1) * rank_weight).sum(1) / np.minimum(n_rel, c)).sum())
        n = max(1, len(self.users))
        res = {c: {"PRECISION": hits_sum / n, "RECALL": recall_sum / n, "MAP": map_sum / n}}
        self.seconds += time.perf_counter() - t0
        self.calls += 1
        return res, "CUTOFF: %d - PRECISION: %.6f, RECALL: %.6f, MAP: %.6f" % (c, res[c]["PRECISION"], res[c]["RECALL"], res[c]["MAP"])


def holdout_split(urm, seed=5):
    """Leave one random interaction out per user with at least two (train, test) -- the shape of the suite's own validation splits."""
    import numpy as np
    import scipy.sparse as sps
    rng = np.random.default_rng(seed)
    lens = np.diff(urm.indptr)
    pick = urm.indptr[:-1] + (rng.random(len(lens)) * np.maximum(lens, 1)).astype(np.int64)
    pick = pick[lens >= 2]
    keep = np.ones(urm.nnz, bool)
    keep[pick] = False
    rows = np.repeat(np.arange(urm.shape[0]), lens)
    train = sps.csr_matrix((urm.data[keep], (rows[keep], urm.indices[keep])), shape=urm.shape, dtype=np.float32)
    test = sps.csr_matrix((urm.data[pick], (rows[pick], urm.indices[pick])), shape=urm.shape, dtype=np.float32)
    train.sort_indices()
    return train, test


def fit_rows(urm, out):
    """Whole-`fit()` wall time THROUGH the recommender classes -- the only kind of number the reference publishes (BASELINE.md
    section 1) and what "drops into the evaluation harness" costs end to end:
      ItemKNNCFRecommender(urm).fit(topK=100, shrink=0)  (KNN/ItemKNNCFRecommender.py:31-54): PCIe upload, device constructor, build,
        download, host CSR assembly into W_sparse;
      MatrixFactorization_BPR_MI355X(train).fit(epochs=50, validation_every_n=5, evaluator_object=...)
        (MatrixFactorization_Cython.py:37-143, Incremental_Training_Early_Stopping.py:91-192): one blocking call per epoch, and per
        validation the factor download (_prepare_model_for_validation) + the evaluator's recommend() blocks on the device scorer."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import ItemKNNCFRecommender, MatrixFactorization_BPR_MI355X
    rec = ItemKNNCFRecommender(urm, verbose=False)
    walls = []
    for _ in range(4):
        t0 = time.perf_counter()
        rec.fit(topK=TOPK, shrink=0)
        walls.append(time.perf_counter() - t0)
    st = rec.similarity_stats
    out["itemknn_recommender_fit"] = {
        "value": min(walls[1:]), "unit": "s", "first_call_s": walls[0], "repeats_s": walls[1:], "W_sparse_nnz": int(rec.W_sparse.nnz),
        "kernel_ms": st.get("kernel_ms"), "definition": "ItemKNNCFRecommender(urm).fit(topK=100, shrink=0) wall: upload + constructor + build + "
        "download + W_sparse (scipy CSR) assembly; best of 3 after one warm-up call"}
    del rec

    train, test = holdout_split(urm)
    ev = HoldoutEvaluator(test, cutoff=10)

    class Timed(MatrixFactorization_BPR_MI355X):
        t_epoch = t_prepare = 0.0

        def _run_epoch(self, num_epoch):
            t0 = time.perf_counter()
            super()._run_epoch(num_epoch)
            Timed.t_epoch += time.perf_counter() - t0

        def _prepare_model_for_validation(self):
            t0 = time.perf_counter()
            super()._prepare_model_for_validation()
            Timed.t_prepare += time.perf_counter() - t0

    epochs, every = 50, 5
    m = Timed(train, verbose=False)
    t0 = time.perf_counter()
    m.fit(epochs=epochs, batch_size=BATCH, num_factors=K_FACTORS, learning_rate=1e-3, sgd_mode="sgd", random_seed=42,
          validation_every_n=every, stop_on_validation=False, validation_metric="MAP", evaluator_object=ev)
    wall = time.perf_counter() - t0
    per_epoch = (train.shape[0] // BATCH + 1) * BATCH
    out["bpr_mf_recommender_fit_50_epochs_validation_every_5"] = {
        "value": wall / epochs, "unit": "s/epoch", "fit_wall_s": wall, "epochs": epochs, "validations": ev.calls,
        "train_s": Timed.t_epoch, "factor_download_s": Timed.t_prepare, "evaluate_s": ev.seconds,
        "evaluated_users_per_validation": int(len(ev.users)), "train_samples_per_s": epochs * per_epoch / max(Timed.t_epoch, 1e-9),
        "end_to_end_samples_per_s": epochs * per_epoch / wall, "best_MAP": m.best_validation_metric,
        "definition": "MatrixFactorization_BPR_MI355X(train).fit(epochs=50, batch 1000, k=128, validation_every_n=5, MAP@10 on a leave-one-out "
                      "split, every test user) wall / 50: constructor, one blocking epoch call per epoch, factor download + device-scored "
                      "recommend() blocks per validation"}

    # SLIM-BPR (BASELINE config 3) through its recommender: per validation the fit loop asks for get_S() AND W_sparse =
    # similarityMatrixTopK(get_S(), topK) (SLIM_BPR_Cython.py:186-197) -- both selections run on the device (mi355rec_slim_get_W_csr)
    from recsys2019_deeplearning_evaluation_amd import SLIM_BPR_MI355X
    ev2 = HoldoutEvaluator(test, cutoff=10)

    class TimedSlim(SLIM_BPR_MI355X):
        t_epoch = t_prepare = 0.0

        def _run_epoch(self, num_epoch):
            t0 = time.perf_counter()
            super()._run_epoch(num_epoch)
            TimedSlim.t_epoch += time.perf_counter() - t0

        def _prepare_model_for_validation(self):
            t0 = time.perf_counter()
            super()._prepare_model_for_validation()
            TimedSlim.t_prepare += time.perf_counter() - t0

    epochs, every = 30, 10
    sl = TimedSlim(train, verbose=False)
    t0 = time.perf_counter()
    sl.fit(epochs=epochs, symmetric=False, topK=TOPK, sgd_mode="adagrad", learning_rate=1e-4, random_seed=42,
           validation_every_n=every, stop_on_validation=False, validation_metric="MAP", evaluator_object=ev2)
    wall = time.perf_counter() - t0
    out["slim_bpr_recommender_fit_30_epochs_validation_every_10"] = {
        "value": wall / epochs, "unit": "s/epoch", "fit_wall_s": wall, "epochs": epochs, "validations": ev2.calls,
        "train_s": TimedSlim.t_epoch, "get_S_and_W_s": TimedSlim.t_prepare, "evaluate_s": ev2.seconds,
        "train_samples_per_s": epochs * (train.shape[0] + 1) / max(TimedSlim.t_epoch, 1e-9), "W_sparse_nnz": int(sl.W_sparse.nnz),
        "definition": "SLIM_BPR_MI355X(train).fit(epochs=30, dense store, topK=100, adagrad, validation_every_n=10, MAP@10 on the same split) "
                      "wall / 30: constructor, one blocking epoch call per epoch, per validation get_S + the column top-K of W_sparse (device) "
                      "+ the evaluator's recommend() blocks on the sparse device scorer"}
    return out


DRIVER_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
               "config")
EXTRA_FILE = "bench_extra.json"


def _sig(x, digits=5):
    """Numbers of the compact line carry 5 significant digits (the full-precision record is the extra file)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def _slim(d, keys):
    return {k: _sig(d[k]) for k in keys if d.get(k) is not None}


def compact_line(out):
    """The ONE line of stdout: the driver's keys, `roofline` and `cpu_baseline` of the headline kernel, and one short row per other
    hot path (value, unit, bound, fraction of that bound, CPU figure of the same run) -- no notes, no per-piece arrays.  Everything
    else the run measured is in EXTRA_FILE (and, as one earlier line, on stderr).  Stays far below 8 KB whatever the sections hold
    (tests/test_bench_line.py builds it from a recorded run)."""
    line = {k: _sig(out[k]) for k in DRIVER_KEYS if k in out}
    line["roofline"] = _slim(out.get("roofline", {}), ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                                                      "algorithmic_bytes_per_launch", "timed_launches"))
    line["roofline"].setdefault("traffic", None)
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _slim(out["cpu_baseline"], ("value", "unit", "cores", "kind", "host_cpu_count"))
        line["cpu_baseline"]["sample"] = str(out["cpu_baseline"].get("sample", ""))[:160]
    extra = out.get("extra", {})
    for key in ("itemknn", "itemknn_netflix_config4"):
        blk = extra.get(key)
        if isinstance(blk, dict) and blk.get("cosine_build_s") is not None:
            line[key + "_build_s"] = _sig(blk["cosine_build_s"])
    if isinstance(extra.get("ials"), dict) and extra["ials"].get("seconds_per_epoch") is not None:
        line["ials_k200_sharded_epoch_s"] = _sig(extra["ials"]["seconds_per_epoch"])
    if "speedup_vs_cpu_baseline" in extra:
        line["speedup_vs_cpu_baseline"] = _sig(extra["speedup_vs_cpu_baseline"])
    rows = {}
    for name, row in out.get("paths", {}).items():
        if row.get("value") is None:
            continue
        rows[name] = _slim(row, ("value", "unit", "bound", "frac", "cpu_value", "cpu_kind", "build_s", "fit_resident_s", "build_gather_to_rank0_s", "predicted_8_gpu_one_ring"))
        if isinstance(rows[name].get("bound"), str):
            rows[name]["bound"] = rows[name]["bound"].split(" ")[0]
    line["paths"] = rows
    for key in ("paths_error", "ials_error", "fit_rows_error", "cpu_legs_error"):
        if key in extra:
            line[key] = str(extra[key])[:120]
    line["extra_file"] = EXTRA_FILE
    return line


def emit(out):
    """Full record -> EXTRA_FILE next to bench.py (and under gpurun_out/ when that scratch directory exists) and one stderr line;
    compact record -> the one stdout line."""
    full = json.dumps(out)
    for folder in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(folder):
            try:
                with open(os.path.join(folder, EXTRA_FILE), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
    print("[bench extra] " + full, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(out))
    assert len(line) < 8192, "the stdout line must stay short enough for the driver to parse (%d bytes)" % len(line)
    print(line, flush=True)


def visible_devices():
    """HIP devices this process could bind, counted in a child process (the launcher itself never creates a HIP context).
    BENCH_ASSUME_DEVICES overrides the count -- for the launcher's own CPU tests (tests/test_bench_launcher.py) only."""
    import subprocess
    if os.environ.get("BENCH_ASSUME_DEVICES"):
        return int(os.environ["BENCH_ASSUME_DEVICES"])
    code = ("import sys; sys.path.insert(0, %r); from recsys2019_deeplearning_evaluation_amd import _native; "
            "_native.load(); print(_native.device_count())" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if res.returncode != 0:
        raise SystemExit("bench.py: cannot count the HIP devices (%s)" % (res.stderr.strip().splitlines() or ["no output"])[-1])
    return int(res.stdout.strip().splitlines()[-1])


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here -- one process per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment exactly as torch.distributed.run would set them -- relay
    rank 0's JSON line (it inherits this process's stdout), and fail if any rank fails.  Never falls back to fewer GPUs: asking for
    more ranks than there are devices is an error, so a line that says n_gpus = N was produced by N devices (the communicator census
    inside every rank is the second check).  Returns the exit code of the job."""
    import socket
    import subprocess
    n = args.gpus
    share = os.environ.get("BENCH_SHARE_GPU") == "1"          # dry run of the N > 1 path on one device (gloo transport)
    have = visible_devices()
    if have < (1 if share else n):
        print("bench.py: --gpus %d needs %d HIP devices, %d visible; refusing to run on fewer" % (n, n, have), file=sys.stderr)
        return 2
    with socket.socket() as sock:                              # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if rank == 0 else subprocess.DEVNULL       # one JSON line on stdout: rank 0's
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, text=True))

    def relay(stream):
        # rank 0's JSON line goes to stdout; anything a library prints there (gloo's connection banner) goes to stderr
        for line in stream:
            print(line, end="", file=sys.stdout if line.lstrip().startswith("{") else sys.stderr, flush=True)
    pump = threading.Thread(target=relay, args=(children[0].stdout,), daemon=True)
    pump.start()
    code = 0
    pending = dict(enumerate(children))
    while pending:
        for rank, child in list(pending.items()):
            rc = child.poll()
            if rc is None:
                continue
            del pending[rank]
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % (rank, rc), file=sys.stderr)
                for other in pending.values():                 # exactly the processes started above
                    other.terminate()
        time.sleep(0.05)
    pump.join(timeout=10)
    return code


class Net:
    """Barrier + max-over-ranks over whichever transport the run uses (torch.distributed or RCCL through ctypes)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.dist = self.torch = self.comm = None
        if self.world == 1:
            return
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("BENCH_SHARE_GPU") == "1":      # dry run of the N > 1 path on one GPU (gloo only: RCCL refuses shared devices)
            self.local_rank = 0
        if os.environ.get("BENCH_TRANSPORT", "torch") == "rccl":
            return                                        # communicator is created after the device is bound (see attach)
        # torch first: its bundled HIP runtime (same SONAME) is then the one libmi355rec.so binds to
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl" or torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        else:
            dist.init_process_group(backend)

    def attach(self):
        """After _native.set_device(): the ctypes RCCL communicator, if that transport was asked for."""
        if self.world > 1 and self.dist is None:
            from recsys2019_deeplearning_evaluation_amd.rccl_direct import RcclCommunicator
            from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
            self.comm = RcclCommunicator(self.rank, self.world)
            self._one, self._all = DeviceArray(2), DeviceArray(2 * self.world)

    def _gather_double(self, x):
        import numpy as np
        from recsys2019_deeplearning_evaluation_amd import _native as N
        word = np.array([x], np.float64).view(np.int32)
        N.check(N.load().mi355rec_device_memcpy(self._one.ptr, N.ptr(word), 8, 1))
        self.comm.all_gather_words(self._one.address(), self._all.address(), 2)
        return self._all.to_host().view(np.float64)

    def barrier(self):
        if self.dist is not None:
            self.torch.cuda.synchronize()
            self.dist.barrier()
        elif self.comm is not None:
            self._gather_double(0.0)

    def barrier_host(self):
        """Barrier that never touches a device (the launch-only check)."""
        if self.dist is not None:
            self.dist.barrier()

    def max(self, x):
        if self.dist is not None:
            t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.dist.get_backend() == "gloo" else "cuda")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        if self.comm is not None:
            return float(self._gather_double(x).max())
        return x

    def census(self):
        """What the transport itself says about the job: number of ranks in the communicator and the set of ranks an all-gather
        reaches.  Raises if that is not WORLD_SIZE distinct ranks (so that a mis-launched job cannot report an N-GPU number)."""
        import numpy as np
        info = {"world_size_env": self.world, "transport": "none"}
        if self.world == 1:
            info["ranks_seen"] = 1
            return info
        if self.comm is not None:
            info.update(transport="rccl-ctypes", nccl_comm_count=self.comm.count(), nccl_user_rank=self.comm.user_rank())
            seen = sorted(int(round(x)) for x in self._gather_double(float(self.rank)))
            count = info["nccl_comm_count"]
        else:
            t = self.torch.tensor([float(self.rank)], dtype=self.torch.float64, device="cpu" if self.dist.get_backend() == "gloo" else "cuda")
            parts = [self.torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(parts, t)
            seen = sorted(int(round(float(x.item()))) for x in parts)
            count = self.dist.get_world_size()
            info.update(transport="torch." + self.dist.get_backend(), torch_world_size=count)
        info["ranks_seen"] = len(set(seen))
        if count != self.world or seen != list(range(self.world)):
            raise SystemExit("bench.py: the communicator reaches ranks %s (count %d) but WORLD_SIZE is %d" % (seen, count, self.world))
        return info

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
        elif self.comm is not None:
            self.barrier()
            self.comm.close()


def ials_section(urm, net, args, extra):
    """BASELINE config 5: IALS k = 200 on the ML-20M shape with the row solves of each half-step split over the ranks
    (sharding.ShardedIALSEpoch: cost-balanced ranges, one all-gather of the solved rows per half-step).  At N = 1 it also runs the
    ranges of the 8-way split one after the other and models the two all-gathers from their size."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import IALS_MI355X_Epoch
    from recsys2019_deeplearning_evaluation_amd.sharding import ShardedIALSEpoch, ials_row_ranges
    k = 200
    conf = urm.copy()
    conf.data = (1.0 + 1.0 * conf.data).astype(np.float32)
    V0 = k ** -0.5 * np.random.default_rng(0).random((urm.shape[1], k))
    ia = IALS_MI355X_Epoch(conf, k, 1e-3, V0)
    job = ShardedIALSEpoch(ia, conf, net.dist, net.rank, net.world, net.comm)
    job.run_epoch()                                   # warm-up (first touch of the buffers, communicator)
    best = None
    for _ in range(2):
        net.barrier()
        t0 = time.perf_counter()
        job.run_epoch()
        net.barrier()
        dt = net.max(time.perf_counter() - t0)
        best = dt if best is None or dt < best else best
    nnz, n_u, n_i = float(urm.nnz), urm.shape[0], urm.shape[1]
    flops = 2.0 * (2.0 * nnz * k * k) + (n_u + n_i) * (k ** 3 / 3.0 + 2.0 * k * k)
    block = {"seconds_per_epoch": best, "definition": "both half-steps; every rank's device holds the complete updated U and V (max over ranks, best of 2)",
             "n_factors": k, "TFLOPs_algorithmic": flops / best / 1e12, "frac_of_fp64_peak_all_gpus": flops / best / 1e12 / (FP64_PEAK_TF * net.world),
             "exchange_bytes_per_rank_per_epoch": job.exchange_bytes_per_rank_per_epoch(),
             "rows_this_rank": [int(job.user_ranges[net.rank][1] - job.user_ranges[net.rank][0]),
                                int(job.item_ranges[net.rank][1] - job.item_ranges[net.rank][0])]}
    job.close()
    if net.world == 1 and not args.no_extras:
        G = 8
        ur, ir = ials_row_ranges(conf, G, k)
        t_user, t_item = [], []
        for r in range(G):
            ia.user_half(*ur[r]); ia.synchronize(); t_user.append(ia.stats()["call_ms"])
        for r in range(G):
            ia.item_half(*ir[r]); ia.synchronize(); t_item.append(ia.stats()["call_ms"])
        slab_u = max(e - s for s, e in ur) * k * 8
        slab_i = max(e - s for s, e in ir) * k * 8
        ring = lambda slab: (G - 1) * slab / 50e9 * 1e3 + 0.05
        direct = lambda slab: slab / 50e9 * 1e3 + 0.05
        # what ShardedIALSEpoch runs at N > 1: a rank's user rows in 4 pieces, the all-gather of a finished piece behind the solve of the
        # next one; its item rows in one piece (four pieces of an item range cost 10.3 ms against 7.6 ms in one go: the longest-first order
        # inside a range keeps its tail short).  The pieces of the slowest range are measured, the gathers modelled (piece c's starts
        # when both its solve and piece c - 1's gather are done).
        from recsys2019_deeplearning_evaluation_amd.sharding import chunk_bounds

        def pieces_of(half, rng, CH):
            out = []
            for r0, r1 in chunk_bounds(rng[1] - rng[0], CH):
                if r1 > r0:
                    half(rng[0] + r0, rng[0] + r1); ia.synchronize(); out.append(ia.stats()["call_ms"])
            return out

        def pipelined(pieces, gather_ms):
            t_solve = t_net = 0.0
            for p_ms in pieces:
                t_solve += p_ms
                t_net = max(t_net, t_solve) + gather_ms / len(pieces)
            return t_net

        pu = pieces_of(ia.user_half, ur[int(np.argmax(t_user))], 4)
        pi = pieces_of(ia.item_half, ir[int(np.argmax(t_item))], 1)
        one_ring_end = max(t_user) + max(t_item) + ring(slab_u) + ring(slab_i)
        all_links_end = max(t_user) + max(t_item) + direct(slab_u) + direct(slab_i)
        one_ring = pipelined(pu, ring(slab_u)) + pipelined(pi, ring(slab_i))
        all_links = pipelined(pu, direct(slab_u)) + pipelined(pi, direct(slab_i))
        block["emulated_8_way"] = {
            "user_half_ms_per_range": t_user, "item_half_ms_per_range": t_item, "slowest_user_ms": max(t_user), "slowest_item_ms": max(t_item),
            "kernel_speedup_vs_1gpu": best * 1e3 / (max(t_user) + max(t_item)),
            "slab_MB_per_rank": {"users": slab_u / 1e6, "items": slab_i / 1e6},
            "modelled_allgather_ms": {"one_ring_50GBps_per_link": ring(slab_u) + ring(slab_i), "seven_links_at_once": direct(slab_u) + direct(slab_i)},
            "pieces_ms_of_the_slowest_range": {"users": pu, "items": pi},
            "predicted_seconds_per_epoch_one_exchange_at_the_end": {"one_ring": one_ring_end * 1e-3, "seven_links": all_links_end * 1e-3},
            "predicted_seconds_per_epoch": {"one_ring": one_ring * 1e-3, "seven_links": all_links * 1e-3},
            "predicted_speedup": {"one_ring": best * 1e3 / one_ring, "seven_links": best * 1e3 / all_links},
            "note": "the 8 cost-balanced ranges of each half-step run one after the other on ONE GPU (measured), the slowest range again in the "
                    "pieces ShardedIALSEpoch runs it in (users 4, items 1; measured); the all-gathers are modelled from their size, a piece's "
                    "gather behind the next piece's solve; unmeasured on hardware"}
    ia.close()
    extra["ials"] = block


def itemknn_section(urm, net, args, extra, key="itemknn"):
    """The ItemKNN cosine build at this N: constructor, sharded build (device-resident result on every rank), download on rank 0,
    roofline blocks of the column kernel, and -- at N = 1 -- the per-range kernel times of the 8-way split."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
    from recsys2019_deeplearning_evaluation_amd.sharding import ShardedSimilarityBuild, similarity_column_ranges
    world, rank = net.world, net.rank
    n_items = urm.shape[1]
    # constructor = H2D of the URM + all of the set-up on the device (CSC view, profile stream, norms, costs): timed
    # because `ItemKNNCFRecommender.fit` pays it, like the reference's __init__
    sim = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
    sim.close()
    t_c = time.perf_counter()
    sim = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
    sim.synchronize()
    create_s = time.perf_counter() - t_c
    # the same constructor from a URM that is already in HBM (uploaded once per search, ResidentURM): the `fit` of the bench
    # contract (inputs resident when the timed region starts); the PCIe-inclusive figure stays next to it
    from recsys2019_deeplearning_evaluation_amd import ResidentURM
    resident = ResidentURM(urm)
    create_resident_s = None
    for _ in range(3):
        sim.close()
        t_c = time.perf_counter()
        sim = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine", resident=resident)
        sim.synchronize()
        dt = time.perf_counter() - t_c
        create_resident_s = dt if create_resident_s is None else min(create_resident_s, dt)
    resident.close()
    costs = sim.column_costs()
    # partition + buffers once, outside the timed region.  BENCH_SIM_EXCHANGE=gather: only rank 0 receives (north_star's wording; the default
    # all-gather leaves the result on every rank, which is what this block's `definition` has said since round 1)
    job = ShardedSimilarityBuild(sim, net.dist, rank, world, net.comm, exchange=os.environ.get("BENCH_SIM_EXCHANGE", "allgather"))
    my_columns = job.columns[rank]
    best, kernel_ms = None, None
    for rep in range(4):
        net.barrier()
        t1 = time.perf_counter()
        k_ms = job.build()                       # kernel on this rank's columns + the all-gather: result resident on every device
        net.barrier()
        dt = net.max(time.perf_counter() - t1)
        if rep > 0 and (best is None or dt < best):       # the first repetition warms the communicator up
            best, kernel_ms = dt, k_ms                    # (k_ms: the column kernel's time summed over ALL pieces of this rank's build)
    t2 = time.perf_counter()
    idx, val = job.download() if rank == 0 else (None, None)
    download_s = time.perf_counter() - t2
    # north_star's wording of the exchange -- "a final RCCL gather": only rank 0 ends with every column (sharding exchange="gather": each
    # peer sends its slab over its own xGMI link) -- timed next to the all-gather above, whichever of the two this block's headline figure is
    other_s = other_err = None
    other_kind = "allgather" if job.root is not None else "gather"
    if world > 1 and os.environ.get("BENCH_SIM_BOTH_EXCHANGES", "1") != "0":
        try:
            job2 = ShardedSimilarityBuild(sim, net.dist, rank, world, net.comm, exchange=other_kind)
            for rep in range(3):
                net.barrier()
                t1 = time.perf_counter()
                job2.build()
                net.barrier()
                dt = net.max(time.perf_counter() - t1)
                if rep > 0 and (other_s is None or dt < other_s):
                    other_s = dt
            if rank == 0 and idx is not None:
                idx2, val2 = job2.download()
                if not (np.array_equal(idx2, idx) and np.array_equal(val2, val)):
                    other_err = "the two exchanges assembled different results"
            job2.close()
        except Exception as exc:                      # (recorded, not fatal: the all-gather figure above stands)
            other_err = repr(exc)
    sst = sim.stats()
    # the bound of the accumulation, measured on THIS device in THIS run (the round-1 microbenchmark figure is the fall-back)
    global LDS_ATOMIC_PEAK
    fixed_peak = 21.6e9 * 256
    peak_source = "profiles/r1_lds_atomics_microbench.txt (21.6 lane-adds per ns and CU x 256)"
    try:
        from recsys2019_deeplearning_evaluation_amd import _native as _N
        measured = _N.lds_atomic_rate()
        if measured > fixed_peak:                      # the stricter (higher) of the two peaks is the one the fraction is taken of
            LDS_ATOMIC_PEAK = measured
            peak_source = "mi355rec_lds_atomic_rate in this run: ds_add_u32 on uniformly random cells, one 1024-thread workgroup per CU"
        else:
            LDS_ATOMIC_PEAK = fixed_peak
            peak_source += "; live measurement in this run was lower (%.3e)" % measured
    except Exception as exc:
        peak_source += "; live measurement failed: %r" % (exc,)
    pairs = float(np.asarray(costs, dtype=np.float64)[my_columns].sum())
    pair_rate = pairs / (kernel_ms * 1e-3)
    alg_gbps = 8.0 * pairs / (kernel_ms * 1e-3) / 1e9
    block = {
        "cosine_build_s": best, "definition": "kernel on this rank's columns (interleaved partition: equal counts, equal cost) + one "
                                              "all-gather; full (n_cols x topK) result resident on every rank's device (max over ranks, best of 3)",
        "create_s": create_s, "fit_s": create_s + best,
        "fit_definition": "constructor from the HOST URM (PCIe upload + device set-up) + build: the same meaning as in rounds 1-4; Python host code included",
        "create_incl_pcie_upload_s": create_s, "fit_incl_pcie_upload_s": create_s + best,
        "create_resident_s": create_resident_s, "fit_resident_s": create_resident_s + best,
        "fit_resident_definition": "constructor from the device-resident URM (ResidentURM: uploaded once per search; best of 3) + build",
        "download_to_host_rank0_s": download_s,
        "topK": TOPK, "columns_this_rank": int(len(my_columns)), "kernel_ms_this_rank": kernel_ms,
        "exchange_bytes_per_rank": job.exchange_bytes_per_rank(),
        "cosine_build_other_exchange_s": other_s, "other_exchange": None if world == 1 else ("gather to rank 0" if other_kind == "gather" else "all-gather"),
        "other_exchange_error": other_err,
        "exchange": "none" if world == 1 else ("%s of %d cost-sized pieces, %s" % ("gather to rank 0" if job.root is not None else "all-gather", len(job.rows),
                                                "6-byte cells (float32 value + 16-bit id)" if job.packed else "8-byte cells")),
        "transport": "none" if world == 1 else ("rccl-ctypes" if net.comm is not None else "torch." + net.dist.get_backend()),
        "nnz_out": int((idx >= 0).sum()) if idx is not None else None,
        "roofline": {"bound": "lds-atomics", "kernel": "sim_packed_kernel + sim_column_kernel (one timed region: the packed-counts launch and the 32-bit launch behind it)",
                     "achieved": pair_rate, "peak": LDS_ATOMIC_PEAK,
                     "unit": "pair-adds/s", "frac": pair_rate / LDS_ATOMIC_PEAK, "peak_source": peak_source, "pairs_this_rank": pairs,
                     "peak_fixed_round1": fixed_peak, "frac_of_fixed_round1_peak": pair_rate / fixed_peak,
                     "stream_GBps": 2.0 * pairs / (kernel_ms * 1e-3) / 1e9,
                     "survey_8d_algorithmic_GBps": alg_gbps, "survey_8d_algorithmic_over_hbm_peak": alg_gbps / HBM_PEAK_GBPS,
                     "note": "SURVEY 8(d)'s byte model (8 B per co-occurrence pair) exceeds the HBM peak because the kernel streams "
                             "2-byte ids from L2/MALL; the bound that holds is the LDS atomic rate"}}
    if not (0.0 < block["roofline"]["frac"] <= 1.0):
        # a fraction above 1 means the timed kernel is not the work that was divided by it: never print such a block
        block["roofline"] = {"error": "refused: frac %.3f outside (0, 1] (kernel_ms %.4f for %.3e pairs)" % (block["roofline"]["frac"], kernel_ms, pairs)}
    if world == 1 and not args.no_extras:
        # the 8-GPU build of BASELINE config 4, one part after the other on this GPU: kernel time per part (measured), the
        # gathered buffer assembled with device copies (measured: what the last ring step leaves behind), and the exchange
        # itself modelled from its size two ways -- NOT measured on hardware until the driver's 8-GPU run exists
        from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
        G = 8
        widest8 = -(-n_items // G)
        slab_words = 2 * widest8 * TOPK
        local8, gathered8 = DeviceArray(slab_words), DeviceArray(G * slab_words)
        from recsys2019_deeplearning_evaluation_amd.sharding import cost_sized_pieces, chunk_bounds, packed_words, piece_order, FIXED_PAIRS_PER_CELL
        per_part, per_piece = [], []
        t_copy = 0.0
        # what ShardedSimilarityBuild does at world > 1: 4 pieces per part sized by cost (the cheap rows in equal-cost pieces first, the small
        # head of the most expensive columns last: only the last piece's exchange is exposed), each packed into 6-byte cells (n_items <=
        # 65 535) by a small kernel behind the piece's column kernel
        cols0 = sim.part_columns(0, G)
        row_cost = np.zeros(widest8)
        row_cost[:len(cols0)] = np.asarray(costs, np.float64)[cols0] + FIXED_PAIRS_PER_CELL * n_items
        pieces_in_row_order = cost_sized_pieces(row_cost, 4)
        pieces = [pieces_in_row_order[c] for c in piece_order(pieces_in_row_order)]     # in BUILD order: cheapest rows first, the small head last
        pieces_by_count = chunk_bounds(widest8, 4)
        can_pack = n_items <= 65535
        packed8 = DeviceArray(packed_words(widest8 * TOPK) + 4)
        pack_ms = []
        for r in range(G):
            sim.compute_part_device(r, G, local8.address(), local8.address(widest8 * TOPK))
            sim.synchronize()
            per_part.append(sim.stats()["kernel_ms"])
            t3 = time.perf_counter()
            gathered8.copy_from_device(local8, slab_words, r * slab_words)
            t_copy += time.perf_counter() - t3
            n_mine = len(sim.part_columns(r, G))
            row = []
            for r0, r1 in pieces:
                cnt = max(0, min(r1, n_mine) - r0)
                if cnt:
                    sim.compute_part_chunk_device(r, G, r0, cnt, local8.address(2 * r0 * TOPK), local8.address(2 * r0 * TOPK + (r1 - r0) * TOPK))
                    sim.synchronize()
                    k_ms = sim.stats()["kernel_ms"]
                    if can_pack:                          # (wall time of the pack launch, blocking: an upper bound of what it adds to a piece)
                        t3 = time.perf_counter()
                        sim.pack_slab_device(local8.address(2 * r0 * TOPK), local8.address(2 * r0 * TOPK + (r1 - r0) * TOPK), (r1 - r0) * TOPK, packed8.address())
                        sim.synchronize()
                        pack_ms.append((time.perf_counter() - t3) * 1e3)
                        k_ms += pack_ms[-1]
                    row.append(k_ms)
                else:
                    row.append(0.0)
            per_piece.append(row)
        cell_bytes = 6 if can_pack else 8
        slab = cell_bytes * widest8 * TOPK
        ring_of = lambda nbytes: (G - 1) * nbytes / 50e9 * 1e3 + 0.05     # ONE ring over one xGMI link direction (~50 GB/s effective) + launch latency
        direct_of = lambda nbytes: nbytes / 50e9 * 1e3 + 0.05             # all 7 links at once (every peer is one hop away on the xGMI mesh)
        ring_ms, direct_ms = ring_of(slab), direct_of(slab)

        def overlapped(row, gather_of):
            """kernel of piece c + 1 hides the exchange of piece c; what sticks out is added (the last piece's exchange always does)"""
            t = 0.0
            for c, k_ms in enumerate(row):
                g_prev = gather_of(slab * (pieces[c - 1][1] - pieces[c - 1][0]) / widest8) if c else 0.0
                t += max(k_ms, g_prev)
            return t + gather_of(slab * (pieces[-1][1] - pieces[-1][0]) / widest8)
        ranges8 = similarity_column_ranges(sim, 8)
        from recsys2019_deeplearning_evaluation_amd.sharding import default_chunks
        part_pairs = (float(np.asarray(costs, np.float64).sum()) + FIXED_PAIRS_PER_CELL * float(n_items) ** 2) / G
        chunks_allgather, chunks_gather = default_chunks(part_pairs, slab, G, "allgather"), default_chunks(part_pairs, slab, G, "gather")
        block["emulated_8_way"] = {
            "partition": "interleaved (serpentine deal of the cost order): %d columns and 1/8 of the cost per part" % widest8,
            "kernel_ms_per_part": per_part, "slowest_part_ms": max(per_part),
            "kernel_speedup_vs_1gpu": kernel_ms / max(per_part),
            "cell_bytes": cell_bytes, "slab_MB_per_rank": slab / 1e6, "slab_MB_per_rank_8_byte_cells": 8 * widest8 * TOPK / 1e6,
            "slab_MB_per_rank_with_contiguous_ranges": 8 * max(b - a for a, b in ranges8) * TOPK / 1e6,
            "device_copy_of_8_slabs_ms": t_copy * 1e3,
            "pieces_rows_in_build_order": [b - a for a, b in pieces], "pieces_first_row_in_build_order": [a for a, b in pieces],
            "pieces_rows_by_count_until_round_5": [b - a for a, b in pieces_by_count],
            "pack_kernel_wall_ms_per_piece": (sum(pack_ms) / len(pack_ms)) if pack_ms else None,
            "modelled_allgather_ms": {"one_ring_50GBps_per_link": ring_ms, "seven_links_at_once": direct_ms},
            "modelled_gather_to_root_ms": direct_ms,
            "predicted_build_speedup_one_exchange_at_the_end": {"one_ring": (best * 1e3) / (max(per_part) + ring_ms),
                                                                "seven_links": (best * 1e3) / (max(per_part) + direct_ms)},
            "kernel_ms_per_piece": per_piece,
            "predicted_build_speedup_in_4_pieces": {"one_ring": (best * 1e3) / max(overlapped(row, ring_of) for row in per_piece),
                                                    "seven_links": (best * 1e3) / max(overlapped(row, direct_of) for row in per_piece)},
            # what the library does by default (sharding.default_chunks: pieces only where the modelled exchange is long next to the kernel)
            "default_chunks": {"allgather": chunks_allgather, "gather_to_root": chunks_gather},
            "predicted_build_speedup": {
                "one_ring": (best * 1e3) / (max(overlapped(row, ring_of) for row in per_piece) if chunks_allgather > 1 else max(per_part) + ring_ms),
                "seven_links": (best * 1e3) / (max(overlapped(row, direct_of) for row in per_piece) if chunks_allgather > 1 else max(per_part) + direct_ms),
                "gather_to_root": (best * 1e3) / (max(overlapped(row, direct_of) for row in per_piece) if chunks_gather > 1 else max(per_part) + direct_ms)},
            "exchange": "4 cost-sized pieces per part built cheapest rows first (kernel_ms_per_piece is in that order), %d-byte cells, the exchange of a finished piece behind the kernel of the next one "
                        "(ShardedSimilarityBuild); all-gather = every rank ends with W (one ring, or all seven links at once); gather_to_root = only "
                        "rank 0 does (exchange='gather': each peer sends over its own link, the model of 'seven_links')" % cell_bytes,
            "note": "parts run one after the other on ONE GPU; the exchange is modelled from its size (7 x %.2f MB), unmeasured on hardware; kernel_ms_per_piece "
                    "includes the blocking wall time of the piece's pack launch" % (slab / 1e6)}
        packed8.close()
        local8.close(); gathered8.close()
    job.close()
    sim.close()
    block["shape"] = "%dx%d nnz=%d" % (urm.shape[0], urm.shape[1], urm.nnz)
    extra[key] = block
    if key == "itemknn":     # flat aliases kept for continuity with round 1's line
        extra.update({"itemknn_cosine_build_s": best, "itemknn_create_s": create_s, "itemknn_fit_s": create_s + best,
                      "itemknn_create_resident_s": create_resident_s, "itemknn_fit_resident_s": create_resident_s + best,
                      "itemknn_kernel_ms_this_rank": kernel_ms, "itemknn_frac_of_lds_atomic_peak": block["roofline"].get("frac")})
    return costs


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))                   # the ranks come back through main() with WORLD_SIZE set
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)      # a stall leaves its stack in the log
    if os.environ.get("BENCH_LAUNCH_ONLY") == "1" and os.environ.get("BENCH_TEST_BREAK_RANK") == os.environ.get("RANK"):
        raise SystemExit(3)                            # (launcher test: a rank that dies before the rendezvous)
    net = Net(args)
    rank, world = net.rank, net.world
    if os.environ.get("BENCH_LAUNCH_ONLY") == "1":
        # launcher + communicator check without a device (tests/test_bench_launcher.py): census, one line, no measurement
        census = net.census()
        net.barrier_host()
        if rank == 0:
            print(json.dumps({"launch_only": True, "n_gpus": world, "communicator": census, "launched_by": os.environ.get("BENCH_LAUNCHED_BY", "external")}))
        net.close()
        faulthandler.cancel_dump_traceback_later()
        return
    import numpy as np  # noqa: F401
    from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch, _native
    _native.load()
    if _native.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    _native.set_device(net.local_rank)
    net.attach()

    note("device %d bound, communicator census" % net.local_rank)
    census = net.census()            # fails loudly if the communicator does not reach WORLD_SIZE distinct ranks
    # BASELINE config 4 (the Netflix shape: the build the >= 6x target at 8 GPUs is written for) belongs in the default line too, but its
    # synthetic URM takes ~45 s of host time to generate: a CHILD process writes it to the URM cache while this process measures on the
    # GPU (the timed regions here are graph replays and kernels with an idle host; the CPU baseline legs run after the child is done).
    netflix_child = None
    if (world == 1 and args.workload == "ml20m" and not args.no_sim and not args.no_extras and not args.no_netflix
            and os.environ.get("BENCH_NO_NETFLIX_CHILD") != "1"):
        import subprocess
        netflix_child = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; bench.load_urm('netflix')" % ROOT],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    urm = load_urm_once_per_node(args.workload, net)
    n_users, n_items = urm.shape
    per_epoch = (n_users // BATCH + 1) * BATCH
    n_batches = per_epoch // BATCH

    note("URM %s ready: %d x %d, %d interactions" % (args.workload, n_users, n_items, urm.nnz))
    # ------------------------------------------------------------------ BPR-MF epochs (headline value)
    mf = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH,
                                          learning_rate=1e-3, sgd_mode="sgd", init_std_dev=0.1, random_seed=42 + rank)
    if args.warmup > 0:
        mf.epochIteration_Cython(args.warmup)
    net.barrier()
    t0 = time.perf_counter()
    mf.epochIteration_Cython(args.steps)            # blocking: returns after the stream has drained; pure hipGraph replay
    net.barrier()
    elapsed = time.perf_counter() - t0
    st = mf.stats()
    elapsed = net.max(elapsed)
    total_samples = args.steps * per_epoch * world
    value = total_samples / elapsed
    # per-launch duration of the dominant kernel: a SEPARATE, untimed call whose mini-batch launches carry their own HIP
    # start/stop events on the handle's stream (plain launches instead of graph replay; not part of `value`)
    mf.set_profiling(5 * n_batches)
    mf.epochIteration_Cython(5)
    pst = mf.stats()
    mf.set_profiling(0)
    avg_launch_s = (pst["kernel_ms"] / max(1, pst["n_timed"])) * 1e-3
    bytes_per_launch = st["algorithmic_bytes"] / max(1, st["n_launches"])
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    traffic, traffic_source = pmc_traffic("mf_batch_kernel")
    roofline = {"bound": "hbm", "kernel": "mf_batch_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_launch_s * 1e6,
                "timed_launches": pst["n_timed"],
                "whole_epoch_achieved_GBps": st["algorithmic_bytes"] / (st["call_ms"] * 1e-3) / 1e9,
                "whole_epoch_frac": st["algorithmic_bytes"] / (st["call_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    mf.close()

    note("headline: %.1f M samples/s" % (value / 1e6))
    extra = {"bpr_loss_per_sample": st["loss"] / max(1, st["n_units"]), "bpr_stream_ms": st["call_ms"], "communicator": census,
             "launched_by": os.environ.get("BENCH_LAUNCHED_BY", "external launcher" if world > 1 else "single process")}
    costs = None
    if not args.no_sim:
        costs = itemknn_section(urm, net, args, extra)
        extra["itemknn"]["communicator"] = census
        if world > 1 and args.workload != "netflix" and not args.no_netflix:
            # BASELINE config 4 is the build with the >= 6x target at 8 GPUs: always part of a multi-GPU run
            big = load_urm_once_per_node("netflix", net)
            itemknn_section(big, net, args, extra, key="itemknn_netflix_config4")
            del big
    note("itemknn sections done")
    if not args.no_ials:
        try:
            ials_section(urm, net, args, extra)
        except Exception as exc:
            if world > 1:
                raise
            extra["ials_error"] = repr(exc)

    shape_name = {"ml20m": "ML-20M-shaped", "ml1m": "ML-1M-shaped", "netflix": "Netflix-Prize-shaped (BASELINE configs[3])"}[args.workload]
    out = {"metric": "BPR-MF SGD samples/sec (k=128, batch 1000) + ItemKNN cosine build sec on %s URM" % shape_name,
           "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BPR-MF epoch (%d mini-batches x 1000 on-device samples), k=128, sgd, on synthetic %s URM "
                                  "%dx%d nnz=%d; replicas (one independent model per GPU)" % (n_batches, args.workload, n_users, n_items, urm.nnz),
                      "batch_size": BATCH, "n_factors": K_FACTORS, "parallelism": "replicas x%d" % world},
           "roofline": roofline, "extra": extra}

    note("ials section done")
    cpu_jobs = []
    if rank == 0 and world == 1 and not args.no_extras and not args.no_paths:
        out["extra"]["paths"] = {}
        try:
            other_paths(urm, args, out["extra"]["paths"], cpu_jobs)
        except Exception as exc:                       # the headline line must survive a failure in the side measurements, and so
            out["extra"]["paths_error"] = repr(exc)    # must the paths measured before it
        note("other paths done")
        try:
            fit_rows(urm, out["extra"]["paths"])
        except Exception as exc:
            out["extra"]["fit_rows_error"] = repr(exc)
        note("recommender fit() rows done")
    if netflix_child is not None:
        try:
            netflix_child.wait(timeout=float(os.environ.get("BENCH_NETFLIX_WAIT_S", "90")))
            if netflix_child.returncode != 0:
                raise RuntimeError("the URM generator exited with code %s" % netflix_child.returncode)
            big = load_urm("netflix")
            slim_args = argparse.Namespace(**dict(vars(args), no_extras=False))
            itemknn_section(big, net, slim_args, out["extra"], key="itemknn_netflix_config4")
            del big
            note("netflix-shape itemknn section done")
        except Exception as exc:                        # the headline line must survive: the section is reported as skipped
            try:
                netflix_child.kill()
                netflix_child.wait(timeout=10)
            except Exception:
                pass
            out["extra"]["itemknn_netflix_config4"] = {"skipped": repr(exc)}
    # ---- CPU baseline legs: the LAST thing the run does -- every GPU measurement is finished and the generator child has exited
    # (or was killed), so each leg has the host to itself, one after the other
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline_bpr(urm, args.cpu_seconds)
        base["host_cpu_count"] = os.cpu_count()
        out["cpu_baseline"] = base
        out["extra"]["speedup_vs_cpu_baseline"] = value / base["value"]
        if costs is not None:
            sb = cpu_baseline_sim(urm, costs, args.cpu_seconds)
            ik = out["extra"]["itemknn"]
            ik["cpu_baseline"] = sb
            ik["speedup_vs_cpu_baseline"] = sb["value"] / ik["cosine_build_s"]
            ik["fit_speedup_vs_cpu_baseline"] = sb["value"] / ik["fit_s"]                 # (upload-inclusive fit)
            ik["fit_resident_speedup_vs_cpu_baseline"] = sb["value"] / ik["fit_resident_s"]
            rf = out["extra"].get("paths", {}).get("itemknn_recommender_fit")
            if rf:
                rf["cpu_baseline"] = sb
                rf["speedup_vs_cpu_baseline"] = sb["value"] / rf["value"]
        for tag, job in cpu_jobs:
            note("cpu leg: %s" % tag)
            try:
                job()
            except Exception as exc:
                out["extra"].setdefault("cpu_legs_error", "")
                out["extra"]["cpu_legs_error"] += "%s: %r; " % (tag, exc)
    # one compact row per path at the TOP level (value, unit, fraction of the bound, CPU baseline of the same run)
    table = {"bpr_mf_k128_batch1000_one_model": {"value": value, "unit": "samples/s", "bound": "hbm", "frac": roofline["frac"],
                                                 "cpu_value": out.get("cpu_baseline", {}).get("value"), "cpu_kind": out.get("cpu_baseline", {}).get("kind")}}
    for name, blk in out["extra"].get("paths", {}).items():
        if not isinstance(blk, dict):
            continue
        if "value" in blk and "unit" in blk and "definition" in blk:          # (the end-to-end fit() rows)
            table[name] = {"value": blk["value"], "unit": blk["unit"], "bound": "end-to-end", "cpu_value": (blk.get("cpu_baseline") or {}).get("value"),
                           "cpu_kind": (blk.get("cpu_baseline") or {}).get("kind")}
            continue
        if "samples_per_s" not in blk and "users_per_s" not in blk and "seconds_per_epoch" not in blk:
            continue                                                             # (a block without a rate: modelled figures only)
        val = blk.get("samples_per_s", blk.get("users_per_s", blk.get("seconds_per_epoch")))
        table[name] = {"value": val, "unit": "samples/s" if "samples_per_s" in blk else ("users/s" if "users_per_s" in blk else "s/epoch"),
                       "bound": blk.get("bound"), "frac": blk.get("frac"), "cpu_value": (blk.get("cpu_baseline") or {}).get("value"),
                       "cpu_kind": (blk.get("cpu_baseline") or {}).get("kind")}
        fx = blk.get("cpu_baseline_reference_fixture")          # the reference's own class, timed where /root/reference exists (committed fixture)
        if fx:
            table[name].update({"cpu_reference_fixture_value": fx.get("value"), "cpu_reference_fixture_kind": fx.get("kind")})
    if "itemknn" in out["extra"]:
        ik = out["extra"]["itemknn"]
        table["itemknn_cosine_top100"] = {"value": ik.get("fit_incl_pcie_upload_s"), "unit": "s (upload + constructor + build)", "build_s": ik.get("cosine_build_s"),
                                          "fit_resident_s": ik.get("fit_resident_s"), "build_gather_to_rank0_s": ik.get("cosine_build_other_exchange_s"),
                                          "bound": "lds-atomics", "frac": ik.get("roofline", {}).get("frac"),
                                          "cpu_value": (ik.get("cpu_baseline") or {}).get("value"), "cpu_kind": (ik.get("cpu_baseline") or {}).get("kind")}
    nf = out["extra"].get("itemknn_netflix_config4")
    if isinstance(nf, dict) and "fit_s" in nf:
        e8 = nf.get("emulated_8_way", {})
        table["itemknn_cosine_top100_netflix_shape_config4"] = {
            "value": nf.get("fit_incl_pcie_upload_s"), "unit": "s (upload + constructor + build)", "build_s": nf.get("cosine_build_s"),
            "fit_resident_s": nf.get("fit_resident_s"), "build_gather_to_rank0_s": nf.get("cosine_build_other_exchange_s"),
            "predicted_8_gpu_one_ring": (e8.get("predicted_build_speedup") or {}).get("one_ring"),
            "bound": "lds-atomics", "frac": nf.get("roofline", {}).get("frac"),
            "emulated_8_way_kernel_speedup": e8.get("kernel_speedup_vs_1gpu"),
            "predicted_8_gpu_build_speedup": e8.get("predicted_build_speedup"),
            "predicted_8_gpu_build_speedup_one_exchange_at_the_end": e8.get("predicted_build_speedup_one_exchange_at_the_end"),
            "note": "parts run one after the other on ONE GPU, exchange modelled from its size: unmeasured on hardware"}
    out["paths"] = table
    note("done")
    faulthandler.cancel_dump_traceback_later()
    if rank == 0:
        emit(out)
    net.close()


if __name__ == "__main__":
    main()
