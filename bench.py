#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: BPR-MF SGD samples/sec (+ ItemKNN cosine build
seconds) on an ML-20M-shaped synthetic URM (138 493 x 26 744, ~20 M interactions), k=128, batch 1000.

  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one reference epoch of MatrixFactorization_BPR_Cython (n_users // batch_size + 1 = 139 mini-batches of
1000 samples drawn on the device, MatrixFactorization_Cython_Epoch.pyx:583).  Inputs (URM, factors) are resident
in HBM before the timed region.  BPR-MF does not shard (every sample reads and writes shared factor rows): with
N GPUs every rank trains an independent replica with its own seed -- how the reference's hyper-parameter search
parallelises (run_parameter_search.py:498) -- so `value` is the aggregate samples/s of N replicas ("weak").  The
ItemKNN cosine build IS sharded (item columns, cost-balanced, one RCCL all-gather); its wall seconds at this N
are reported in `extra` (strong scaling of a fixed build).

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel of the headline metric, the BPR gradient
kernel (mf_batch_kernel): algorithmic bytes of one launch (batch_size x 24*k B) over the average
launch duration measured with the dispatch's own HIP start/stop events in a separate, untimed call of the same run.
`cpu_baseline` is the reference's own compiled Cython kernel (oracle/_ref) on one host core, same URM / k / batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
K_FACTORS = 128
BATCH = 1000
TOPK = 100


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ml20m", choices=["ml20m", "ml1m"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sim", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of each CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the other hot paths (FunkSVD, large-batch BPR, SLIM-BPR, IALS)")
    return ap.parse_args()


def load_urm(name):
    """Synthetic URM of the named shape, cached as .npz under /tmp (generation is not part of any timing)."""
    import numpy as np
    import scipy.sparse as sps
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mi355rec_urm_%s_binary_r%s.npz" % (name, os.environ.get("RANK", "0")))
    if os.path.isfile(cache):
        try:
            return sps.load_npz(cache).tocsr().astype(np.float32)
        except Exception:
            pass
    urm = named_urm(name, "binary")
    try:
        sps.save_npz(cache, urm, compressed=False)
    except Exception:
        pass
    return urm


def pmc_traffic(kernel):
    """HBM bytes per launch from the latest committed PMC collection (profiles/*_pmc_traffic.json, produced by
    scripts/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes of this same workload; FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the counters itself; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


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
    import numpy as np
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


def other_paths(urm):
    """The remaining rows of SURVEY.md section 8 on the same URM shape, one short run each (N = 1 only).  Throughputs come from
    the handle's own stream events (call_ms); fractions are ALGORITHMIC work / time against the MI355X peaks."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import (IALS_MI355X_Epoch, MatrixFactorization_MI355X_Epoch,
                                                        SLIM_BPR_MI355X_Epoch)
    out = {}

    def mf_run(tag, epochs, **kw):
        m = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, learning_rate=1e-3, init_std_dev=0.1, random_seed=7, **kw)
        m.epochIteration_Cython(1)
        m.epochIteration_Cython(epochs)
        st = m.stats()
        sec = st["call_ms"] * 1e-3
        out[tag] = {"samples_per_s": st["n_units"] / sec, "algorithmic_GBps": st["algorithmic_bytes"] / sec / 1e9,
                    "frac_of_hbm_peak": st["algorithmic_bytes"] / sec / 1e9 / HBM_PEAK_GBPS, "epochs": epochs, "seconds": sec}
        m.close()

    # the same BPR epoch at a batch size where a mini-batch fills the chip (the reference's search space stops at 1024)
    mf_run("bpr_mf_k128_batch65536", 200, algorithm_name="MF_BPR", batch_size=65536, sgd_mode="sgd")
    mf_run("bpr_mf_k128_batch1000_adagrad", 50, algorithm_name="MF_BPR", batch_size=BATCH, sgd_mode="adagrad")
    mf_run("funk_svd_k128_batch1000_bias", 1, algorithm_name="FUNK_SVD", batch_size=BATCH, sgd_mode="sgd", use_bias=True,
           negative_interactions_quota=0.0)
    for symmetric in (False, True):
        sl = SLIM_BPR_MI355X_Epoch(urm, symmetric=symmetric, sgd_mode="adagrad", learning_rate=1e-4, topK=TOPK, random_seed=7)
        sl.epochIteration_Cython(1)
        sl.epochIteration_Cython(2)
        st = sl.stats()
        sec = st["call_ms"] * 1e-3
        t0 = time.perf_counter()
        sl.get_S_slabs(TOPK)
        out["slim_bpr_%s" % ("symmetric" if symmetric else "dense")] = {
            "samples_per_s": st["n_units"] / sec, "algorithmic_GBps": st["algorithmic_bytes"] / sec / 1e9,
            "launches_per_epoch": st["n_launches"] / 2, "seconds_per_epoch": sec / 2, "get_S_topk_s": time.perf_counter() - t0}
        sl.close()
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
    out["mf_scoring_1000_users_cutoff20"] = {"users_per_s": 1000 / wall, "device_ms": st["call_ms"], "gemm_ms": st["kernel_ms"],
                                             "gemm_f32_TFLOPs": st["algorithmic_flops"] / (st["kernel_ms"] * 1e-3) / 1e12,
                                             "frac_of_f32_mfma_peak_157TF": st["algorithmic_flops"] / (st["kernel_ms"] * 1e-3) / 1e12 / 157.3,
                                             "host_numpy_users_per_s": 1000 / host_wall}
    sc.close()
    k = 200
    conf = urm.copy()
    conf.data = (1.0 + 1.0 * conf.data).astype(np.float32)
    V0 = k ** -0.5 * np.random.default_rng(0).random((urm.shape[1], k))
    ia = IALS_MI355X_Epoch(conf, k, 1e-3, V0)
    ia.run_epochs(1)
    st = ia.stats()
    sec = st["call_ms"] * 1e-3
    out["ials_k200"] = {"seconds_per_epoch": sec, "row_solves_per_s": st["n_units"] / sec,
                        "algorithmic_fp64_TFLOPs": st["algorithmic_flops"] / sec / 1e12,
                        "frac_of_fp64_vector_peak_78.6TF": st["algorithmic_flops"] / sec / 1e12 / 78.6,
                        "row_kernel_ms": st["kernel_ms"]}
    ia.close()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    dist = None
    torch = None
    if world > 1:
        # torch first: its bundled HIP runtime (same SONAME) is then the one libmi355rec.so binds to
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")     # "gloo" + BENCH_SHARE_GPU=1: dry run of the N>1 path on one GPU
        if os.environ.get("BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import (Compute_Similarity_MI355X, MatrixFactorization_MI355X_Epoch, _native)
    from recsys2019_deeplearning_evaluation_amd.sharding import similarity_column_ranges, gather_slabs
    _native.load()
    if _native.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    _native.set_device(local_rank)

    def barrier():
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    urm = load_urm(args.workload)
    n_users, n_items = urm.shape
    per_epoch = (n_users // BATCH + 1) * BATCH

    # ------------------------------------------------------------------ BPR-MF epochs (headline value)
    mf = MatrixFactorization_MI355X_Epoch(urm, n_factors=K_FACTORS, algorithm_name="MF_BPR", batch_size=BATCH,
                                          learning_rate=1e-3, sgd_mode="sgd", init_std_dev=0.1, random_seed=42 + rank)
    if args.warmup > 0:
        mf.epochIteration_Cython(args.warmup)
    barrier()
    t0 = time.perf_counter()
    mf.epochIteration_Cython(args.steps)            # blocking: returns after the stream has drained; pure hipGraph replay
    barrier()
    elapsed = time.perf_counter() - t0
    st = mf.stats()
    elapsed = max_over_ranks(elapsed)
    total_samples = args.steps * per_epoch * world
    value = total_samples / elapsed
    # per-launch duration of the dominant kernel: a SEPARATE, untimed call whose mini-batch launches carry their own HIP
    # start/stop events on the handle's stream (plain launches instead of graph replay; not part of `value`)
    n_batches = per_epoch // BATCH
    mf.set_profiling(5 * n_batches)
    mf.epochIteration_Cython(5)
    pst = mf.stats()
    mf.set_profiling(0)
    avg_launch_s = (pst["kernel_ms"] / max(1, pst["n_timed"])) * 1e-3
    bytes_per_launch = st["algorithmic_bytes"] / max(1, st["n_launches"])
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "mf_batch_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic("mf_batch_kernel"),
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_launch_s * 1e6,
                "timed_launches": pst["n_timed"],
                "whole_epoch_achieved_GBps": st["algorithmic_bytes"] / (st["call_ms"] * 1e-3) / 1e9,
                "whole_epoch_frac": st["algorithmic_bytes"] / (st["call_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    mf.close()

    # ------------------------------------------------------------------ ItemKNN cosine build (second half of the metric)
    extra = {"bpr_loss_per_sample": st["loss"] / max(1, st["n_units"]), "bpr_stream_ms": st["call_ms"]}
    costs = None
    if not args.no_sim:
        # constructor = H2D of the URM + all of the set-up on the device (CSC view, profile stream, norms, costs): timed
        # because `ItemKNNCFRecommender.fit` pays it, like the reference's __init__
        sim = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
        sim.close()
        t_c = time.perf_counter()
        sim = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, normalize=True, similarity="cosine")
        sim.synchronize()
        extra["itemknn_create_s"] = time.perf_counter() - t_c
        costs = sim.column_costs()
        ranges = similarity_column_ranges(sim, world)
        s, e = ranges[rank]
        best = None
        for rep in range(3):
            barrier()
            t1 = time.perf_counter()
            if world == 1:
                idx, val, _ = sim.compute_slabs()
            else:
                widest = max(b - a for a, b in ranges)
                d_idx = torch.empty((widest, TOPK), dtype=torch.int32, device="cuda")
                d_val = torch.empty((widest, TOPK), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                sim.compute_slabs_device(s if s > 0 else None, e if e < n_items else None, d_idx.data_ptr(), d_val.data_ptr())
                sim.synchronize()
                f_idx, f_val = gather_slabs(d_idx, d_val, ranges, rank, TOPK, dist)
                idx, val = f_idx.cpu().numpy(), f_val.cpu().numpy()
            barrier()
            dt = max_over_ranks(time.perf_counter() - t1)
            best = dt if best is None else min(best, dt)
        sst = sim.stats()
        sim_gbps = sst["algorithmic_bytes"] / (sst["kernel_ms"] * 1e-3) / 1e9
        # what actually bounds the column kernel: LDS atomic adds, one per co-occurrence pair (ds_add_u32: 21.6 lane-adds
        # per CU and ns measured on MI355X with random cells, scripts/micro/lds_atomics.hip; 256 CUs), and the bytes its
        # own layout streams (uint16 ids, no values for all-ones data)
        pairs = float(np.asarray(costs[s:e], dtype=np.float64).sum())
        pair_rate = pairs / (sst["kernel_ms"] * 1e-3)
        extra.update({"itemknn_pairs_this_rank": pairs, "itemknn_pairs_per_s": pair_rate,
                      "itemknn_frac_of_lds_atomic_peak": pair_rate / (21.6e9 * 256),
                      "itemknn_stream_GBps_this_rank": 2.0 * pairs / (sst["kernel_ms"] * 1e-3) / 1e9})
        extra["itemknn_fit_s"] = extra["itemknn_create_s"] + best
        extra.update({"itemknn_cosine_build_s": best, "itemknn_topK": TOPK,
                      "itemknn_kernel_ms_this_rank": sst["kernel_ms"], "itemknn_columns_this_rank": int(e - s),
                      "itemknn_algorithmic_GBps_this_rank": sim_gbps, "itemknn_algorithmic_over_hbm_peak": sim_gbps / HBM_PEAK_GBPS,
                      "itemknn_nnz_out": int((idx >= 0).sum())})
        sim.close()

    out = {"metric": "BPR-MF SGD samples/sec (k=128, batch 1000) + ItemKNN cosine build sec on ML-20M-shaped URM",
           "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BPR-MF epoch (139 mini-batches x 1000 on-device samples), k=128, sgd, on synthetic %s URM "
                                  "%dx%d nnz=%d; replicas (one independent model per GPU)" % (args.workload, n_users, n_items, urm.nnz),
                      "batch_size": BATCH, "n_factors": K_FACTORS, "parallelism": "replicas x%d" % world},
           "roofline": roofline, "extra": extra}

    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out["extra"]["other_paths"] = other_paths(urm)
        except Exception as exc:                       # the headline line must survive a failure in the side measurements
            out["extra"]["other_paths_error"] = repr(exc)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline_bpr(urm, args.cpu_seconds)
        base["host_cpu_count"] = os.cpu_count()
        out["cpu_baseline"] = base
        out["extra"]["speedup_vs_cpu_baseline"] = value / base["value"]
        if costs is not None:
            sb = cpu_baseline_sim(urm, costs, args.cpu_seconds)
            out["extra"]["itemknn_cpu_baseline"] = sb
            out["extra"]["itemknn_speedup_vs_cpu_baseline"] = sb["value"] / out["extra"]["itemknn_cosine_build_s"]
            out["extra"]["itemknn_fit_speedup_vs_cpu_baseline"] = sb["value"] / out["extra"]["itemknn_fit_s"]
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
