"""The N>1 code paths with real device kernels, exercised on ONE GPU: two processes share device 0 and exchange
through gloo (RCCL refuses two ranks on one device; the collective itself is covered by the driver's multi-GPU run).
Checks: column-sharded similarity build == single-process build; row-sharded IALS epoch == single-process epoch."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist            # torch first: one HIP runtime in the process
import numpy as np
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X, IALS_MI355X_Epoch, _native
from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch
from recsys2019_deeplearning_evaluation_amd.sharding import (sharded_similarity_build, sharded_ials_epoch, ials_row_ranges,
                                                           balanced_column_ranges, sharded_bpr_epoch, ShardedIALSEpoch)
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
torch.cuda.set_device(0); _native.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()

X = named_urm("ml1m", "binary", scale=0.2)
sim = Compute_Similarity_MI355X(X, topK=30, shrink=2, similarity="jaccard")
full_idx, full_val, _ = sim.compute_slabs()
for partition in ("interleaved", "ranges"):
    # default: 6-byte packed cells in cost-sized pieces, all-gather; then the 8-byte words, and both as a gather to rank 0 / rank 1
    for how in (dict(), dict(pack=False), dict(exchange="gather"), dict(exchange="gather", pack=False, chunks=3, root=1), dict(chunks=1)):
        out = sharded_similarity_build(sim, dist, rank, world, partition=partition, **how)
        if how.get("exchange") == "gather" and rank != how.get("root", 0):
            assert out is None
            continue
        idx, val = out
        assert np.array_equal(idx, full_idx) and np.array_equal(val, full_val), "sharded similarity (%%s, %%s) differs on rank %%d" %% (partition, how, rank)
ranges = balanced_column_ranges(sim.column_costs(), world)
assert ranges[0][1] < X.shape[1] // 2, "cost balancing must give the popular (low-index) columns the shorter range"

C = X.copy(); C.data = (1.0 + 3.0 * C.data).astype(np.float32)
k = 40
V0 = k ** -0.5 * np.random.default_rng(0).random((X.shape[1], k))
single = IALS_MI355X_Epoch(C, k, 1e-2, V0); single.run_epochs(2)
Us, Vs = single.get_factors()
shard = IALS_MI355X_Epoch(C, k, 1e-2, V0)
ur, ir = ials_row_ranges(C, world, k)
for _ in range(2):
    sharded_ials_epoch(shard, dist, rank, world, ur, ir)
Ud, Vd = shard.get_factors()
assert np.abs(Ud - Us).max() <= 1e-12 * np.abs(Us).max() and np.abs(Vd - Vs).max() <= 1e-12 * np.abs(Vs).max(), "sharded IALS differs"
# the same as a reusable object with its buffers allocated once (what bench.py --gpus N times)
obj_epoch = IALS_MI355X_Epoch(C, k, 1e-2, V0)
job = ShardedIALSEpoch(obj_epoch, C, dist, rank, world)
for _ in range(2):
    job.run_epoch()
Uo, Vo = obj_epoch.get_factors()
assert np.abs(Uo - Us).max() <= 1e-12 * np.abs(Us).max() and np.abs(Vo - Vs).max() <= 1e-12 * np.abs(Vs).max(), "ShardedIALSEpoch differs"
assert job.exchange_bytes_per_rank_per_epoch() > 0
job.close()
# exact multi-GPU BPR mini-batches: the tasks of every batch split over the two ranks == the single-process epochs, bit for bit
kw = dict(n_factors=32, algorithm_name="MF_BPR", batch_size=512, learning_rate=0.05, sgd_mode="sgd", user_reg=0.01, positive_reg=0.02,
          negative_reg=0.03, random_seed=11)
one = MatrixFactorization_MI355X_Epoch(X, **kw); one.epochIteration_Cython(2)
two = MatrixFactorization_MI355X_Epoch(X, **kw)
for _ in range(2):
    sharded_bpr_epoch(two, dist, rank, world)
assert np.array_equal(one.get_USER_factors(), two.get_USER_factors()) and np.array_equal(one.get_ITEM_factors(), two.get_ITEM_factors()), "sharded BPR differs"
dist.barrier()
if rank == 0:
    print("SHARDED_GPU_OK")
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_share_one_gpu(gpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDED_GPU_OK" in outs[0]


def test_packed_exchange_cells_on_the_device(gpu):
    """mi355rec_sim_pack_slab_device / unpack: the 6-byte cells of the sharded build's exchange equal the host restatement
    (sharding.pack_cells), for even and odd cell counts, and unpack gives the slabs back with -1 in the empty slots."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X, _native as N
    from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
    from recsys2019_deeplearning_evaluation_amd.sharding import pack_cells, packed_words, ShardedSimilarityBuild
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    X = named_urm("ml1m", "binary", scale=0.3)
    topK = 7
    sim = Compute_Similarity_MI355X(X, topK=topK, shrink=0)
    idx, val, _ = sim.compute_slabs()
    built_idx, built_val = idx.copy(), val.copy()
    idx[::5, -2:] = -1; val[::5, -2:] = 0.0                    # (columns with fewer than topK neighbours: empty slots travel too)
    for rows in (len(idx), 331, 1, 0):
        cells = rows * topK
        slab = np.concatenate([idx[:rows].reshape(-1), val[:rows].reshape(-1).view(np.int32)])
        work, packed, back = DeviceArray(max(2 * cells, 1)), DeviceArray(max(packed_words(cells), 1)), DeviceArray(max(2 * cells, 1))
        if cells:
            N.check(N.load().mi355rec_device_memcpy(work.ptr, N.ptr(slab), 4 * len(slab), 1))
        sim.pack_slab_device(work.address(), work.address(cells), cells, packed.address())
        sim.unpack_slab_device(packed.address(), cells, back.address(), back.address(cells))
        sim.synchronize()
        assert np.array_equal(packed.to_host()[:packed_words(cells)], pack_cells(idx[:rows], val[:rows]))
        assert np.array_equal(back.to_host()[:2 * cells], slab)
        work.close(); packed.close(); back.close()
    # world == 1: the sharded object is the plain build (nothing packed, nothing exchanged)
    job = ShardedSimilarityBuild(sim)
    job.build()
    got_idx, got_val = job.download()
    assert np.array_equal(got_idx, built_idx) and np.array_equal(got_val, built_val) and job.exchange_bytes_per_rank() == 0 and not job.packed
    job.close(); sim.close()


def test_parts_at_ml20m_shape_with_heavy_columns_routed_to_the_32_bit_launch(gpu, monkeypatch):
    """At the ML-20M shape the heavy columns of an 8-way part are 60 % of its pair-adds: the packed-counts call hands them to the 32-bit launch
    behind it (sim.hip, run_columns_lds: MI355REC_SIM_PACKED_DEMOTE).  Parts 0 and 7 built with that rule (the default), with it forced
    off, and without the packed kernel at all are identical cell for cell -- whole and in the pieces the sharded build cuts them into --
    and the schedule shows that the rule fired (fewer split columns in the packed call); the whole-shape build does not use it."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
    from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
    from recsys2019_deeplearning_evaluation_amd.sharding import cost_sized_pieces, FIXED_PAIRS_PER_CELL
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    X = named_urm("ml20m", "binary")
    n, G, topK = X.shape[1], 8, 100
    w = -(-n // G)
    buf = DeviceArray(2 * w * topK)
    results, schedules = {}, {}
    for tag, env in (("rule", {}), ("rule off", {"MI355REC_SIM_PACKED_DEMOTE": "0"}), ("32-bit only", {"MI355REC_SIM_PACKED": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = Compute_Similarity_MI355X(X, topK=topK, shrink=0, normalize=True, similarity="cosine")
        out = []
        for part in (0, 7):
            sim.compute_part_device(part, G, buf.address(), buf.address(w * topK))
            sim.synchronize()
            out.append(buf.to_host().copy())
            schedules[(tag, part)] = sim.schedule_info()
        cost = np.asarray(sim.column_costs(), np.float64)
        cols0 = sim.part_columns(0, G)
        row_cost = np.zeros(w); row_cost[:len(cols0)] = cost[cols0] + FIXED_PAIRS_PER_CELL * n
        for r0, r1 in cost_sized_pieces(row_cost, 4):
            cnt = min(r1, len(cols0)) - r0
            sim.compute_part_chunk_device(0, G, r0, cnt, buf.address(), buf.address(cnt * topK))
            sim.synchronize()
            out.append(buf.to_host()[:2 * cnt * topK].copy())
        results[tag] = out
        sim.close()
        for k in env:
            monkeypatch.delenv(k)
    for tag in ("rule off", "32-bit only"):
        assert len(results[tag]) == len(results["rule"])
        for a, b in zip(results["rule"], results[tag]):
            assert np.array_equal(a, b), "parts differ between the heavy-column rule and '%s'" % tag
    # (work items, split columns, parts): with the rule the packed call splits nothing heavy, the 32-bit launch's own limit applies
    assert schedules[("rule", 0)] != schedules[("rule off", 0)] and schedules[("rule", 0)][1] < schedules[("rule off", 0)][1]
    buf.close()


def test_interleaved_parts_on_one_gpu(gpu):
    """The 8-way interleaved partition, part after part on one device: equal counts, equal cost, and the parts put together are the
    single build bit for bit; the host restatement of the partition (sharding.interleaved_parts) names the same columns."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
    from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
    from recsys2019_deeplearning_evaluation_amd.sharding import interleaved_parts
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    X = named_urm("ml1m", "real", scale=0.3)
    topK = 25
    sim = Compute_Similarity_MI355X(X, topK=topK, shrink=3)
    full_idx, full_val, _ = sim.compute_slabs()
    n, G = X.shape[1], 8
    costs = sim.column_costs()
    want_parts = interleaved_parts(costs, G)
    widest = -(-n // G)
    buf = DeviceArray(2 * widest * topK)
    idx = np.full((n, topK), -7, np.int32); val = np.zeros((n, topK), np.float32)
    part_cost = []
    for r in range(G):
        cols = sim.part_columns(r, G)
        np.testing.assert_array_equal(cols, want_parts[r])
        assert len(cols) in (n // G, widest)
        sim.compute_part_device(r, G, buf.address(), buf.address(widest * topK))
        sim.synchronize()
        host = buf.to_host().reshape(2, widest, topK)
        idx[cols] = host[0, :len(cols)]
        val[cols] = host[1, :len(cols)].view(np.float32)
        part_cost.append(float(costs[cols].sum()))
    np.testing.assert_array_equal(idx, full_idx)
    np.testing.assert_array_equal(val, full_val)
    assert max(part_cost) <= 1.02 * min(part_cost) + float(costs.max())
    buf.close(); sim.close()


def test_interleaved_parts_with_topk_beyond_the_lds_selection(gpu, monkeypatch):
    """topK > 4096 goes through dense columns + a segmented sort, a block of columns at a time.  An interleaved part (the default
    partition of the sharded build) is walked in blocks too: 3 parts, blocks of 100 columns, against the single build."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
    from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
    from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
    n, topK, G = 4500, 4200, 3
    X = synthetic_urm(700, n, 30 * n // 10, 5, 500, seed=21, values="real", zipf_exponent=0.5)
    sim = Compute_Similarity_MI355X(X, topK=topK, shrink=1)
    full_idx, full_val, _ = sim.compute_slabs()
    monkeypatch.setenv("MI355REC_SIM_WIDE_CELLS", str(100 * n))
    widest = -(-n // G)
    buf = DeviceArray(2 * widest * topK)
    idx = np.full((n, topK), -7, np.int32); val = np.zeros((n, topK), np.float32)
    for r in range(G):
        cols = sim.part_columns(r, G)
        sim.compute_part_device(r, G, buf.address(), buf.address(widest * topK))
        sim.synchronize()
        host = buf.to_host().reshape(2, widest, topK)
        idx[cols] = host[0, :len(cols)]
        val[cols] = host[1, :len(cols)].view(np.float32)
    np.testing.assert_array_equal(idx, full_idx)
    np.testing.assert_array_equal(val, full_val)
    buf.close(); sim.close()


def test_part_pieces_with_topk_beyond_the_lds_selection(gpu, monkeypatch, n=4500, topK=4200, cells=100):
    """The sharded build computes a rank's part in 4 pieces by default (ShardedSimilarityBuild.rows); a handle whose top-K goes through
    dense columns + the segmented sort -- topK > 4096 candidates, or a wide catalogue whose accumulator tiles x topK exceed the merge
    buffer -- must serve those pieces too (round 4 raised MI355REC_E_UNSUPPORTED there):
    pieces of 3 interleaved parts, walked in blocks of a few columns, against the single build."""
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X
    from recsys2019_deeplearning_evaluation_amd._native import DeviceArray
    from recsys2019_deeplearning_evaluation_amd.sharding import chunk_bounds
    from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
    G = 3
    X = synthetic_urm(500, n, 6 * n, 3, 400, seed=23, values="binary", zipf_exponent=0.5)
    sim = Compute_Similarity_MI355X(X, topK=topK, shrink=1)
    full_idx, full_val, _ = sim.compute_slabs()
    monkeypatch.setenv("MI355REC_SIM_WIDE_CELLS", str(cells * n))
    widest = -(-n // G)
    pieces = chunk_bounds(widest, 4)
    idx = np.full((n, topK), -7, np.int32); val = np.zeros((n, topK), np.float32)
    for r in range(G):
        cols = sim.part_columns(r, G)
        for r0, r1 in pieces:
            count = max(0, min(r1, len(cols)) - r0)
            if count == 0:
                continue
            buf = DeviceArray(2 * (r1 - r0) * topK)
            sim.compute_part_chunk_device(r, G, r0, count, buf.address(), buf.address((r1 - r0) * topK))
            sim.synchronize()
            host = buf.to_host().reshape(2, r1 - r0, topK)
            idx[cols[r0:r0 + count]] = host[0, :count]
            val[cols[r0:r0 + count]] = host[1, :count].view(np.float32)
            buf.close()
    np.testing.assert_array_equal(idx, full_idx)
    np.testing.assert_array_equal(val, full_val)
    sim.close()


@pytest.mark.parametrize("world,batch_size,k", [(1, 1000, 128), (4, 1000, 128), (8, 4096, 64), (3, 37, 20), (2, 2000, 8)])
def test_exact_multi_gpu_bpr_emulated_on_one_gpu(gpu, world, batch_size, k):
    """SURVEY 8(e)'s exact mode: `world` identical replicas in ONE process stand for the ranks; every mini-batch each runs its share of
    the row tasks, the exchange slabs are copied between them device to device (what the all-gather does), everybody merges.
    All replicas must end bit-identical to the plain single-GPU epochs -- including lists split over a workgroup (large batches)
    and the any-k kernel (k = 20)."""
    import ctypes as C
    import numpy as np
    from recsys2019_deeplearning_evaluation_amd import MatrixFactorization_MI355X_Epoch, _native as N
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    X = named_urm("ml1m", "binary", scale=0.5)
    kw = dict(n_factors=k, algorithm_name="MF_BPR", batch_size=batch_size, learning_rate=0.05, sgd_mode="sgd", user_reg=0.01,
              positive_reg=0.02, negative_reg=0.03, random_seed=5)
    single = MatrixFactorization_MI355X_Epoch(X, **kw)
    single.epochIteration_Cython(2)
    ranks = [MatrixFactorization_MI355X_Epoch(X, **kw) for _ in range(world)]
    lib = N.load()
    for _ in range(2):
        slabs = [m.shard_begin_epoch(r, world) for r, m in enumerate(ranks)]
        n_batches, nbytes = slabs[0][3], slabs[0][2]
        for b in range(n_batches):
            for m in ranks:
                m.shard_batch(b)
            for dst in range(world):                 # the all-gather: rank src's slab lands at offset src in everybody's receive buffer
                for src in range(world):
                    N.check(lib.mi355rec_device_memcpy(C.c_void_p(slabs[dst][1] + src * nbytes), C.c_void_p(slabs[src][0]), nbytes, 2))
            for m in ranks:
                m.shard_merge(b)
        for m in ranks:
            m.shard_end_epoch()
    U, V = single.get_USER_factors(), single.get_ITEM_factors()
    for m in ranks:
        np.testing.assert_array_equal(m.get_USER_factors(), U)
        np.testing.assert_array_equal(m.get_ITEM_factors(), V)
    # a third epoch through the ordinary entry point continues from the same state
    single.epochIteration_Cython(1); ranks[-1].epochIteration_Cython(1)
    np.testing.assert_array_equal(ranks[-1].get_USER_factors(), single.get_USER_factors())
    with pytest.raises(NotImplementedError):
        MatrixFactorization_MI355X_Epoch(X, **dict(kw, sgd_mode="adam")).shard_begin_epoch(0, 2)
