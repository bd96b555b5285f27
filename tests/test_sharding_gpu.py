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
from recsys2019_deeplearning_evaluation_amd.sharding import (sharded_similarity_build, sharded_ials_epoch, ials_row_ranges,
                                                           balanced_column_ranges)
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
torch.cuda.set_device(0); _native.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()

X = named_urm("ml1m", "binary", scale=0.2)
sim = Compute_Similarity_MI355X(X, topK=30, shrink=2, similarity="jaccard")
full_idx, full_val, _ = sim.compute_slabs()
for partition in ("interleaved", "ranges"):
    idx, val = sharded_similarity_build(sim, dist, rank, world, partition=partition)
    assert np.array_equal(idx, full_idx) and np.array_equal(val, full_val), "sharded similarity (%%s) differs on rank %%d" %% (partition, rank)
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
