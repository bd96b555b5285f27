"""The N>1 path of the column-sharded similarity build over torch.distributed (gloo, world_size 2, CPU).

Each rank builds its cost-balanced column range with a stand-in column builder (the CPU oracle -- there is no GPU
here), the ranks exchange the padded slabs with the same gather code the GPU path uses, and every rank must end
up with the single-process result."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from oracle import oracle as O
from recsys2019_deeplearning_evaluation_amd.sharding import balanced_column_ranges, gather_slabs
from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
X = synthetic_urm(400, 150, 9000, 3, 90, seed=7, values="real")
orc = O.OracleSimilarity(X, topK=12, shrink=2)
L = np.diff(X.indptr).astype(np.int64); C = X.tocsc()
cost = np.array([L[C.indices[C.indptr[c]:C.indptr[c + 1]]].sum() for c in range(X.shape[1])])
ranges = balanced_column_ranges(cost, world)
s, e = ranges[rank]
idx, val = orc.build_slabs(s, e)
full_idx, full_val = gather_slabs(torch.from_numpy(idx), torch.from_numpy(val), ranges, rank, 12, dist)
ref_idx, ref_val = orc.build_slabs(0, X.shape[1])
assert np.array_equal(full_idx.numpy(), ref_idx) and np.array_equal(full_val.numpy(), ref_val), "rank %%d mismatch" %% rank
# interleaved partition (equal counts, equal cost): unpadded equal-size slabs, ONE all_gather_into_tensor of [2][widest][topK] words
from recsys2019_deeplearning_evaluation_amd.sharding import interleaved_parts
parts = interleaved_parts(cost, world)
assert sorted(np.concatenate(parts).tolist()) == list(range(X.shape[1])) and max(map(len, parts)) - min(map(len, parts)) <= 1
widest = max(map(len, parts))
mine = torch.zeros((2, widest, 12), dtype=torch.int32)
mine[0] = -1
for q, c in enumerate(parts[rank]):
    mine[0, q] = torch.from_numpy(ref_idx[c]); mine[1, q] = torch.from_numpy(ref_val[c].view(np.int32))      # (stand-in for the column kernel)
everything = torch.empty(world * mine.numel(), dtype=torch.int32)
dist.all_gather_into_tensor(everything, mine.reshape(-1))
everything = everything.reshape(world, 2, widest, 12)
got_idx = np.empty_like(ref_idx); got_val = np.empty_like(ref_val)
for r, cols in enumerate(parts):
    got_idx[cols] = everything[r, 0, :len(cols)].numpy(); got_val[cols] = everything[r, 1, :len(cols)].numpy().view(np.float32)
assert np.array_equal(got_idx, ref_idx) and np.array_equal(got_val, ref_val), "interleaved rank %%d mismatch" %% rank
costs_per_part = [cost[p].sum() for p in parts]
assert max(costs_per_part) - min(costs_per_part) <= cost.max(), costs_per_part
dist.barrier()
if rank == 0:
    print("SHARDING_OK", ranges)
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDING_OK" in outs[0]
