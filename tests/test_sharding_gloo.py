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
# the same in PIECES (ShardedSimilarityBuild builds a part in chunks and gathers a finished one behind the next one's kernel): piece c of
# a slab is [2][rows_c][topK], the gathered buffer piece-major; reassembled by the function download() uses
from recsys2019_deeplearning_evaluation_amd.sharding import assemble_gathered_pieces, chunk_bounds
for chunks in (1, 3, 4, widest + 5):
    rows = chunk_bounds(widest, chunks)
    assert rows[0][0] == 0 and max(r1 for _, r1 in rows) == widest and all(a <= b for a, b in rows)
    gathered = []
    for r0, r1 in rows:
        piece = mine[:, r0:r1, :].contiguous().reshape(-1)
        out = torch.empty(world * piece.numel(), dtype=torch.int32)
        if piece.numel():
            dist.all_gather_into_tensor(out, piece)
        gathered.append(out)
    got_idx, got_val = assemble_gathered_pieces(torch.cat(gathered).numpy(), world, rows, 12, parts, X.shape[1])
    assert np.array_equal(got_idx, ref_idx) and np.array_equal(got_val, ref_val), "pieces (%%d) rank %%d mismatch" %% (chunks, rank)
# the exchange as ShardedSimilarityBuild runs it: cost-sized pieces, 6-byte packed cells (values + 16-bit ids), all-gather or gather
# to a root, through ChunkedAllGather itself (host buffers stand in for the device slabs)
import ctypes as C
from recsys2019_deeplearning_evaluation_amd.sharding import ChunkedAllGather, cost_sized_pieces, pack_cells, packed_words, piece_order
class HostWords:
    def __init__(self, n): self.a = np.zeros(n, np.int32)
    def address(self, word_offset=0): return self.a.ctypes.data + 4 * int(word_offset)
as_tensor = lambda address, n: torch.from_numpy(np.ctypeslib.as_array((C.c_int32 * n).from_address(address)))
row_cost = np.zeros(widest); row_cost[:len(parts[0])] = cost[parts[0]]
for rows in (cost_sized_pieces(row_cost, 3, min_rows=10), chunk_bounds(widest, 4)):
    for packed in (True, False):
        words = [packed_words((r1 - r0) * 12) if packed else 2 * (r1 - r0) * 12 for r0, r1 in rows]
        for root in (None, 0, 1):
            send, recv = HostWords(sum(words)), HostWords(world * sum(words))
            recv.a[:] = 12345
            gather = ChunkedAllGather(send, recv, words, world, dist, None, root=root, rank=rank, as_tensor=as_tensor)
            for c in piece_order(rows):
                r0, r1 = rows[c]
                piece = mine[:, r0:r1, :].numpy()
                send.a[gather.offsets[c]:gather.offsets[c + 1]] = pack_cells(piece[0], piece[1].view(np.float32)) if packed else piece.reshape(-1)
                gather.start(c)
            gather.finish()
            if root is None or root == rank:
                got_idx, got_val = assemble_gathered_pieces(recv.a, world, rows, 12, parts, X.shape[1], packed)
                assert np.array_equal(got_idx, ref_idx) and np.array_equal(got_val, ref_val), "exchange (packed %%s, root %%s) rank %%d mismatch" %% (packed, root, rank)
            else:
                assert (recv.a == 12345).all(), "a gather to rank %%d wrote on rank %%d" %% (root, rank)
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


BPR_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from recsys2019_deeplearning_evaluation_amd.sharding import sharded_bpr_epoch
from recsys2019_deeplearning_evaluation_amd.synthetic import synthetic_urm


class StandInEpoch:
    """CPU stand-in with the shard_* surface of MatrixFactorization_MI355X_Epoch (there is no GPU here): BPR mini-batches in NumPy,
    one task per (row, mini-batch) in sorted row order, the tasks of a mini-batch dealt to the ranks in contiguous slot ranges."""

    def __init__(self, X, k, B, lr, seed):
        self.X, self.k, self.B, self.lr = X.tocsr(), k, B, lr
        rng = np.random.default_rng(seed)
        self.nu, self.ni = X.shape
        self.W = rng.normal(0, 0.1, (self.nu + self.ni, k)).astype(np.float32)      # users, then items
        self.rng = np.random.default_rng(seed + 1)

    def _draw(self):
        n = (self.nu // self.B + 1) * self.B
        u = self.rng.integers(0, self.nu, n)
        i = np.array([self.X.indices[self.rng.integers(self.X.indptr[a], self.X.indptr[a + 1])] for a in u])
        j = self.rng.integers(0, self.ni, n)
        return u, i + self.nu, j + self.nu

    def shard_begin_epoch(self, rank, world):
        self.rank, self.world = rank, world
        self.u, self.i, self.j = self._draw()
        self.n_batches = len(self.u) // self.B
        self.spr = -(-3 * self.B // world)
        self.send = np.zeros((self.spr, self.k), np.float32)
        self.recv = np.zeros((world * self.spr, self.k), np.float32)
        return 0, 1, self.send.nbytes, self.n_batches

    def shard_tensor(self, address, n_words):
        return torch.from_numpy((self.recv if address else self.send).reshape(-1).view(np.int32))

    def _tasks(self, b):
        sl = slice(b * self.B, (b + 1) * self.B)
        return np.unique(np.concatenate([self.u[sl], self.i[sl], self.j[sl]])), sl

    def shard_batch(self, b):
        rows, sl = self._tasks(b)
        old = self.W.copy()
        u, i, j = self.u[sl], self.i[sl], self.j[sl]
        x = np.einsum("ij,ij->i", old[u], old[i] - old[j])
        g = (1.0 / (1.0 + np.exp(x))).astype(np.float32)[:, None]
        lo, hi = self.rank * self.spr, min((self.rank + 1) * self.spr, len(rows))
        for slot in range(lo, hi):
            r = rows[slot]
            acc = (g[u == r] * (old[i[u == r]] - old[j[u == r]])).sum(0) + (g[i == r] * old[u[i == r]]).sum(0) - (g[j == r] * old[u[j == r]]).sum(0)
            self.W[r] = old[r] + self.lr * acc / self.B
            self.send[slot - lo] = self.W[r]

    def shard_merge(self, b):
        rows, _ = self._tasks(b)
        lo, hi = self.rank * self.spr, (self.rank + 1) * self.spr
        for slot, r in enumerate(rows):
            if not lo <= slot < hi:
                self.W[r] = self.recv[slot]

    def shard_end_epoch(self):
        pass


dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
X = synthetic_urm(300, 120, 5000, 3, 60, seed=9, values="binary")
single = StandInEpoch(X, 16, 64, 0.05, 3)
split = StandInEpoch(X, 16, 64, 0.05, 3)
for _ in range(2):
    sharded_bpr_epoch(single, None, 0, 1)
    sharded_bpr_epoch(split, dist, rank, world)
assert np.array_equal(single.W, split.W), "rank %%d: the split mini-batches differ from the single-process ones" %% rank
dist.barrier()
if rank == 0:
    print("SHARDED_BPR_OK")
dist.destroy_process_group()
'''


def test_exact_bpr_mode_world_size_2_gloo(tmp_path):
    """The host side of SURVEY 8(e)'s exact multi-GPU BPR mode (sharding.sharded_bpr_epoch: per mini-batch kernel share -> one
    all-gather of fixed slabs -> merge) over gloo with a NumPy stand-in for the epoch object; the device side is
    tests/test_sharding_gpu.py."""
    script = tmp_path / "bpr_worker.py"
    script.write_text(BPR_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDED_BPR_OK" in outs[0]


def test_cost_sized_pieces_and_packed_cells():
    """Host pieces of the sharded build's exchange: piece bounds sized by cost (equal-cost pieces, a small last one, every row once, the same
    on every rank) and the 6-byte cell format (values, then 16-bit ids with 0xFFFF for the empty slot's -1)."""
    from recsys2019_deeplearning_evaluation_amd.sharding import cost_sized_pieces, chunk_bounds, pack_cells, unpack_cells, packed_words, piece_order
    rng = np.random.default_rng(5)
    cost = 1.6e5 + 9e5 * np.exp(-np.arange(3343) / 700.0)                  # a part of the ML-20M shape: 3343 rows in descending cost order
    rows = cost_sized_pieces(cost, 4)
    assert rows[0][0] == 0 and rows[-1][1] == len(cost) and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    sizes = [b - a for a, b in rows]
    assert sizes[0] == max(512, len(cost) // 8) and min(sizes) >= 512   # the head: the fewest rows, built (and exchanged) last
    piece_cost = [cost[a:b].sum() for a, b in rows]
    assert max(piece_cost[1:]) <= 1.25 * min(piece_cost[1:])
    assert sizes[1] < sizes[2] < sizes[3]                               # (expensive rows first: fewer of them per piece)
    assert piece_order(rows) == [3, 2, 1, 0]
    assert cost_sized_pieces(cost[:1500], 4) == chunk_bounds(1500, 4)   # too small to cut by cost
    assert cost_sized_pieces(cost, 1) == [(0, len(cost))]
    for n in (0, 1, 2, 7, 100):
        idx = rng.integers(-1, 65535, n).astype(np.int32)
        val = rng.random(n).astype(np.float32)
        words = pack_cells(idx, val)
        assert words.dtype == np.int32 and len(words) == packed_words(n)
        got_idx, got_val = unpack_cells(words, n)
        assert np.array_equal(got_idx, idx) and np.array_equal(got_val, val)


def test_pieces_only_where_the_exchange_is_long_next_to_the_kernel():
    """sharding.default_chunks at the two BASELINE shapes, 8 ranks, topK 100 (pair counts of bench.py's synthetic URMs): a ring all-gather at
    ML-20M shape is built in pieces, everything else in one go."""
    from recsys2019_deeplearning_evaluation_amd.sharding import default_chunks, FIXED_PAIRS_PER_CELL
    ml20m = ((7.82e9 + FIXED_PAIRS_PER_CELL * 26744.0 ** 2) / 8, 6 * 3343 * 100)
    netflix = ((54.09e9 + FIXED_PAIRS_PER_CELL * 17770.0 ** 2) / 8, 6 * 2222 * 100)
    assert default_chunks(*ml20m, 8, "allgather") == 4 and default_chunks(*ml20m, 8, "gather") == 1
    assert default_chunks(*netflix, 8, "allgather") == 1 and default_chunks(*netflix, 8, "gather") == 1
    assert default_chunks(*ml20m, 1, "allgather") == 1
