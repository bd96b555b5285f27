"""Multi-GPU sharding of the hot paths that shard (one process per GPU; RCCL through torch.distributed or directly through ctypes).

Similarity build: every output column depends only on the read-only URM, so each rank holds the whole URM and builds a share
of the columns -- by default an INTERLEAVED part (the columns in cost order dealt to the ranks in serpentine order: every rank
gets n_cols / G columns and 1 / G of the cost, `interleaved_parts`), or a contiguous cost-balanced range along the seam the
reference already has and never uses, `compute_similarity(start_col, end_col)` (Compute_Similarity_Cython.pyx:411,447-451;
`balanced_column_ranges`).  The ranks exchange their (n_local x topK) neighbour / value slabs, which the kernel fills directly,
with ONE all-gather into a buffer allocated once (`ShardedSimilarityBuild`); there is no collective on the data path itself.
IALS: rows of a half-step are split into cost-balanced ranges, solved shards are all-gathered (`sharded_ials_epoch`).
BPR-MF: exact multi-GPU mini-batches (`sharded_bpr_epoch`, one all-gather of the updated rows per mini-batch).
On CPU (tests) the same host code runs over gloo with stand-in builders.
"""
import numpy as np


def balanced_column_ranges(cost, n_parts):
    """Cut [0, n) into n_parts contiguous ranges whose summed cost is as even as a prefix-sum cut allows.

    cost[c] = sum over the users of column c of their profile length (mi355rec_sim_column_costs): item
    popularity is heavily skewed, so equal COUNTS of columns would be badly unbalanced.  Every range is
    non-empty when n >= n_parts.  Returns a list of (start, end)."""
    cost = np.asarray(cost, dtype=np.float64) + 1.0          # +1: empty columns still cost a queue slot
    n = len(cost)
    n_parts = max(1, min(int(n_parts), n))
    prefix = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [0]
    for p in range(1, n_parts):
        target = prefix[-1] * p / n_parts
        cut = int(np.searchsorted(prefix, target, side="left"))
        cut = max(cut, bounds[-1] + 1)                       # keep every range non-empty ...
        cut = min(cut, n - (n_parts - p))                    # ... including the ones still to come
        bounds.append(cut)
    bounds.append(n)
    return [(bounds[i], bounds[i + 1]) for i in range(n_parts)]


# Per-column work that does not depend on the column's popularity -- clearing, normalising and ranking an accumulator
# of n_cols cells -- expressed in co-occurrence pairs: measured on MI355X at ML-20M shape (19 us per column against
# 8.3e6 pairs per workgroup-millisecond), i.e. about 6 pairs per accumulator cell.
FIXED_PAIRS_PER_CELL = 6


def similarity_column_ranges(similarity_object, world):
    """Contiguous column ranges of equal estimated build time: pairs to accumulate + the fixed per-column part."""
    cost = np.asarray(similarity_object.column_costs(), dtype=np.float64)
    return balanced_column_ranges(cost + FIXED_PAIRS_PER_CELL * similarity_object.n_columns, world)


def gather_slabs(local_idx, local_val, ranges, rank, topK, dist, device=None):
    """All-gather the per-rank (n_local, topK) slabs into full (n_cols, topK) arrays on every rank.

    local_idx / local_val: torch tensors (int32 / float32) holding this rank's range (at least n_local rows).
    Slabs are padded to the widest range so that one fixed-size all-gather per array suffices.  With the nccl
    (= RCCL) backend the exchange happens between device buffers over xGMI; with gloo (CPU tests, or the
    single-GPU dry run of bench.py) the slabs are staged through host memory."""
    import torch
    world = len(ranges)
    widest = max(e - s for s, e in ranges)
    on_host = dist.get_backend() == "gloo"
    device = torch.device("cpu") if on_host else (device if device is not None else local_idx.device)
    pad_idx = torch.full((widest, topK), -1, dtype=torch.int32, device=device)
    pad_val = torch.zeros((widest, topK), dtype=torch.float32, device=device)
    n_local = ranges[rank][1] - ranges[rank][0]
    pad_idx[:n_local] = local_idx[:n_local].to(device)
    pad_val[:n_local] = local_val[:n_local].to(device)
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    all_val = [torch.empty_like(pad_val) for _ in range(world)]
    dist.all_gather(all_idx, pad_idx)
    dist.all_gather(all_val, pad_val)
    full_idx = torch.cat([all_idx[r][:ranges[r][1] - ranges[r][0]] for r in range(world)], dim=0)
    full_val = torch.cat([all_val[r][:ranges[r][1] - ranges[r][0]] for r in range(world)], dim=0)
    return full_idx, full_val


def interleaved_parts(cost, n_parts):
    """The partition of mi355rec_sim_compute_part_device, restated on the host: columns sorted by descending cost (stable),
    dealt to the parts in serpentine order.  Returns one int32 array of column ids per part (equal lengths +-1, equal cost)."""
    order = np.argsort(-np.asarray(cost, dtype=np.int64), kind="stable")
    pos = np.arange(len(order))
    group, within = pos // n_parts, pos % n_parts
    owner = np.where(group % 2 == 1, n_parts - 1 - within, within)
    return [order[owner == p].astype(np.int32) for p in range(n_parts)]


def chunk_bounds(n_rows, chunks):
    """Row ranges [(r0, r1), ...] of `chunks` near-equal pieces of n_rows rows (the last ones may be empty for tiny inputs)."""
    chunks = max(1, int(chunks))
    step = -(-n_rows // chunks) if n_rows else 0
    return [(min(c * step, n_rows), min((c + 1) * step, n_rows)) for c in range(chunks)]


def cost_sized_pieces(row_costs, chunks, min_rows=512):
    """Row ranges of a part's output rows for the piece-wise exchange, sized by COST instead of by count, and the order to build them in.

    row_costs[q] = build cost of output row q (the rows of an interleaved part are in descending cost order).  Measured at ML-20M shape,
    8 parts (bench.py emulated_8_way, round 6): the 512 most expensive columns of a part take 0.23 ms on their own -- their heaviest
    columns bound them -- against 0.30 ms for the other 2 831 together, so equal COUNTS, heaviest first, made the first of four pieces as
    long as the whole part in one go (0.52 of 0.55 ms) and every later exchange stick out behind a short kernel.  Only the exchange of
    the piece built LAST is exposed, and an exchange takes time in proportion to its rows: the piece built last is therefore the head --
    the fewest rows the device can still be filled with, max(min_rows, n / (2 chunks)), two workgroups per CU -- and the rows behind it
    are cut into chunks - 1 pieces of equal cost (each at least min_rows rows) that are built FIRST, cheapest rows first, their many-row
    exchanges behind the next, more expensive piece's kernel.  Returns the ranges in ROW order (piece 0 = the head); build and exchange
    them in `reversed` order (`piece_order`).  Every rank computes the same bounds from the same costs.  Falls back to equal counts for
    parts too small to cut."""
    row_costs = np.asarray(row_costs, dtype=np.float64)
    n, chunks = len(row_costs), max(1, int(chunks))
    if chunks == 1 or n < (chunks + 1) * min_rows:
        return chunk_bounds(n, chunks)
    head = max(int(min_rows), n // (2 * chunks))
    prefix = np.concatenate([[0.0], np.cumsum(row_costs[head:] + 1.0)])
    bounds = [0, head]
    for p in range(1, chunks - 1):
        # (an equal share of what is LEFT: a piece that the minimum made larger than its share does not starve the last ones)
        done = bounds[-1] - head
        cut = head + int(np.searchsorted(prefix, prefix[done] + (prefix[-1] - prefix[done]) / (chunks - p), side="left"))
        cut = max(cut, bounds[-1] + min_rows)
        cut = min(cut, n - (chunks - 1 - p) * min_rows)
        bounds.append(cut)
    bounds.append(n)
    return [(bounds[i], bounds[i + 1]) for i in range(chunks)]


def piece_order(rows):
    """The order pieces are built and exchanged in: last rows (the cheapest columns of an interleaved part) first, the head last."""
    return list(range(len(rows) - 1, -1, -1))


# What the piece-wise exchange is decided on: the column kernel's rate on a part of a multi-GPU build (pair-adds per second; ML-20M shape,
# 8 parts: 0.98e9 pairs in 0.46-0.55 ms, Netflix shape 6.8e9 in 2.2 ms) and the per-link rate the exchange is modelled with.
PART_PAIRS_PER_SECOND = 2.5e12
LINK_BYTES_PER_SECOND = 50e9


def default_chunks(part_pairs, slab_bytes, world, exchange):
    """Pieces per part: 4 where the exchange is long next to the kernel that could hide it, else 1.

    Building a part in pieces costs kernel time (ML-20M shape, 8 parts: 0.58 ms in four pieces against 0.47 ms in one go; Netflix shape,
    equal counts: 2.6 against 2.16 ms), so it only pays where the exchange it hides is longer than that: a ring all-gather moves
    (world - 1) slabs over one link (0.33 ms against 0.47 ms of kernel at ML-20M shape: pieces; 0.24 against 2.2 ms at Netflix shape:
    one go), a gather to one rank a single slab per link (0.09 ms: one go at both shapes)."""
    if world <= 1:
        return 1
    steps = (world - 1) if exchange == "allgather" else 1
    exchange_s = steps * float(slab_bytes) / LINK_BYTES_PER_SECOND
    kernel_s = float(part_pairs) / PART_PAIRS_PER_SECOND
    return 4 if exchange_s > 0.3 * kernel_s else 1


def packed_words(n_cells):
    """4-byte words of n_cells packed (float32 value, 16-bit id) cells: mi355rec_sim_pack_slab_device's layout."""
    return int(n_cells) + (int(n_cells) + 1) // 2


def unpack_cells(words, n_cells):
    """(idx int32, val float32) of one packed block (host side of mi355rec_sim_unpack_slab_device)."""
    val = words[:n_cells].view(np.float32)
    ids = words[n_cells:packed_words(n_cells)].view(np.uint16)[:n_cells].astype(np.int32)
    ids[ids == 0xFFFF] = -1
    return ids, val


def pack_cells(idx, val):
    """Host restatement of mi355rec_sim_pack_slab_device (CPU-side tests and stand-ins): int32 words."""
    idx = np.ascontiguousarray(idx, np.int32).reshape(-1)
    n = idx.size
    ids = np.full(2 * ((n + 1) // 2), 0xFFFF, np.uint16)
    ids[:n] = (idx & 0xFFFF).astype(np.uint16)
    return np.concatenate([np.ascontiguousarray(val, np.float32).reshape(-1).view(np.int32), ids.view(np.int32)])


def assemble_gathered_pieces(host_words, world, rows, topK, columns, n_columns, packed=False):
    """The piece-major gathered buffer of a sharded similarity build (piece c: `[world][2][rows_c][topK]` words -- or, packed,
    `[world][packed_words(rows_c * topK)]` -- pieces one after the other) as (idx, val) arrays for all columns; `columns[r]` are rank
    r's columns in the order of its output rows."""
    idx = np.empty((n_columns, topK), np.int32)
    val = np.empty((n_columns, topK), np.int32)
    at = 0
    for r0, r1 in rows:
        cells = (r1 - r0) * topK
        words = packed_words(cells) if packed else 2 * cells
        for r, cols in enumerate(columns):
            have = max(0, min(r1, len(cols)) - r0)
            if have:
                block = host_words[at + r * words:at + (r + 1) * words]
                if packed:
                    p_idx, p_val = unpack_cells(block, cells)
                    idx[cols[r0:r0 + have]] = p_idx.reshape(r1 - r0, topK)[:have]
                    val[cols[r0:r0 + have]] = p_val.view(np.int32).reshape(r1 - r0, topK)[:have]
                else:
                    piece = block.reshape(2, r1 - r0, topK)
                    idx[cols[r0:r0 + have]] = piece[0, :have]
                    val[cols[r0:r0 + have]] = piece[1, :have]
        at += world * words
    return idx, val.view(np.float32)


class ChunkedAllGather:
    """All-gather of a send slab in pieces, so that a finished piece travels while the next one is computed.

    send: piece c = `words[c]` 4-byte words at word offset sum(words[:c]);  receive buffer: piece c of ALL ranks, rank-major, at word
    offset world * sum(words[:c]) -- every piece is one ordinary all-gather into a contiguous block.  start(c) returns at once where
    the transport can (RCCL through ctypes: the collective is enqueued on the null stream, which the library's non-blocking streams
    do not wait for; torch.distributed "nccl": async_op on RCCL's own stream); gloo stages through host memory and blocks (CPU-side
    tests).  The caller has synchronised the stream that produced piece c before start(c).

    root = r: a GATHER to rank r instead (the step the reference's single process does when it assembles W_sparse,
    Compute_Similarity_Cython.pyx:593-607 -- only the process that builds the model needs every column): the other ranks send their
    piece to r over their own link (on the xGMI mesh every peer is one hop away, so the G - 1 transfers run side by side instead of
    making G - 1 steps round a ring); only rank r's receive buffer is filled (recv may be None elsewhere)."""

    def __init__(self, send, recv, words, world, dist=None, comm=None, root=None, rank=0, as_tensor=None):
        self.send, self.recv, self.words, self.world, self.dist, self.comm = send, recv, list(words), world, dist, comm
        self.root, self.rank = root, rank
        # (as_tensor(address, n_words): the CPU-side tests hand in host buffers; the default views device memory of libmi355rec.so)
        tensor = as_tensor or (lambda address, n_words: device_tensor(address, (n_words,), "<i4"))
        self.offsets = [int(o) for o in np.concatenate([[0], np.cumsum(self.words)])]
        self.pending = []
        receives = root is None or rank == root
        if dist is not None:
            self.on_host = dist.get_backend() == "gloo"
            self.t_send = [tensor(send.address(self.offsets[c]), w) if w else None for c, w in enumerate(self.words)]
            self.t_recv = [tensor(recv.address(world * self.offsets[c]), world * w) if w and receives else None for c, w in enumerate(self.words)]

    def start(self, c):
        w = self.words[c]
        if not w or self.world == 1:
            return
        if self.comm is not None:
            if self.root is None:
                self.comm.all_gather_words_async(self.send.address(self.offsets[c]), self.recv.address(self.world * self.offsets[c]), w)
            else:
                self.comm.gather_words_async(self.send.address(self.offsets[c]),
                                             self.recv.address(self.world * self.offsets[c]) if self.rank == self.root else 0, w, self.root)
            return
        import torch
        if self.root is not None:
            mine = self.t_send[c].cpu() if self.on_host else self.t_send[c]
            parts = None
            if self.rank == self.root:
                whole = torch.empty(self.world * w, dtype=torch.int32) if self.on_host else self.t_recv[c]
                parts = list(whole.view(self.world, w).unbind(0))
            work = self.dist.gather(mine, parts, dst=self.root, async_op=not self.on_host)
            if self.on_host:
                if self.rank == self.root:
                    self.t_recv[c].copy_(whole)
            else:
                self.pending.append(work)
        elif self.on_host:
            mine = self.t_send[c].cpu()
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine)
            self.t_recv[c].copy_(torch.cat(parts))
        else:
            self.pending.append(self.dist.all_gather_into_tensor(self.t_recv[c], self.t_send[c], async_op=True))

    def finish(self):
        if self.world == 1:
            return
        if self.comm is not None:
            self.comm.synchronize()
            return
        import torch
        for work in self.pending:
            work.wait()
        self.pending = []
        if torch.cuda.is_available():
            torch.cuda.synchronize()


class ShardedSimilarityBuild:
    """The column-sharded build as a reusable object: partition, the rank's slab and the gathered result are allocated ONCE.

    partition="interleaved" (default): columns in cost order dealt to the ranks in serpentine order -- every rank gets
    n_columns / world columns AND 1 / world of the cost, so the exchanged slabs carry no padding (the contiguous cost-balanced
    ranges of partition="ranges", the reference's own start_col/end_col seam, are up to 2.6 x wider than the average at
    Netflix shape because unpopular columns are cheap).

    The rank's `widest` output rows are built in `chunks` pieces (default: 4 where the modelled exchange is long next to the part's
    kernel, else 1 -- `default_chunks`): the exchange of a finished piece
    overlaps the kernel of the next one, so only the exchange of the piece built LAST is exposed.  With the interleaved partition the
    pieces are sized by cost and built cheapest rows first (`cost_sized_pieces`: a small head of the most expensive columns, built
    last, behind equal-cost pieces of the other rows), with contiguous ranges by count.

    What travels (`pack`, default: whenever n_columns <= 65 535): 6-byte cells -- piece c of the send slab is the float32 values of
    its rows followed by their 16-bit neighbour ids (`mi355rec_sim_pack_slab_device`, one small kernel behind the piece's column
    kernel) -- instead of the `[2][rows_c][topK]` 4-byte words the column kernel writes (which travel as they are when pack=False:
    its two output pointers are then the two halves of the piece).  The gathered buffer holds, piece after piece, `[world][piece]`.

    exchange="allgather" (default): every rank ends with every column (a recommender object per rank can score its users).
    exchange="gather": only rank `root` (0) does -- the step the reference's one process performs when it assembles W_sparse
    (Compute_Similarity_Cython.pyx:593-607); the others' `download()` returns None.  On the xGMI mesh the G - 1 sends run side by
    side on their own links, where a ring all-gather makes G - 1 steps.

    `build()` returns when the result is resident on the receiving rank's (ranks') device -- the same definition at world == 1 (no
    exchange) -- and `download()` copies it to the host, which only a caller that needs NumPy arrays pays.
    Buffers belong to libmi355rec.so (raw device allocations); the transport is either
      `comm`: an rccl_direct.RcclCommunicator (RCCL through ctypes, no PyTorch in the process), or
      `dist`: torch.distributed ("nccl" = RCCL: device to device over xGMI; "gloo": staged through host memory, CPU-side tests)."""

    def __init__(self, similarity_object, dist=None, rank=0, world=1, comm=None, partition="interleaved", chunks=None, exchange="allgather",
                 pack=None, root=0):
        from ._native import DeviceArray
        self.sim, self.dist, self.comm, self.rank, self.world = similarity_object, dist, comm, rank, world
        assert world == 1 or (dist is None) != (comm is None), "exactly one transport: torch.distributed (dist) or RcclCommunicator (comm)"
        assert partition in ("interleaved", "ranges") and exchange in ("allgather", "gather")
        self.topK, self.n = similarity_object.TopK, similarity_object.n_columns
        self.partition = partition if world > 1 else "ranges"
        self.root = int(root) if (exchange == "gather" and world > 1) else None
        self.receives = self.root is None or rank == self.root
        self.packed = bool(world > 1 and self.n <= 65535 and (pack is None or pack))
        assert not (pack and self.n > 65535), "pack=True needs n_columns <= 65 535 (16-bit neighbour ids)"
        if chunks is None:
            cost_all = np.asarray(similarity_object.column_costs(), np.float64)
            chunks = default_chunks((cost_all.sum() + FIXED_PAIRS_PER_CELL * float(self.n) ** 2) / world,
                                    (6 if self.packed else 8) * -(-self.n // world) * self.topK, world, exchange)
        if self.partition == "interleaved":
            self.columns = [similarity_object.part_columns(r, world) for r in range(world)]
            self.ranges = None
            self.widest = max(len(c) for c in self.columns)
            # (the same bounds on every rank: part 0's rows, the most expensive column of every serpentine group among them)
            cost = np.asarray(similarity_object.column_costs(), np.float64) + FIXED_PAIRS_PER_CELL * self.n
            row_cost = np.zeros(self.widest)
            row_cost[:len(self.columns[0])] = cost[self.columns[0]]
            self.rows = cost_sized_pieces(row_cost, chunks)
        else:
            self.ranges = similarity_column_ranges(similarity_object, world) if world > 1 else [(0, self.n)]
            self.columns = [np.arange(s, e, dtype=np.int32) for s, e in self.ranges]
            self.widest = max(e - s for s, e in self.ranges)
            self.rows = chunk_bounds(self.widest, chunks)
        # (topK beyond the in-LDS selection -- more than 4096 neighbours, or accumulator tiles x topK beyond the merge buffer -- is built
        # dense + sorted; mi355rec_sim_compute_part_chunk_device walks the same rows of the part there too, so pieces work either way)
        self.work_words = 2 * self.widest * self.topK
        self.work_offsets = [2 * r0 * self.topK for r0, _ in self.rows]
        self.work = DeviceArray(self.work_words)
        # rows beyond this rank's columns are never written by the kernel but travel in the exchange: (-1, 0.0) like empty slots
        from . import _native as N
        fill = np.concatenate([np.concatenate([np.full((r1 - r0) * self.topK, -1, np.int32), np.zeros((r1 - r0) * self.topK, np.int32)])
                               for r0, r1 in self.rows])
        N.check(N.load().mi355rec_device_memcpy(self.work.ptr, N.ptr(fill), 4 * self.work_words, 1))
        piece_words = [packed_words((r1 - r0) * self.topK) if self.packed else 2 * (r1 - r0) * self.topK for r0, r1 in self.rows]
        self.slab_words = int(sum(piece_words))
        self.local = DeviceArray(self.slab_words) if self.packed else self.work
        if world == 1:
            self.gathered = self.local
        else:
            self.gathered = DeviceArray(world * self.slab_words) if self.receives else None
        self._host = np.empty(world * self.slab_words, np.int32) if self.receives else None
        self.gather = ChunkedAllGather(self.local, self.gathered, piece_words, world, dist, comm, root=self.root, rank=rank)

    def _build_piece(self, c):
        r0, r1 = self.rows[c]
        mine = len(self.columns[self.rank])
        count = max(0, min(r1, mine) - r0)
        at = self.work_offsets[c]
        d_idx, d_val = self.work.address(at), self.work.address(at + (r1 - r0) * self.topK)
        if count:
            if self.partition == "interleaved":
                if len(self.rows) == 1:
                    self.sim.compute_part_device(self.rank, self.world, d_idx, d_val)
                else:
                    self.sim.compute_part_chunk_device(self.rank, self.world, r0, count, d_idx, d_val)
            else:
                s, e = self.ranges[self.rank]
                a, b = s + r0, s + r0 + count
                self.sim.compute_slabs_device(a if a > 0 else None, b if b < self.n else None, d_idx, d_val)
        if self.packed and r1 > r0:
            # (also for a piece this rank has no column in: its empty rows travel like everybody's)
            self.sim.pack_slab_device(d_idx, d_val, (r1 - r0) * self.topK, self.local.address(self.gather.offsets[c]))
            return True
        return count > 0

    def build(self):
        """Kernel on this rank's columns, piece by piece, each piece's exchange behind the next piece's kernel; afterwards the
        gathered slabs are valid on the receiving device(s).  Blocking.  Returns the column kernel's milliseconds summed over the pieces
        (the similarity object's stats() only ever hold the LAST launch: a caller that divides this rank's work by a kernel time needs
        the sum)."""
        kernel_ms = 0.0
        mine = len(self.columns[self.rank])
        for c in piece_order(self.rows):
            if self._build_piece(c):
                self.sim.synchronize()
                if min(self.rows[c][1], mine) > self.rows[c][0]:
                    kernel_ms += float(self.sim.stats()["kernel_ms"])
            self.gather.start(c)
        self.gather.finish()
        self.kernel_ms = kernel_ms
        return kernel_ms

    def download(self):
        """(idx, val) NumPy arrays for ALL columns (None on a rank that the gather does not deliver to)."""
        if not self.receives:
            return None
        return assemble_gathered_pieces(self.gathered.to_host(self._host), self.world, self.rows, self.topK, self.columns, self.n, self.packed)

    def exchange_bytes_per_rank(self):
        return 0 if self.world == 1 else 4 * self.slab_words

    def close(self):
        self.work.close()
        if self.local is not self.work:
            self.local.close()
        if self.gathered is not None and self.gathered is not self.local:
            self.gathered.close()


def sharded_similarity_build(similarity_object, dist=None, rank=0, world=1, comm=None, partition="interleaved", **how):
    """Column-sharded build with a Compute_Similarity_MI355X object: returns (idx, val) numpy arrays for ALL
    columns on every rank (exchange="gather": on rank 0, None elsewhere).  With world == 1 this is the plain single-GPU build.
    `how`: chunks / exchange / pack / root of ShardedSimilarityBuild."""
    if world == 1:
        idx, val, _ = similarity_object.compute_slabs()
        return idx, val
    job = ShardedSimilarityBuild(similarity_object, dist, rank, world, comm, partition, **how)
    job.build()
    out = job.download()
    job.close()
    return out


# --------------------------------------------------------------------------------------------------
#                  BPR-MF: exact mini-batches, the tasks of every batch split over the ranks
# --------------------------------------------------------------------------------------------------

def sharded_bpr_epoch(epoch_object, dist=None, rank=0, world=1, comm=None):
    """One BPR-MF epoch with EXACT single-GPU semantics on `world` GPUs.

    Every rank holds an identical MatrixFactorization_MI355X_Epoch (same URM, same initial factors, same random_seed): the
    sample stream and the row-task schedule are then identical too.  Per mini-batch each rank runs 1 / world of the row tasks,
    the new row versions travel in ONE all-gather of fixed-size slabs (3 * batch_size / world rows of n_factors values per
    rank), and every rank copies the others' rows in: the replicas stay bit-identical and equal to a single-GPU run.  One
    exchange per mini-batch, so this only pays for large batch_size (the reference's search stops at 1024, where replicas --
    one model per GPU -- are the mode that scales)."""
    send, recv, nbytes, n_batches = epoch_object.shard_begin_epoch(rank, world)
    if world > 1 and comm is None:
        import torch
        # (an epoch object may wrap its slabs itself: the CPU stand-in of tests/test_sharding_gloo.py has no device memory)
        as_tensor = getattr(epoch_object, "shard_tensor", None) or (lambda address, n_words: device_tensor(address, (n_words,), "<i4"))
        t_send = as_tensor(send, nbytes // 4)
        t_recv = as_tensor(recv, world * nbytes // 4)
        on_host = dist.get_backend() == "gloo"
    for b in range(n_batches):
        epoch_object.shard_batch(b)
        if world > 1:
            if comm is not None:
                comm.all_gather_words(send, recv, nbytes // 4)
            elif on_host:
                mine = t_send.cpu()
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                t_recv.copy_(torch.cat(parts))
                if t_recv.is_cuda:
                    torch.cuda.synchronize()
            else:
                dist.all_gather_into_tensor(t_recv, t_send)
                torch.cuda.synchronize()
            epoch_object.shard_merge(b)
    epoch_object.shard_end_epoch()


def _run_range(similarity_object, s, e, n, d_idx, d_val):
    # compute_similarity's range rule treats start 0 / end n as "not given" (Compute_Similarity_Cython.pyx:447-451)
    similarity_object.compute_slabs_device(s if s > 0 else None, e if e < n else None, d_idx.data_ptr(), d_val.data_ptr())


# --------------------------------------------------------------------------------------------------
#                                   IALS: row-sharded half-steps
# --------------------------------------------------------------------------------------------------

def device_tensor(ptr, shape, typestr="<f8"):
    """torch view (no copy) of device memory owned by libmi355rec.so, through the CUDA array interface."""
    import torch

    class _Span:
        pass

    span = _Span()
    span.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr,
                                     "data": (int(ptr), False), "version": 2, "strides": None}
    return torch.as_tensor(span, device=torch.device("cuda", torch.cuda.current_device()))


def allgather_rows(T, ranges, rank, dist):
    """T: (n, k) device tensor of which this rank has just solved rows ranges[rank]; afterwards every rank holds
    every range.  One all-gather of shards padded to the widest range (the exchange step of an IALS half-epoch:
    111 MB of user factors + 21 MB of item factors per epoch at ML-20M shape, k = 200, float64)."""
    import torch
    world = len(ranges)
    widest = max(e - s for s, e in ranges)
    on_host = dist.get_backend() == "gloo"
    dev = torch.device("cpu") if on_host else T.device
    s, e = ranges[rank]
    pad = torch.zeros((widest, T.shape[1]), dtype=T.dtype, device=dev)
    pad[:e - s] = T[s:e].to(dev)
    shards = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(shards, pad)
    for r, (a, b) in enumerate(ranges):
        if r != rank:
            T[a:b] = shards[r][:b - a].to(T.device)
    torch.cuda.synchronize()


def sharded_ials_epoch(epoch_object, dist, rank, world, user_ranges, item_ranges):
    """One IALS epoch with the row solves of each half-step split over the ranks (IALS_MI355X_Epoch on every rank,
    identical state).  `user_ranges` / `item_ranges`: one (start, end) per rank, e.g. from ials_row_ranges() (cost-balanced
    cuts of the confidence matrix); they are required because the epoch object does not keep the host CSR."""
    if world == 1:
        epoch_object.run_epochs(1)
        return
    assert user_ranges is not None and item_ranges is not None and len(user_ranges) == world == len(item_ranges), \
        "sharded_ials_epoch: one user range and one item range per rank are required (see ials_row_ranges)"
    n_users, n_items, k = epoch_object.n_users, epoch_object.n_items, epoch_object.num_factors
    dU, dV = epoch_object.device_factor_pointers()
    U = device_tensor(dU, (n_users, k))
    V = device_tensor(dV, (n_items, k))
    epoch_object.user_half(*user_ranges[rank])
    epoch_object.synchronize()
    allgather_rows(U, user_ranges, rank, dist)
    epoch_object.item_half(*item_ranges[rank])
    epoch_object.synchronize()
    allgather_rows(V, item_ranges, rank, dist)


class ShardedIALSEpoch:
    """IALS epochs with the row solves of each half-step split over the ranks, as a reusable object (BASELINE config 5: "user-sharded
    across 8 x MI355X").  Every rank holds an identical IALS_MI355X_Epoch; per half-step a rank solves its cost-balanced range of
    rows (cost = L k^2 + k^3 / 3), stages them in a send slab, ONE all-gather of fixed-size slabs (the widest range, float64) and
    the other ranks' rows are copied into place, so that every rank holds the whole factor matrix again (the next half-step's
    Gramian is recomputed locally from it, IALSRecommender.py:141,156).  Buffers are raw device allocations of libmi355rec.so,
    made once; the transport is an rccl_direct.RcclCommunicator (`comm`, no PyTorch in the process) or torch.distributed (`dist`:
    "nccl" = RCCL device to device over xGMI, "gloo" staged through the host for the CPU-side tests)."""

    def __init__(self, epoch_object, confidence_csr, dist=None, rank=0, world=1, comm=None, chunks=None):
        from . import _native as N
        self._N = N
        self.epoch, self.dist, self.comm, self.rank, self.world = epoch_object, dist, comm, rank, world
        assert world == 1 or (dist is None) != (comm is None), "exactly one transport: torch.distributed (dist) or RcclCommunicator (comm)"
        self.k = epoch_object.num_factors
        self.user_ranges, self.item_ranges = ials_row_ranges(confidence_csr, world, self.k)
        self.dU, self.dV = epoch_object.device_factor_pointers()
        if world > 1:
            # a half-step's rows are solved in pieces: the all-gather of a finished piece travels while the next one is solved.  `chunks`:
            # one number, or (user half, item half).  Default (4, 1): measured on one GPU (bench.py extra.ials.emulated_8_way, ML-20M shape,
            # k = 200) the four pieces of a user range cost 12.3 ms against 11.2 ms in one go -- less than the 3 ms of exchange they hide
            # on one ring --, the four pieces of an item range 10.3 ms against 7.6 ms (the longest-first order inside a range is what keeps
            # its tail short): the item half goes in one piece.
            cu, ci = (chunks if isinstance(chunks, (tuple, list)) else (chunks, chunks)) if chunks is not None else (4, 1)
            self.widest = max(max(e - s for s, e in self.user_ranges), max(e - s for s, e in self.item_ranges))
            self.slab_words = 2 * self.widest * self.k                       # float64 = two 4-byte words
            self.send = N.DeviceArray(self.slab_words)
            self.recv = N.DeviceArray(world * self.slab_words)
            self.rows_of = {"users": chunk_bounds(self.widest, cu), "items": chunk_bounds(self.widest, ci)}
            self.gather_of = {half: ChunkedAllGather(self.send, self.recv, [2 * (r1 - r0) * self.k for r0, r1 in rows], world, dist, comm)
                              for half, rows in self.rows_of.items()}

    def _copy(self, dst, src, nbytes):
        import ctypes as C
        if nbytes:
            self._N.check(self._N.load().mi355rec_device_memcpy(C.c_void_p(dst), C.c_void_p(src), int(nbytes), 2))

    def _half(self, solve, base, ranges, half):
        """The rank's rows piece by piece: solve, stage in the send slab, start the piece's all-gather, go on with the next piece;
        at the end the other ranks' rows are copied into place."""
        row = 8 * self.k
        rows, gather = self.rows_of[half], self.gather_of[half]
        s, e = ranges[self.rank]
        for c, (r0, r1) in enumerate(rows):
            a, b = min(s + r0, e), min(s + r1, e)
            if b > a:
                solve(a, b)
                self.epoch.synchronize()
                self._copy(self.send.address(gather.offsets[c]), base + a * row, (b - a) * row)
                self._N.check(self._N.load().mi355rec_device_synchronize())
            gather.start(c)
        gather.finish()
        for r, (ra, rb) in enumerate(ranges):
            if r == self.rank:
                continue
            for c, (r0, r1) in enumerate(rows):
                a, b = min(ra + r0, rb), min(ra + r1, rb)
                if b > a:
                    self._copy(base + a * row, self.recv.address(self.world * gather.offsets[c] + r * gather.words[c]), (b - a) * row)
        # the next half-step runs on the epoch object's own (non-blocking) stream: do not rely on hipMemcpy being synchronous for
        # device-to-device copies
        self._N.check(self._N.load().mi355rec_device_synchronize())

    def run_epoch(self):
        """One epoch; returns when every rank's device holds the complete, updated U and V.  Blocking."""
        if self.world == 1:
            self.epoch.run_epochs(1)
            return
        self._half(self.epoch.user_half, self.dU, self.user_ranges, "users")
        self._half(self.epoch.item_half, self.dV, self.item_ranges, "items")

    def exchange_bytes_per_rank_per_epoch(self):
        return 0 if self.world == 1 else 2 * 4 * self.slab_words

    def close(self):
        if self.world > 1:
            self.send.close()
            self.recv.close()


# What a row's solve (Cholesky + substitutions + the per-row set-up) costs next to the Gramian of its profile, in profile entries per
# factor: the flop counts alone say k / 3 entries, but the solve stage runs at a fifth of the Gramian stage's rate.
# Fitted to the eight measured item ranges of the ML-20M shape at k = 200 (profiles/r6_ials_ranges.txt): t = 1.53e-6 ms per entry
# + 2.74e-4 ms per row = 180 entries per row; with k / 3 the tail range (9 136 short rows) took 5.83 ms against 4.27 ms for the head.
IALS_ROW_ENTRIES_PER_FACTOR = 0.9


def ials_row_ranges(confidence_csr, world, num_factors):
    """Cost-balanced user and item ranges: cost(row) = (L + 0.9 k) * k^2 -- the Gramian of the row's L profile entries plus its solve,
    priced in entries (IALS_ROW_ENTRIES_PER_FACTOR)."""
    import scipy.sparse as sps
    k = float(num_factors)
    fixed = IALS_ROW_ENTRIES_PER_FACTOR * k
    lu = np.diff(confidence_csr.indptr).astype(np.float64)
    li = np.diff(sps.csc_matrix(confidence_csr).indptr).astype(np.float64)
    return (balanced_column_ranges((lu + fixed) * k * k, world), balanced_column_ranges((li + fixed) * k * k, world))
