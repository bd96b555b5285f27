// mf_schedule.cuh -- (row, mini-batch) incidences -> tasks: the radix-sort schedule of a whole stream and the in-LDS schedule
// (one workgroup per mini-batch; fused and pair tasks, wide lists), stream-end kernels.  Included by mf.hip after mf_core.cuh.
#pragma once

namespace mi355rec {
namespace {

// ---- schedule: (row, mini-batch) incidences -> tasks ----------------------------------------------------------------
struct SchedParams {
    long long n_samples;
    int per;                 // incidences per sample: 3 (BPR) / 2 (FunkSVD)
    int n_users, batch_size, batch_bits, tasks_per_batch;
    const int *su, *si, *sj;
    const float *sr;
    unsigned long long *keys;      // unsorted keys: entry << batch_bits | batch
    int *slots;                    // unsorted values: sample * per + role
    const unsigned long long *keys_sorted;
    const int *slots_sorted;
    int *head;                     // 1 where a new (row, batch) run starts
    const int *head_scan;          // inclusive scan of head
    int *task_at;                  // for run heads: index of the task header
    unsigned char *spar;           // per incidence: buffer of the row's version this sample reads
    unsigned char *par;            // per row: buffer of the current version (advanced by the last task of the row)
    int *batch_count;
    int *slot_flag;                // per original slot (sample * per + role): 1 where a task's first incidence sits
    const int *slot_rank;          // exclusive scan of slot_flag
    TaskHeader *tasks;
    int4 *recs;
};

__global__ __launch_bounds__(256) void mf_keys_kernel(const SchedParams s) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= s.n_samples) return;
    const unsigned long long batch = (unsigned long long)(t / s.batch_size);
    const long long q = t * s.per;
    s.keys[q] = ((unsigned long long)s.su[t] << s.batch_bits) | batch;
    s.slots[q] = (int)q;
    s.keys[q + 1] = ((unsigned long long)(s.n_users + s.si[t]) << s.batch_bits) | batch;
    s.slots[q + 1] = (int)q + 1;
    if (s.per == 3) {
        s.keys[q + 2] = ((unsigned long long)(s.n_users + s.sj[t]) << s.batch_bits) | batch;
        s.slots[q + 2] = (int)q + 2;
    }
}

__global__ __launch_bounds__(256) void mf_heads_kernel(const SchedParams s) {
    const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (q >= s.n_samples * s.per) return;
    const int head = q == 0 || s.keys_sorted[q] != s.keys_sorted[q - 1];
    s.head[q] = head;
    if (head) s.slot_flag[s.slots_sorted[q]] = 1;      // (the sort is stable: the run's first incidence in stream order)
}

// One thread per run head: version parity of the row at this batch, a place in the batch's task array, the header.
__global__ __launch_bounds__(256) void mf_tasks_kernel(const SchedParams s) {
    const long long n = s.n_samples * s.per;
    const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (q >= n || !s.head[q]) return;
    const unsigned long long key = s.keys_sorted[q];
    const int entry = (int)(key >> s.batch_bits);
    const int batch = (int)(key & ((1ull << s.batch_bits) - 1));
    long long end = q + 1;
    while (end < n && !s.head[end]) ++end;
    // rank of this batch among the batches of the stream that touch the row = number of run heads since the row's first
    const unsigned long long first_key = (unsigned long long)entry << s.batch_bits;
    long long lo = 0, hi = q;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (s.keys_sorted[mid] < first_key) lo = mid + 1; else hi = mid;
    }
    const int rank = s.head_scan[q] - s.head_scan[lo];
    const int parity = (s.par[entry] + rank) & 1;
    // the task's place among the mini-batch's headers = the rank of its FIRST incidence (the sort is stable: the smallest
    // sample * per + role of the run) among the first incidences of the batch's tasks: headers are packed at the front of
    // the batch's slots in stream order -- the same layout on every replica of the stream (the exact multi-GPU mode splits a
    // mini-batch's slots over the ranks), with no atomic counter (three mini-batches of 65 536 samples used to serialise
    // 590 k atomics on three addresses: 86 M samples/s against 230 M)
    const long long first_slot = (long long)batch * s.tasks_per_batch;
    // -- or, with a per-batch counter (s.batch_count), simply the next free one: run heads arrive roughly in key order, i.e. the
    // batch's headers end up sorted by row (users first), which the mini-batch kernel of FunkSVD's 20 001 small batches likes
    // better (118 vs 87 M samples/s at ML-20M shape: neighbouring wavefronts gather neighbouring rows, and the workgroups that
    // carry global-bias terms are the leading ones)
    int at = (int)first_slot;
    if (s.batch_count) at += atomicAdd(&s.batch_count[batch], 1);
    else at += s.slot_rank[s.slots_sorted[q]] - s.slot_rank[first_slot];
    s.task_at[q] = at;
    TaskHeader h;
    h.entry = entry;
    h.meta = (int)(end - q) | (parity << 31);
    h.start = (int)q;
    h.pad = 0;
    s.tasks[at].entry = h.entry;
    s.tasks[at].meta = h.meta;
    s.tasks[at].start = h.start;
    s.tasks[at].pad = 0;
    for (long long r = q; r < end; ++r) s.spar[s.slots_sorted[r]] = (unsigned char)parity;
}

// One thread per incidence (sorted order): the sample record with the parities of all its rows.
__global__ __launch_bounds__(256) void mf_recs_kernel(const SchedParams s) {
    const long long n = s.n_samples * s.per;
    const long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int slot = s.slots_sorted[q];
    const int sample = slot / s.per, role = slot - sample * s.per;
    const long long base = (long long)sample * s.per;
    int4 rec;
    rec.x = s.su[sample];
    rec.y = s.si[sample];
    rec.z = s.per == 3 ? s.sj[sample] : __float_as_int(s.sr[sample]);
    rec.w = role | (s.spar[base] << 2) | (s.spar[base + 1] << 3) | (s.per == 3 ? s.spar[base + 2] << 4 : 0);
    s.recs[q] = rec;
    if (s.head[q]) {
        s.tasks[s.task_at[q]].rec0 = rec;
        // last task of this row in the stream: the next stream starts from the other buffer
        const int entry = (int)(s.keys_sorted[q] >> s.batch_bits);
        long long end = q + 1;
        while (end < n && !s.head[end]) ++end;
        if (end == n || (int)(s.keys_sorted[end] >> s.batch_bits) != entry) s.par[entry] = s.spar[slot] ^ 1;
    }
}

// ---- fast schedule: one workgroup per mini-batch sorts its incidences in LDS -----------------------------------------------
// The general path above radix-sorts the whole stream (device-wide sort: 120 us for the 417 k incidences
// of a BPR epoch at ML-20M shape, plus 270 us for the task kernel's dependent searches) -- two thirds of the time of the 139
// mini-batches it prepares.  When a mini-batch fits LDS and the stream has at most 256 mini-batches the same tasks come out of
// three short kernels: (1) per mini-batch, a stable radix sort of (row, slot) keys in LDS, run lengths, task slots (lists longer
// than two rounds of one wavefront get the 4 wavefronts of a workgroup: 4 aligned headers), and one bit per (row, mini-batch) in a global bitmap;
// (2) per incidence, the version parity of each of the sample's rows = parity at stream start + number of earlier
// mini-batches with the row's bit set; (3) per row, the parity after the stream, bitmap cleared for the next one.
constexpr int META_WIDE = 1 << 30;        // header.meta: bits 0-27 list length, 28-29 quarter, 30 wide, 31 buffer of the own row
constexpr int SCHED_THREADS = 1024;
constexpr int SLOT_ABSORBED = 0x3fffffff;   // qtask[] of an incidence whose row is updated by its sample's user task (no header of its own)
constexpr int FAST_MAX_BATCHES = 256, FAST_MAX_SLOTS = 8192;

struct FastSchedParams {
    long long n_samples;
    int per, n_users, n_entries, batch_size, tasks_per_batch, slot_bits, np, words, group;
    int fill_all;                 // 1: every header slot past the last one in use is zeroed (the one-model kernel launches a wavefront per slot);
                                  // 0: only up to the next multiple of 4 (the group kernel walks the slots in use, a workgroup of four wavefronts at a time)
    int mid_bytes;                // LDS bytes between the keys and the once-touched flags (run starts + header slots, or the sort's scratch)
    int entry_bits;               // bits of a row id (users, then items) + 1: the padding keys' all-ones field sorts last
    int fuse;                     // BPR: a sample whose user row is touched once in the batch takes over its other once-touched rows
    const int *su, *si, *sj;
    const float *sr;
    unsigned *touched;            // [n_entries][words]: bit b of row x = mini-batch b of this stream touches x
    unsigned char *par;           // [n_entries]: buffer of every row's current version at stream start
    int *sorted_slot;             // [n_batches][tasks_per_batch]: per batch-local incidence (sample * per + role) its position in (row, id) order | also << 16
    int *qtask;                   // per sorted position: batch-local slot of its task's (first) header | wide << 30
    int *used;                    // [n_batches]: header slots in use
    TaskHeader *tasks;
    int4 *recs;
    int4 *slot_recs;              // [n_batches][tasks_per_batch][3]: see MfParams
};

// The keys (row << slot_bits | incidence, in incidence order) only have to be grouped by row with the incidences of a row in
// stream order: a STABLE radix sort on the row bits does it in (row bits + 7) / 8 passes (rocPRIM block sort) where the bitonic
// network of round 2 needed 78 barrier-separated stages for 4096 keys (74 us per mini-batch: as much as the mini-batch itself
// once 32 models share a launch).
template <int IPT> struct SchedSort {
    using type = rocprim::block_radix_sort<unsigned, SCHED_THREADS, IPT>;
    static __device__ __forceinline__ void run(unsigned *K, void *storage, int begin_bit, int end_bit) {
        unsigned keys[IPT];
#pragma unroll
        for (int i = 0; i < IPT; ++i) keys[i] = K[threadIdx.x * IPT + i];
        type().sort(keys, *reinterpret_cast<typename type::storage_type *>(storage), (unsigned)begin_bit, (unsigned)end_bit);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IPT; ++i) K[threadIdx.x * IPT + i] = keys[i];
    }
};
constexpr size_t sched_sort_storage_bytes(int np) {
    return np <= SCHED_THREADS ? sizeof(SchedSort<1>::type::storage_type)
           : (np <= 2 * SCHED_THREADS ? sizeof(SchedSort<2>::type::storage_type)
              : (np <= 4 * SCHED_THREADS ? sizeof(SchedSort<4>::type::storage_type) : sizeof(SchedSort<8>::type::storage_type)));
}

__device__ __forceinline__ void mf_sched_sort_body(const FastSchedParams &s, const int b) {
    extern __shared__ __attribute__((aligned(16))) unsigned sched_lds[];
    unsigned *K = sched_lds;                                   // [np] keys: row << slot_bits | incidence
    int *hpos = reinterpret_cast<int *>(sched_lds + s.np);     // [np + 1] first sorted position of every run
    int *tpos = hpos + s.np + 1;                               // [np] header slot of every task
    typedef rocprim::block_scan<int, SCHED_THREADS> Scan;
    __shared__ typename Scan::storage_type scan_tmp;
    const int tid = threadIdx.x, np = s.np, sb = s.slot_bits;
    const long long first = (long long)b * s.batch_size;
    const int n_in = (int)min((long long)s.batch_size, s.n_samples - first);
    const int m = n_in * s.per;
    for (int q = tid; q < np; q += SCHED_THREADS) {
        unsigned key = 0xFFFFFFFFu;
        if (q < m) {
            const int smp = q / s.per, role = q - smp * s.per;
            const long long t = first + smp;
            const int entry = role == 0 ? s.su[t] : s.n_users + (role == 1 ? s.si[t] : s.sj[t]);
            key = ((unsigned)entry << sb) | (unsigned)q;
        }
        K[q] = key;
    }
    __syncthreads();
    // (the sort's scratch lives where the run starts / header slots go afterwards)
    switch (np / SCHED_THREADS) {
        case 1: SchedSort<1>::run(K, hpos, sb, sb + s.entry_bits); break;
        case 2: SchedSort<2>::run(K, hpos, sb, sb + s.entry_bits); break;
        case 4: SchedSort<4>::run(K, hpos, sb, sb + s.entry_bits); break;
        default: SchedSort<8>::run(K, hpos, sb, sb + s.entry_bits); break;
    }
    __syncthreads();
    // run heads -> hpos[]
    const int C = np / SCHED_THREADS;
    int cnt = 0;
    for (int c = 0; c < C; ++c) {
        const int q = tid * C + c;
        cnt += q < m && (q == 0 || (K[q] >> sb) != (K[q - 1] >> sb));
    }
    int off = 0, total = 0;
    Scan().exclusive_scan(cnt, off, 0, total, scan_tmp);
    for (int c = 0; c < C; ++c) {
        const int q = tid * C + c;
        if (q < m && (q == 0 || (K[q] >> sb) != (K[q - 1] >> sb))) hpos[off++] = q;
    }
    if (tid == 0) hpos[total] = m;
    __syncthreads();
    // FUSED SAMPLE TASKS (BPR).  Most rows of a mini-batch are touched by exactly one sample (users nearly always, uniformly drawn
    // negative items mostly): as separate tasks each of them gathers the sample's three rows again -- nine row reads and three
    // wavefronts per sample.  A sample whose USER row is touched once keeps one task (the user's) which also applies the update of
    // the sample's item rows that are touched once (record bits 5 / 6: positive / negative item); those rows get no task of their
    // own.  And `group` such tasks that sit next to each other in the sorted order share one wavefront (PAIR tasks, header word 3 =
    // 1): its lane groups, which walk a list's samples `group` at a time, each take one of the samples and write that sample's
    // rows -- a single-sample task leaves all but one lane group of its wavefront idle otherwise.  The arithmetic per row is
    // unchanged.  single[q] = incidence q (sample * per + role) is alone in its run.
    unsigned char *single = reinterpret_cast<unsigned char *>(sched_lds) + sizeof(unsigned) * (size_t)np + s.mid_bytes;
    const unsigned qmask = (1u << sb) - 1u;
    if (s.fuse) {
        for (int q = tid; q < np; q += SCHED_THREADS) single[q] = 0;
        __syncthreads();
        for (int t = tid; t < total; t += SCHED_THREADS)
            if (hpos[t + 1] - hpos[t] == 1) single[K[hpos[t]] & qmask] = 1;
        __syncthreads();
    }
    auto absorbed = [&](int t) -> bool {          // a once-touched item row whose sample's user row is touched once, too
        if (!s.fuse || hpos[t + 1] - hpos[t] != 1) return false;
        const int inc = (int)(K[hpos[t]] & qmask), smp = inc / s.per;
        return inc != smp * s.per && single[smp * s.per];
    };
    auto lone_user = [&](int q) -> bool {         // sorted position q is the single-sample run of a user row
        if (q >= m) return false;
        const unsigned e = K[q] >> sb;
        return e < (unsigned)s.n_users && (q == 0 || (K[q - 1] >> sb) != e) && (q + 1 == m || (K[q + 1] >> sb) != e);
    };
    auto paired = [&](int q) -> bool {            // q lies in an aligned block of `group` positions that are all such runs
        if (!s.fuse || s.group < 2) return false;
        const int q0 = q - q % s.group;
        for (int e = 0; e < s.group; ++e)
            if (!lone_user(q0 + e)) return false;
        return true;
    };
    // header slots: wide tasks first (4 aligned slots each), then the others; both in row order
    const int CT = (total + SCHED_THREADS - 1) / SCHED_THREADS;
    const int t_lo = min(tid * CT, total), t_hi = min(t_lo + CT, total);
    int wcnt = 0, acnt = 0;
    const int wide_min = 2 * s.group;               // longer than two rounds of one wavefront: split over a workgroup
    for (int t = t_lo; t < t_hi; ++t) {
        const int start = hpos[t], len = hpos[t + 1] - start;
        wcnt += len > wide_min;
        acnt += absorbed(t) || (len == 1 && start % s.group != 0 && paired(start));      // runs without a header of their own
    }
    int woff = 0, n_wide = 0, aoff = 0, n_abs = 0;
    Scan().exclusive_scan(wcnt, woff, 0, n_wide, scan_tmp);
    __syncthreads();
    Scan().exclusive_scan(acnt, aoff, 0, n_abs, scan_tmp);
    TaskHeader *out = s.tasks + (size_t)b * s.tasks_per_batch;
    for (int t = t_lo; t < t_hi; ++t) {
        const int start = hpos[t], len = hpos[t + 1] - start;
        const bool wide = len > wide_min;
        const int entry = (int)(K[start] >> sb);
        atomicOr(&s.touched[(size_t)entry * s.words + (b >> 5)], 1u << (b & 31));      // (rows without a task advance a version, too)
        const bool pair = len == 1 && paired(start);
        if (absorbed(t) || (pair && start % s.group != 0)) {
            ++aoff;
            tpos[t] = SLOT_ABSORBED;
            continue;
        }
        const int slot = wide ? 4 * woff : 4 * n_wide + (t - woff - aoff);
        woff += wide;
        tpos[t] = slot | (wide ? META_WIDE : 0);
        for (int part = 0; part < (wide ? 4 : 1); ++part) {
            *reinterpret_cast<int4 *>(out + slot + part) =
                make_int4(entry, (pair ? s.group : len) | (wide ? META_WIDE | (part << 28) : 0), b * s.tasks_per_batch + start, pair ? 1 : 0);
            out[slot + part].rec0 = make_int4(0, 0, 0, 0);     // (a short wide list leaves its last quarters without a record)
        }
    }
    const int used = 4 * n_wide + (total - n_wide - n_abs);
    if (tid == 0) s.used[b] = used;
    const int fill_end = s.fill_all ? s.tasks_per_batch : min(s.tasks_per_batch, (used + 3) & ~3);
    for (int slot = used + tid; slot < fill_end; slot += SCHED_THREADS)
        *reinterpret_cast<int4 *>(out + slot) = make_int4(0, 0, 0, 0);          // no samples: the slot's wavefront idles
    __syncthreads();
    for (int q = tid; q < m; q += SCHED_THREADS) {
        int lo = 0, hi = total;                       // last run starting at or before q
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (hpos[mid] <= q) lo = mid; else hi = mid;
        }
        const int inc = (int)(K[q] & qmask);
        int also = 0;                                 // the item rows a once-touched user row's task takes over (bits 16-17 here, 5-6 of the record)
        if (s.fuse && hpos[lo + 1] - hpos[lo] == 1) {
            const int smp = inc / s.per;
            if (inc == smp * s.per) also = (single[inc + 1] ? 1 : 0) | (s.per == 3 && single[inc + 2] ? 2 : 0);
        }
        // (by INCIDENCE: the emit kernel runs one thread per sample, which looks its three incidences' sorted positions up; a scattered
        // 4-byte store into the mini-batch's 12 KB here, three coalesced loads there)
        s.sorted_slot[(size_t)b * s.tasks_per_batch + inc] = q | (also << 16);
        s.qtask[(size_t)b * s.tasks_per_batch + q] = tpos[lo];
    }
}
__global__ __launch_bounds__(SCHED_THREADS) void mf_sched_sort_kernel(const FastSchedParams s) { mf_sched_sort_body(s, blockIdx.x); }

__device__ __forceinline__ int version_parity(const FastSchedParams &s, int entry, int b) {
    const unsigned *w = s.touched + (size_t)entry * s.words;
    int cnt = 0;
    for (int k = 0; k < (b >> 5); ++k) cnt += __popc(w[k]);
    cnt += __popc(w[b >> 5] & ((1u << (b & 31)) - 1u));
    return (s.par[entry] + cnt) & 1;
}

// One thread per SAMPLE (until round 6: per incidence -- every sample's ids were then fetched by three threads at three unrelated
// sorted positions and the version parities of its three rows evaluated three times over: 1.08 GB fetched per 32-model epoch).  The
// sample's ids come in coalesced, each row's parity is evaluated once, and the record goes to the sample's three sorted positions.
__device__ __forceinline__ void mf_sched_emit_body(const FastSchedParams &s) {
    const int b = blockIdx.y, smp = blockIdx.x * 256 + threadIdx.x;
    const long long first = (long long)b * s.batch_size;
    const int n_in = (int)min((long long)s.batch_size, s.n_samples - first);
    if (smp >= n_in) return;
    const size_t base = (size_t)b * s.tasks_per_batch;
    const long long t = first + smp;
    const int u = s.su[t], i = s.si[t], j = s.per == 3 ? s.sj[t] : 0;
    const int third = s.per == 3 ? j : __float_as_int(s.sr[t]);
    int where[3];
#pragma unroll
    for (int role = 0; role < 3; ++role) where[role] = role < s.per ? s.sorted_slot[base + (size_t)smp * s.per + role] : 0;
    const int pu = version_parity(s, u, b), pi = version_parity(s, s.n_users + i, b);
    const int pj = s.per == 3 ? version_parity(s, s.n_users + j, b) : 0;
    const int par_bits = (pu << 2) | (pi << 3) | (pj << 4);
#pragma unroll
    for (int role = 0; role < 3; ++role) {
        if (role >= s.per) break;
        const int q = where[role] & 0xffff, also = where[role] >> 16;
        const size_t at = base + q;
        const int4 rec = make_int4(u, i, third, role | par_bits | (also << 5));
        s.recs[at] = rec;
        const int tp = s.qtask[at];
        if (tp == SLOT_ABSORBED) {
            // a sample of a PAIR task other than its first (pairs are aligned blocks of `group` sorted positions, the header belongs to the
            // first): its record also goes where the mini-batch kernel finds it without having seen the header -- by header slot
            const int q0 = q - q % s.group;
            if (s.fuse && q0 != q) {
                const int tp0 = s.qtask[base + q0];
                if (tp0 != SLOT_ABSORBED && !(tp0 & META_WIDE)) {
                    const TaskHeader *lead = s.tasks + base + tp0;
                    if (lead->pad == 1 && lead->start == (int)(base + q0)) s.slot_recs[(base + tp0) * 3 + (q - q0 - 1)] = rec;
                }
            }
            continue;
        }
        TaskHeader *hd = s.tasks + base + (tp & (META_WIDE - 1));
        const int off = (int)(at - (size_t)hd->start);
        const int own = role == 0 ? pu : (role == 1 ? pi : pj);
        int part = -1;
        if (tp & META_WIDE) {           // quarter k of a wide list starts at position k * group
            if (off % s.group == 0 && off / s.group < 4) part = off / s.group;      // (quarters past the end of the list stay empty)
        } else if (off == 0) {
            part = 0;
        }
        if (part >= 0) {
            hd[part].rec0 = rec;
            hd[part].meta |= own << 31;
        }
    }
}

__global__ __launch_bounds__(256) void mf_sched_emit_kernel(const FastSchedParams s) { mf_sched_emit_body(s); }

__device__ __forceinline__ void mf_sched_finish_body(const FastSchedParams &s) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= s.n_entries) return;
    unsigned *w = s.touched + (size_t)x * s.words;
    int cnt = 0;
    for (int k = 0; k < s.words; ++k) {
        const unsigned v = w[k];
        if (v) { cnt += __popc(v); w[k] = 0; }
    }
    if (cnt & 1) s.par[x] ^= 1;
}
__global__ __launch_bounds__(256) void mf_sched_finish_kernel(const FastSchedParams s) { mf_sched_finish_body(s); }

// The global-bias ring is indexed by the mini-batch's position in its STREAM (a kernel knows that from its arguments: the ring
// entry is requested together with everything else, not after the global index has arrived).  A stream of n mini-batches leaves
// the newest state and terms in entry (n - 1) % 3; mini-batch 0 of the next stream looks for them in entry 2 and adds its own
// terms to entry 0.
template <class T>
__device__ __forceinline__ void ring_to_stream_start(const MfParams<T> &p, const long long n_batches, const int slot) {
    if (n_batches <= 0) return;
    const int src = (int)((n_batches - 1) % 3);
    if (src != 2) {
        if (slot == 0) p.mu_state[2] = p.mu_state[src];
        p.mu_acc[2 * MU_SLOTS + slot] = p.mu_acc[src * MU_SLOTS + slot];
    }
    p.mu_acc[slot] = (T)0;
}

template <class T>
__global__ void mf_stream_end_kernel(const MfParams<T> p, const long long n_batches) {      // one wavefront
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) p.state->batch_base += n_batches;
    if (threadIdx.x < MU_SLOTS) ring_to_stream_start(p, n_batches, (int)threadIdx.x);
}

template <class T>
__global__ void mf_group_stream_end_kernel(const MfParams<T> *table, const int n_models, const long long n_batches) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_models) return;
    table[m].state->batch_base += n_batches;
    for (int slot = 0; slot < MU_SLOTS; ++slot) ring_to_stream_start(table[m], n_batches, slot);
}

}  // namespace
}  // namespace mi355rec
