// mf.hip -- BPR-MF, FunkSVD and AsySVD mini-batch SGD epochs on MI355X (gfx950).
//
// Replaces MatrixFactorization/Cython/MatrixFactorization_Cython_Epoch.pyx (reference):
//   epochIteration_Cython_BPR_SGD :580-649, epochIteration_Cython_FUNK_SVD_SGD :286-361,
//   sampleBPR_Cython :940-985, sampleMSE_Cython :878-935, _add_*_sample_in_minibatch :737-766,
//   _apply_minibatch_updates_to_latent_factors :770-829, adaptive_gradient :835-873, ASY_SVD :393-541.
//
// Design.  The reference's mini-batch rule -- every gradient of a batch is taken against the
// start-of-batch factors, the gradients of a row are summed, the sum is applied once -- has two consequences:
//   (1) the samples of a batch are independent, and
//   (2) the sample stream does not depend on the factors, so everything about WHO touches WHICH row WHEN is known
//       before the first mini-batch runs.
// Round 1 used (1) only: a scatter kernel (float atomics into accumulators, first-toucher flags) and an apply kernel per
// mini-batch, i.e. two dependent launches and five dependent memory round trips per batch (10.2 us per batch of 1000).
// This version uses (2) as well and turns the batch into ONE gather kernel with no atomics, no flags and no
// accumulators:
//   mf_sample_kernel   one thread per sample of the epoch (counter-based RNG, binary-search rejection);
//   schedule           the epoch's (row, batch) incidences are grouped (in-LDS radix sort per mini-batch, or one rocPRIM sort of
//                      the whole stream) into TASKS: one task per row touched
//                      in a batch, holding the list of that batch's samples which touch the row.  Because every
//                      row's tasks are ordered by batch, the VERSION of a row each sample must read is static: factor
//                      rows live in two buffers, version v of a row in buffer (v & 1); a task reads version v of
//                      every row it needs and writes version v + 1 of its own row into the other buffer, which no
//                      reader of the same batch looks at.  Parities are baked into the sample records;
//   mf_batch_kernel    one wavefront per task: for each sample of the list, gather the 3 (BPR) / 2 (FunkSVD) rows at
//                      their parities, x_uij by DPP / v_permlane swap reduction inside a LPR-lane group (16-byte
//                      loads; a 64-lane wavefront works on 64/LPR samples at a time), own-row gradient summed in
//                      registers in sample order (deterministic), then mean over batch_size, optimiser, += lr * step,
//                      one store of the new row version.
// One dependent launch and three dependent memory round trips per mini-batch (task header -> rows -> store).
// Round 3: rows touched by ONE sample of the mini-batch (most of them) no longer get tasks of their own -- the sample's user
// task updates them too (fused sample tasks), and neighbouring single-sample tasks share a wavefront (pair tasks): 7 row
// transfers per sample instead of 12, HBM traffic 1.22 x the algorithmic 24 k bytes (was 1.84 x); mf_group_batch_kernel runs
// mini-batch b of R independent models as one grid (mi355rec_mf_group_*).  The arithmetic type is the storage type: float32 for plain sgd (north_star),
// float64 factors + moments for adagrad / rmsprop / adam, whose per-component normalisation amplifies float32 rounding
// to O(lr); outputs are float32 either way.
// There is no dense contraction here, hence no MFMA.
#include "common.h"
#include "sampling.cuh"
#include "wave.cuh"

#include <rocprim/rocprim.hpp>

#include <memory>

namespace mi355rec {
namespace {

struct MfState {           // lives in device memory so that graph replays carry no per-epoch host arguments
    long long batch_base;  // mini-batches executed before the stream now in the buffers (Adam's t, global-bias ring)
    long long epoch;       // index (since create) of the epoch the next sampling kernel draws
    double beta_1_power, beta_2_power;   // AsySVD: Adam's running products (advanced once per step, .pyx:536-539)
    double asy_loss;
};

struct TaskHeader {        // 32 bytes, one per (row, mini-batch) incidence
    int entry;             // user row u, or n_users + item
    int meta;              // bit 31: buffer holding the row's current version; bits 0..27: number of samples
    int start;             // first record of the list (recs[], sorted order)
    int pad;
    int4 rec0;             // the first record itself: single-sample tasks (most of them) need no second load
};
constexpr int LEN_MASK = 0x0fffffff;
typedef int int8v __attribute__((ext_vector_type(8)));
// record .w: bits 0-1 role of the task's row in this sample (0 user, 1 item / positive item, 2 negative item),
//            bit 2 / 3 / 4: buffer of the sample's user / item / negative-item row
//            bit 5 / 6 (user role, fast schedule): this task also updates the sample's positive / negative item row
constexpr int ROLE_U = 0, ROLE_I = 1;   // 2: negative item

template <class T> struct MuState { T mu, c1, c2, pad; };
// one L2 atomic per workgroup that carries global-bias terms: atomics on ONE address retire at about 0.2 us each (measured through
// the batch kernel: 31 per address cost 3 us per mini-batch), so the terms are spread over a wavefront's worth of addresses
constexpr int MU_SLOTS = 64;
// (Measured and rejected: no atomics at all -- one cell per workgroup, every wavefront of the next batch folds the ~500 cells in a
// fixed order, which makes the global bias bit-reproducible -- costs 8 loads per lane and wavefront: 157 ms per FunkSVD epoch at
// ML-20M shape against 145 ms with 64 atomic slots and 164 ms with 16.)

template <class T>
struct MfParams {
    int n_users, n_items, k, batch_size;
    int use_bias, sgd_mode, sample_negatives, tasks_per_batch;
    T lr, user_reg, item_reg, bias_reg, positive_reg, negative_reg, inv_batch;
    T gamma, beta_1, beta_2, one_m_gamma, one_m_beta_1, one_m_beta_2;
    float quota;
    double beta_1_d, beta_2_d;
    unsigned long long seed;
    const int *indptr, *indices;
    const float *data;
    T *U0, *U1, *V0, *V1;                // two buffers per factor matrix: version v of a row lives in buffer v & 1
    T *bu0, *bu1, *bi0, *bi1;
    T *c1U, *c2U, *c1V, *c2V;            // optimiser state (one copy: only the row's own task touches it)
    T *c1_bu, *c2_bu, *c1_bi, *c2_bi;
    MuState<T> *mu_state;                // [3] ring: global bias after batch b - 1, written by batch b        (FunkSVD)
    T *mu_acc;                           // [3][MU_SLOTS] ring: batch b's global-bias gradient terms, spread over MU_SLOTS addresses
    T *asy_mu, *asy_c_mu;                // AsySVD: global bias and its optimiser state, updated in place
    unsigned char *par;                  // [n_u_rows + n_items] buffer of every row's current version at stream start
    double *loss_slots;                  // [tasks_per_batch * 4] per (wavefront, group) running loss
    MfState *state;
    // sample stream: one epoch drawn by mf_sample_kernel (native) or the caller's stream (replay)
    int *su, *si, *sj;
    float *sr;
    long long samples_per_epoch;
    // schedule
    const TaskHeader *tasks;
    const int4 *recs;
    const int4 *slot_recs;               // in-LDS schedule: [mini-batch][slot][3] records 1 .. group - 1 of a PAIR task, by header slot: their
    long long slot_rec_stride;           // address does not depend on the header (0: radix-sort schedule, one dummy mini-batch)
    const int *used;                     // fast schedule: header slots in use per mini-batch of the stream (NULL: all of them may be)
    unsigned long long *ticks;           // optional [tasks_per_batch][8] shader-clock stamps of the last mini-batch (MI355REC_MF_TICKS=1)
    int wg_base, wg_stride;              // workgroup b of the launch is workgroup wg_base + b * wg_stride of the mini-batch (exact
                                         // multi-GPU mode: rank r of G runs workgroups r, r + G, ...; otherwise 0 and 1)
};

__device__ __forceinline__ unsigned long long stamp() {   // shader clock; not reordered against memory operations
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ float sigmoid_of_minus(float x) { return 1.f / (1.f + __expf(x)); }   // .pyx:619
__device__ __forceinline__ double sigmoid_of_minus(double x) { return 1.0 / (1.0 + exp(x)); }
__device__ __forceinline__ float root(float x) { return sqrtf(x); }
__device__ __forceinline__ double root(double x) { return sqrt(x); }

// adaptive_gradient (.pyx:835-873) on one cell whose state is passed by reference; pw1/pw2 = 1 - beta^t
template <class T, class P>
__device__ __forceinline__ T adapt_cell(const P &p, T g, T &c1, T &c2, T pw1, T pw2) {
    switch (p.sgd_mode) {
        case MI355REC_ADAGRAD:
            c1 = c1 + g * g;
            return g / (root(c1) + (T)1e-8);
        case MI355REC_RMSPROP:
            c1 = c1 * p.gamma + p.one_m_gamma * (g * g);
            return g / (root(c1) + (T)1e-8);
        case MI355REC_ADAM: {
            c1 = c1 * p.beta_1 + p.one_m_beta_1 * g;
            c2 = c2 * p.beta_2 + p.one_m_beta_2 * (g * g);
            return (c1 / pw1) / (root(c2 / pw2) + (T)1e-8);
        }
        default:
            return g;
    }
}
// the same on a cell in memory
template <class T, class P>
__device__ __forceinline__ T adapt(const P &p, T g, T *c1, T *c2, size_t at, T pw1, T pw2) {
    if (p.sgd_mode == MI355REC_SGD) return g;
    T a = c1[at], b = p.sgd_mode == MI355REC_ADAM ? c2[at] : (T)0;
    const T step = adapt_cell(p, g, a, b, pw1, pw2);
    c1[at] = a;
    if (p.sgd_mode == MI355REC_ADAM) c2[at] = b;
    return step;
}

// The gradient of one sample on one row and the application of a row's summed gradient, with every operation rounded on its own
// (no fused multiply-add, whatever the surrounding code looks like): the same row may be updated by its own task or by the
// sample's user task (fused sample tasks), alone or next to other samples in one launch (replica batches), and the result must not
// depend on which -- the backend contracts a * b - c * d differently from one call site to the next.  This is also how the
// reference's scalar double code rounds.
template <class T> __device__ __forceinline__ T grad_term(T scale, T x, T reg, T w) {       // scale * x - reg * w   (.pyx:626-639, 343-352)
#pragma clang fp contract(off)
    const T a = scale * x;
    const T b = reg * w;
    return a - b;
}
// a * b + c in ONE rounding, spelled out: left to the backend, the sum of products of a dot product came out fused in one
// instantiation of the mini-batch body and as packed multiply + packed add in another (seen in round 4 between a model trained
// alone and the same model inside a group, once the two were built from different instantiations: 4 of 72 480 cells one ulp apart)
__device__ __forceinline__ float fused_add(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fused_add(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <class T> __device__ __forceinline__ T diff_of(T a, T b) {
#pragma clang fp contract(off)
    return a - b;
}
template <class T> __device__ __forceinline__ T mean_of(T sum, T inv_batch) {
#pragma clang fp contract(off)
    return sum * inv_batch;
}
template <class T> __device__ __forceinline__ T moved(T w, T lr, T step) {                   // w + lr * step   (.pyx:809-812)
#pragma clang fp contract(off)
    const T a = lr * step;
    return w + a;
}

// One thread per sample of one epoch (sampleBPR_Cython .pyx:940-985 / sampleMSE_Cython :878-935).
template <int ALGO, class T>
__device__ __forceinline__ void mf_sample_body(const MfParams<T> &p) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long epoch = p.state->epoch;
    if (t < p.samples_per_epoch) {
        const unsigned long long sid = (unsigned long long)(epoch * p.samples_per_epoch + t);   // global sample id
        unsigned d = 0;
        int u, start = 0, n_seen = 0;
        do {   // users with no interactions or with no negative item are skipped (.pyx:950-958)
            u = bounded(draw32(p.seed, sid, d++), p.n_users);
            start = p.indptr[u];
            n_seen = p.indptr[u + 1] - start;
        } while (n_seen == 0 || n_seen == p.n_items);
        const int *row = p.indices + start;
        p.su[t] = u;
        if (ALGO == MI355REC_MF_BPR) {
            p.si[t] = row[bounded(draw32(p.seed, sid, d++), n_seen)];
            int j;
            do { j = bounded(draw32(p.seed, sid, d++), p.n_items); } while (!profile_lacks(row, n_seen, j));
            p.sj[t] = j;
        } else {
            // .pyx:898: a POSITIVE is drawn with probability `quota` (sic); no quota -> always positive
            bool positive = true;
            if (p.sample_negatives) positive = (float)draw32(p.seed, sid, d++) * 2.3283064365386963e-10f <= p.quota;
            if (positive) {
                const int at = bounded(draw32(p.seed, sid, d++), n_seen);
                p.si[t] = row[at];
                p.sr[t] = p.data[start + at];
            } else {
                int i;
                do { i = bounded(draw32(p.seed, sid, d++), p.n_items); } while (!profile_lacks(row, n_seen, i));
                p.si[t] = i;
                p.sr[t] = 0.f;
            }
        }
    }
    // (the epoch counter is advanced by the NEXT kernel on the stream, mf_epoch_advance_kernel: a grid larger than the device's
    // residency -- FunkSVD draws 20 M samples per epoch -- still has blocks to start when the first ones retire, and they must
    // read the same epoch)
}
template <int ALGO, class T>
__global__ __launch_bounds__(256) void mf_sample_kernel(const MfParams<T> p) { mf_sample_body<ALGO, T>(p); }
__global__ void mf_epoch_advance_kernel(MfState *state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state->epoch += 1;
}

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
    int mid_bytes;                // LDS bytes between the keys and the once-touched flags (run starts + header slots, or the sort's scratch)
    int entry_bits;               // bits of a row id (users, then items) + 1: the padding keys' all-ones field sorts last
    int fuse;                     // BPR: a sample whose user row is touched once in the batch takes over its other once-touched rows
    const int *su, *si, *sj;
    const float *sr;
    unsigned *touched;            // [n_entries][words]: bit b of row x = mini-batch b of this stream touches x
    unsigned char *par;           // [n_entries]: buffer of every row's current version at stream start
    int *sorted_slot;             // [n_batches][tasks_per_batch]: batch-local incidence ids in (row, id) order
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
    for (int slot = used + tid; slot < s.tasks_per_batch; slot += SCHED_THREADS)
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
        s.sorted_slot[(size_t)b * s.tasks_per_batch + q] = inc | (also << 16);
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

__device__ __forceinline__ void mf_sched_emit_body(const FastSchedParams &s) {
    const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const long long first = (long long)b * s.batch_size;
    const int n_in = (int)min((long long)s.batch_size, s.n_samples - first);
    if (q >= n_in * s.per) return;
    const size_t at = (size_t)b * s.tasks_per_batch + q;
    const int slot_word = s.sorted_slot[at];
    const int slot = slot_word & 0xffff, also = slot_word >> 16;
    const int smp = slot / s.per, role = slot - smp * s.per;
    const long long t = first + smp;
    const int u = s.su[t], i = s.si[t], j = s.per == 3 ? s.sj[t] : 0;
    const int pu = version_parity(s, u, b), pi = version_parity(s, s.n_users + i, b);
    const int pj = s.per == 3 ? version_parity(s, s.n_users + j, b) : 0;
    const int4 rec = make_int4(u, i, s.per == 3 ? j : __float_as_int(s.sr[t]), role | (pu << 2) | (pi << 3) | (pj << 4) | (also << 5));
    s.recs[at] = rec;
    const int tp = s.qtask[at];
    if (tp == SLOT_ABSORBED) {
        // a sample of a PAIR task other than its first (pairs are aligned blocks of `group` sorted positions, the header belongs to the
        // first): its record also goes where the mini-batch kernel finds it without having seen the header -- by header slot
        const int q0 = q - q % s.group;
        if (s.fuse && q0 != q) {
            const int tp0 = s.qtask[(size_t)b * s.tasks_per_batch + q0];
            if (tp0 != SLOT_ABSORBED && !(tp0 & META_WIDE)) {
                const TaskHeader *lead = s.tasks + (size_t)b * s.tasks_per_batch + tp0;
                if (lead->pad == 1 && lead->start == (int)((size_t)b * s.tasks_per_batch + q0))
                    s.slot_recs[((size_t)b * s.tasks_per_batch + tp0) * 3 + (q - q0 - 1)] = rec;
            }
        }
        return;
    }
    TaskHeader *hd = s.tasks + (size_t)b * s.tasks_per_batch + (tp & (META_WIDE - 1));
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

// ---- the mini-batch --------------------------------------------------------------------------------------------------
template <class T, int VEC> struct alignas(sizeof(T) * VEC) Chunk { T v[VEC]; };

// Loads are issued unconditionally from clamped (always valid) addresses and masked afterwards: no branch sits between
// two loads, so the compiler batches them under one wait.
template <class T, int VEC>
__device__ __forceinline__ Chunk<T, VEC> load_chunk(const T *row, int chunk, bool ok) {
    Chunk<T, VEC> r = *reinterpret_cast<const Chunk<T, VEC> *>(row + (size_t)(ok ? chunk : 0) * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.v[e] = ok ? r.v[e] : (T)0;
    return r;
}

// Adam's 1 - beta^t for the 1-based mini-batch index t
template <class T, class P> __device__ __forceinline__ void adam_powers(const P &p, long long t, T &pw1, T &pw2) {
    pw1 = (T)1;
    pw2 = (T)1;
    if (p.sgd_mode == MI355REC_ADAM) {
        pw1 = (T)(1.0 - pow(p.beta_1_d, (double)t));
        pw2 = (T)(1.0 - pow(p.beta_2_d, (double)t));
    }
}

// Global bias as the batch `gb` must see it (FunkSVD with bias): the value after batch gb - 2 plus batch gb - 1's step,
// computed identically by every wavefront from the ring; wavefront 0 files the result for the next batch.
// In two halves: the ring is REQUESTED before the row gathers of the wavefront's first sample are issued and folded after
// them -- with one call in front of the gathers the fold's wait put the whole ring round trip (batch index -> ring -> sum) in
// front of the gathers: 8.4 us per mini-batch against 5.0 without biases.
template <class T> struct MuRequest { MuState<T> st; T part; };

template <class T>
__device__ __forceinline__ MuRequest<T> global_bias_request(const MfParams<T> &p, const int prev, int lane) {
    MuRequest<T> r;
    r.st = p.mu_state[prev];
    r.part = p.mu_acc[prev * MU_SLOTS + lane];
    return r;
}

template <class T>
__device__ __forceinline__ T global_bias_finish(const MfParams<T> &p, MuRequest<T> r, long long gb, const int at, bool writer, int lane) {
    const int cur = at % 3, nxt = (at + 1) % 3;                   // `at`: position in the stream, gb: global index
    MuState<T> st = r.st;
    const T sum = wave_sum(r.part);
    if (gb > 0) {
        T pw1, pw2;
        adam_powers(p, gb, pw1, pw2);           // the step belongs to batch gb - 1, whose 1-based index is gb
        const T step = adapt_cell(p, sum * p.inv_batch, st.c1, st.c2, pw1, pw2);
        st.mu += p.lr * step;
    }
    if (writer) {
        if (lane == 0) p.mu_state[cur] = st;
        p.mu_acc[nxt * MU_SLOTS + lane] = (T)0;
    }
    return st.mu;
}

template <class T>
__device__ __forceinline__ T global_bias_at(const MfParams<T> &p, long long gb, const int at, bool writer, int lane) {
    return global_bias_finish(p, global_bias_request(p, (at + 2) % 3, lane), gb, at, writer, lane);
}

// the three rows of one sample, KI chunks of VEC elements per lane
template <class T, int VEC, int KI, bool BPR> struct Rows {
    Chunk<T, VEC> A[KI], B[KI], C[BPR ? KI : 1];
    T bu, bi;
};

template <class T, int VEC, int LPR, int KI, bool BPR>
__device__ __forceinline__ Rows<T, VEC, KI, BPR> load_rows(const MfParams<T> &p, const int4 rec, int li, const bool (&cok)[KI],
                                                           bool bias) {
    Rows<T, VEC, KI, BPR> r;
    const int k = p.k;
    const T *Wu = ((rec.w >> 2) & 1 ? p.U1 : p.U0) + (size_t)rec.x * k;
    const T *Hi = ((rec.w >> 3) & 1 ? p.V1 : p.V0) + (size_t)rec.y * k;
    const T *Hj = ((rec.w >> 4) & 1 ? p.V1 : p.V0) + (size_t)(BPR ? rec.z : 0) * k;
#pragma unroll
    for (int c = 0; c < KI; ++c) {
        r.A[c] = load_chunk<T, VEC>(Wu, c * LPR + li, cok[c]);
        r.B[c] = load_chunk<T, VEC>(Hi, c * LPR + li, cok[c]);
        if (BPR) r.C[c] = load_chunk<T, VEC>(Hj, c * LPR + li, cok[c]);
    }
    r.bu = (T)0;
    r.bi = (T)0;
    if (bias) {                                   // wave-uniform
        r.bu = ((rec.w >> 2) & 1 ? p.bu1 : p.bu0)[rec.x];
        r.bi = ((rec.w >> 3) & 1 ? p.bi1 : p.bi0)[rec.y];
    }
    return r;
}

// KI chunks of VEC elements per lane, LPR lanes per row (64 / LPR samples of a task's list in flight per wavefront).
// `wg` = this workgroup's index within the mini-batch's launch of ONE model (blockIdx.x; the group launch below puts the model
// on blockIdx.y).
template <int ALGO, class T, int VEC, int LPR, int KI>
__device__ __forceinline__ void mf_batch_body(const MfParams<T> &p, const int batch_local, const int wg) {
    constexpr int G = 64 / LPR;
    constexpr bool BPR = ALGO == MI355REC_MF_BPR;
    using Ch = Chunk<T, VEC>;
    using R = Rows<T, VEC, KI, BPR>;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((wg * p.wg_stride + p.wg_base) * 4 + (threadIdx.x >> 6));
    const unsigned long long tk0 = p.ticks ? stamp() : 0ull;
    // every wave-uniform input is requested before the first one is waited for (scalar loads, one wait)
    // (unused task slots of a batch are zero: a header with no samples means there is nothing to do)
    // (the grid is rounded up to whole workgroups: wavefronts past the batch's last slot re-read that slot and idle)
    const bool bias = !BPR && p.use_bias;
    long long gb = batch_local;
    const TaskHeader *hp = p.tasks + ((size_t)batch_local * p.tasks_per_batch + min(wv, p.tasks_per_batch - 1));
    const int8v hd = *reinterpret_cast<const int8v *>(hp);      // one 32-byte load: header and first record
    // The global mini-batch index (Adam and the global bias need it) and the global-bias ring entry are requested right behind the
    // header (the wait for the header does not cover younger loads; both were written by the kernel before this one and take
    // 2 400 cycles to arrive where the header, last written by the schedule, takes 900) -- not after it has arrived, and the ring
    // not after the index: a FunkSVD kernel began with three round trips one after the other.  Unconditionally for FunkSVD: a
    // branch around a load makes the compiler wait for it before the next one is issued.
    // a pair task's other records: by slot, requested with the header (through the record list they were a dependent round trip
    // in front of the row gathers of the lane groups 1 .. G - 1: 2 300 cycles until the rows were there against 1 800 for one sample)
    int4 slot_rec = make_int4(0, 0, 0, 0);
    if constexpr (BPR && G > 1)
        slot_rec = p.slot_recs[(size_t)batch_local * p.slot_rec_stride + (size_t)min(wv, p.tasks_per_batch - 1) * 3 + max(lane / LPR - 1, 0)];
    long long batch_base = 0;
    MuRequest<T> mu_req;
    mu_req.st = MuState<T>{};
    mu_req.part = (T)0;
    if constexpr (!BPR) {
        // (through a zero the compiler cannot see: it moves the result of a load it knows to be wave-uniform into scalar
        // registers on the spot, which is a wait for these loads in front of the row gathers)
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        batch_base = (&p.state->batch_base)[zero];
        mu_req = global_bias_request(p, (batch_local + 2) % 3 + zero, lane);
    }
    // (the kernel arguments the row gathers need are requested now, next to the header, rather than in a second scalar
    // round trip after the header has arrived)
    asm volatile("" ::"s"(p.k), "s"(p.U0), "s"(p.U1), "s"(p.V0), "s"(p.V1), "s"(p.recs));
    if constexpr (BPR) {
        if (p.sgd_mode == MI355REC_ADAM) gb += p.state->batch_base;
    } else {
        gb += batch_base;
    }
    const int4 h0 = make_int4(hd[0], hd[1], hd[2], hd[3]), h1 = make_int4(hd[4], hd[5], hd[6], hd[7]);
    const bool active = (h0.y & LEN_MASK) != 0 && wv < p.tasks_per_batch;
    __shared__ T s_mu[4];
    __shared__ T s_wide[4][LPR * KI * VEC];
    __shared__ T s_wide_bias[4];
    T mu_term = (T)0;
    T mu_eff = (T)0;
    unsigned long long tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, tk5 = 0;
    if (p.ticks) tk1 = stamp();          // header has arrived (its value decided `active`)
    if (bias && !active) mu_eff = global_bias_finish(p, mu_req, gb, batch_local, wv == 0, lane);      // (wavefront 0 files the value either way)
    if (active) {
        const int entry = h0.x, len = h0.y & LEN_MASK, own_par = (unsigned)h0.y >> 31, start = h0.z;
        // a wide task (list longer than two rounds of a wavefront) owns the 4 wavefronts of this workgroup: quarter `part` takes list
        // positions part * G + g, then every 4 * G; h1 is the record at part * G
        const bool wide = (h0.y & META_WIDE) != 0;
        const int part = (h0.y >> 28) & 3;
        const int base = wide ? part * G : 0, step = wide ? 4 * G : G;
        const int g = lane / LPR, li = lane % LPR;
        const int k = p.k, chunks = k / VEC;
        bool cok[KI];
#pragma unroll
        for (int c = 0; c < KI; ++c) cok[c] = c * LPR + li < chunks;
        const int iters = len > base ? (len - base + step - 1) / step : 0;      // (a short wide list leaves late quarters empty)
        // software pipeline: records two list positions ahead of the arithmetic, rows one ahead.  Positions past the
        // end of the list are clamped to the last record (valid addresses) and contribute nothing.
        int4 rec = h1;
        // PAIR task (fast schedule, BPR): G single-sample user tasks share this wavefront, lane group g has sample g of the "list"
        // -- its own row to write, nothing to sum across groups
        const bool pair = BPR && G > 1 && h0.w == 1;
        T sg_first = (T)0;
        if (BPR && G > 1 && h0.w == 1) {           // pair task: the lane groups' records came with the header
            if (g != 0) rec = slot_rec;
        } else if (G > 1 && len > 1) {             // single-sample tasks (most of them) go straight from the header to the rows
            const int4 r = p.recs[start + min(base + g, len - 1)];
            if (g != 0) rec = r;
        }
        const int4 rec_first = rec;                // single-sample and pair tasks: THE record of this lane group
        int4 rec_n = rec;
        if (iters > 1) rec_n = p.recs[start + min(base + step + g, len - 1)];
        R rows = load_rows<T, VEC, LPR, KI, BPR>(p, rec, li, cok, bias);
        if (bias) mu_eff = global_bias_finish(p, mu_req, gb, batch_local, wv == 0, lane);   // folded behind the gathers just issued
        T pw1, pw2;
        adam_powers(p, gb + 1, pw1, pw2);

        Ch acc[KI], own[KI];
#pragma unroll
        for (int c = 0; c < KI; ++c)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { acc[c].v[e] = (T)0; own[c].v[e] = (T)0; }
        T bias_acc = (T)0, own_bias = (T)0;
        double loss = 0.0;

        for (int it = 0; it < iters; ++it) {
            const int idx = base + it * step + g;
            const bool valid = idx < len;
            // issue the next position's loads before this position's arithmetic (wave-uniform conditions)
            int4 rec_nn = rec_n;
            if (it + 2 < iters) rec_nn = p.recs[start + min(idx + 2 * step, len - 1)];
            R rows_n = rows;
            if (it + 1 < iters) rows_n = load_rows<T, VEC, LPR, KI, BPR>(p, rec_n, li, cok, bias);

            const int role = rec.w & 3;
            T dot = (T)0;
#pragma unroll
            for (int c = 0; c < KI; ++c)
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    dot = fused_add(rows.A[c].v[e], BPR ? diff_of(rows.B[c].v[e], rows.C[c].v[e]) : rows.B[c].v[e], dot);
            dot = group_sum<LPR>(dot);
            if (p.ticks && it == 0) {
                asm volatile("" ::"v"(dot));
                tk2 = stamp();           // first rows have arrived
            }
            if (BPR) {
                const T x = dot;
                const T sg = sigmoid_of_minus(x);
                if (it == 0) sg_first = sg;
                if (valid && role == ROLE_U && li == 0) loss += (double)x * (double)x;
#pragma unroll
                for (int c = 0; c < KI; ++c)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const T a = rows.A[c].v[e], b = rows.B[c].v[e], cc = rows.C[c].v[e];
                        const T gU = grad_term(sg, diff_of(b, cc), p.user_reg, a);        // .pyx:626-639
                        const T gI = grad_term(sg, a, p.positive_reg, b);
                        const T gJ = grad_term(sg, -a, p.negative_reg, cc);
                        const T gr = role == ROLE_U ? gU : (role == ROLE_I ? gI : gJ);
                        acc[c].v[e] += valid ? gr : (T)0;
                        if (it == 0) own[c].v[e] = role == ROLE_U ? a : (role == ROLE_I ? b : cc);
                    }
            } else {
                T pred = dot;
                if (bias) pred += mu_eff + rows.bu + rows.bi;
                const T err = (T)__int_as_float(rec.z) - pred;
                if (valid && role == ROLE_U) {
                    if (li == 0) loss += (double)err * (double)err;
                    if (bias) mu_term += err - p.bias_reg * mu_eff;            // .pyx:329-336
                }
                if (bias && valid) bias_acc += err - p.bias_reg * (role == ROLE_U ? rows.bu : rows.bi);
                if (it == 0) own_bias = role == ROLE_U ? rows.bu : rows.bi;
#pragma unroll
                for (int c = 0; c < KI; ++c)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const T a = rows.A[c].v[e], b = rows.B[c].v[e];
                        // NB the item gradient is regularised with positive_reg (sic, .pyx:346), never item_reg
                        const T gU = grad_term(err, b, p.user_reg, a);
                        const T gI = grad_term(err, a, p.positive_reg, b);
                        const T gr = role == ROLE_U ? gU : gI;
                        acc[c].v[e] += valid ? gr : (T)0;
                        if (it == 0) own[c].v[e] = role == ROLE_U ? a : b;
                    }
            }
            rec = rec_n;
            rec_n = rec_nn;
            rows = rows_n;
        }
        // totals over the groups, in a fixed order
        if (G > 1 && !pair) {
#pragma unroll
            for (int c = 0; c < KI; ++c)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[c].v[e] = cross_group_sum<LPR>(acc[c].v[e]);
            if (bias) bias_acc = cross_group_sum<LPR>(bias_acc);
        }
        if (bias) {   // every lane of a group carries the group's terms: one lane per group counts
            mu_term = li == 0 ? mu_term : (T)0;
            mu_term = wave_sum(mu_term);
        }
        // (wavefront, group) slots are private; an atomic without return value instead of load + add + store keeps a
        // dependent memory round trip out of the tail of the wavefront
        if (li == 0 && loss != 0.0) atomicAdd(&p.loss_slots[wv * 4 + g], loss);
        if (p.ticks) tk3 = stamp();      // list done
        if (wide) {                      // the four quarters meet in LDS and are summed in quarter order by the first
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < KI; ++c)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) s_wide[part][(c * VEC + e) * LPR + li] = acc[c].v[e];
                if (li == 0) s_wide_bias[part] = bias_acc;
            }
            __syncthreads();
            if (part == 0 && g == 0) {
#pragma unroll
                for (int c = 0; c < KI; ++c)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int at = (c * VEC + e) * LPR + li;
                        acc[c].v[e] = ((s_wide[0][at] + s_wide[1][at]) + s_wide[2][at]) + s_wide[3][at];
                    }
                bias_acc = ((s_wide_bias[0] + s_wide_bias[1]) + s_wide_bias[2]) + s_wide_bias[3];
            }
        }
        // _apply_minibatch_updates_to_latent_factors (.pyx:770-829): mean over batch_size (NOT over the row's count)
        if ((g == 0 || pair) && (!wide || part == 0)) {
            const int own_entry = pair ? rec_first.x : entry;                       // (a pair task's rows are user rows)
            const int own_buf = pair ? (rec_first.w >> 2) & 1 : own_par;
            const bool is_item = own_entry >= p.n_users;
            const int row = is_item ? own_entry - p.n_users : own_entry;
            T *Wn = (is_item ? (own_buf ? p.V0 : p.V1) : (own_buf ? p.U0 : p.U1)) + (size_t)row * k;
            T *c1 = (is_item ? p.c1V : p.c1U) + (size_t)row * k, *c2 = (is_item ? p.c2V : p.c2U) + (size_t)row * k;
#pragma unroll
            for (int c = 0; c < KI; ++c) {
                if (!cok[c]) continue;
                const size_t at = (size_t)(c * LPR + li) * VEC;
                Ch m1, m2, out;
                if (p.sgd_mode != MI355REC_SGD) m1 = *reinterpret_cast<const Ch *>(c1 + at);
                if (p.sgd_mode == MI355REC_ADAM) m2 = *reinterpret_cast<const Ch *>(c2 + at);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const T gm = mean_of(acc[c].v[e], p.inv_batch);
                    const T step = adapt_cell(p, gm, m1.v[e], m2.v[e], pw1, pw2);
                    out.v[e] = moved(own[c].v[e], p.lr, step);
                }
                *reinterpret_cast<Ch *>(Wn + at) = out;
                if (p.sgd_mode != MI355REC_SGD) *reinterpret_cast<Ch *>(c1 + at) = m1;
                if (p.sgd_mode == MI355REC_ADAM) *reinterpret_cast<Ch *>(c2 + at) = m2;
            }
            if (bias && li == 0) {
                T *bn = is_item ? (own_buf ? p.bi0 : p.bi1) : (own_buf ? p.bu0 : p.bu1);
                T *b1 = is_item ? p.c1_bi : p.c1_bu, *b2 = is_item ? p.c2_bi : p.c2_bu;
                const T step = adapt(p, bias_acc * p.inv_batch, b1, b2, (size_t)row, pw1, pw2);
                bn[row] = own_bias + p.lr * step;
            }
        }
        if (p.ticks) tk4 = stamp();      // own row written
        const int also = BPR && (len == 1 || pair) ? (rec_first.w >> 5) & 3 : 0;
        if (BPR && also) {
            // The item rows this single-sample user task took over (mf_sched_sort_kernel): the arithmetic their own tasks would
            // have done -- gradient of one sample, mean over batch_size, optimiser, one store of the next row version.  In a
            // single task every group of the wavefront holds the same record, rows and sigmoid (positions past the end of the
            // list are clamped to the last record), so the rows are dealt to the groups: row e (1 positive, 2 negative item) to
            // group e % G; in a pair task every group looks after its own sample.
            // (`rows` still holds the first record's rows: these lists run one iteration and load nothing else)
#pragma unroll
            for (int e = 1; e <= 2; ++e) {
                if (!(also & e) || !(pair || g == e % G)) continue;
                const int item = e == 1 ? rec_first.y : rec_first.z;
                const int cur = (rec_first.w >> (e == 1 ? 3 : 4)) & 1;               // buffer of the version just read
                T *Wn = (cur ? p.V0 : p.V1) + (size_t)item * k;
                T *c1 = p.c1V + (size_t)item * k, *c2 = p.c2V + (size_t)item * k;
#pragma unroll
                for (int c = 0; c < KI; ++c) {
                    if (!cok[c]) continue;
                    const size_t at = (size_t)(c * LPR + li) * VEC;
                    Ch m1, m2, out;
                    if (p.sgd_mode != MI355REC_SGD) m1 = *reinterpret_cast<const Ch *>(c1 + at);
                    if (p.sgd_mode == MI355REC_ADAM) m2 = *reinterpret_cast<const Ch *>(c2 + at);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const T a = rows.A[c].v[v], b = rows.B[c].v[v], cc = rows.C[c].v[v];
                        const T gr = e == 1 ? grad_term(sg_first, a, p.positive_reg, b) : grad_term(sg_first, -a, p.negative_reg, cc);   // .pyx:632-639
                        const T gm = mean_of((T)0 + gr, p.inv_batch);
                        const T step = adapt_cell(p, gm, m1.v[v], m2.v[v], pw1, pw2);
                        out.v[v] = moved(e == 1 ? b : cc, p.lr, step);
                    }
                    *reinterpret_cast<Ch *>(Wn + at) = out;
                    if (p.sgd_mode != MI355REC_SGD) *reinterpret_cast<Ch *>(c1 + at) = m1;
                    if (p.sgd_mode == MI355REC_ADAM) *reinterpret_cast<Ch *>(c2 + at) = m2;
                }
            }
        }
    }
    if (p.ticks) tk5 = stamp();          // item rows the task took over written
    if (bias) {   // the batch's global-bias terms: per workgroup through LDS, then one atomic on one of 16 addresses
        if (lane == 0) s_mu[threadIdx.x >> 6] = mu_term;
        __syncthreads();
        if (threadIdx.x == 0) {
            const T sum = (s_mu[0] + s_mu[1]) + (s_mu[2] + s_mu[3]);
            if (sum != (T)0) atomicAdd(&p.mu_acc[(batch_local % 3) * MU_SLOTS + (wg & (MU_SLOTS - 1))], sum);
        }
    }
    if (p.ticks && lane == 0 && wv < p.tasks_per_batch) {
        unsigned long long *o = p.ticks + (size_t)wv * 8;
        o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = tk3; o[4] = stamp(); o[5] = (unsigned long long)(h0.y & LEN_MASK);
        o[6] = tk4; o[7] = tk5;
    }
}

// PLAIN_SGD: the instance for sgd_mode == "sgd" (the reference's default, the headline): every branch on the optimiser is decided at
// compile time.  The update of the item rows a pair task took over ran through 1 600 instructions of optimiser cases -- 1 730 cycles
// of the 6 500 a wavefront lives (MI355REC_MF_TICKS, round 4).
template <int ALGO, class T, int VEC, int LPR, int KI, bool PLAIN_SGD>
__global__ __launch_bounds__(256) void mf_batch_kernel(const MfParams<T> p, const int batch_local) {
    // (an assumption about the argument, not a modified copy: a copy that is passed on by reference lands in scratch memory)
    if constexpr (PLAIN_SGD) __builtin_assume(p.sgd_mode == MI355REC_SGD);
    // Every kernel argument the start of the kernel needs is requested in ONE batch of scalar loads: left to itself the compiler
    // fetched them piecewise as the code came to need them -- three waits for a cold kernarg segment before FunkSVD's header load
    // was even issued, one for BPR's.
    asm volatile("" ::"s"(p.tasks), "s"(p.tasks_per_batch), "s"(p.wg_base), "s"(p.wg_stride), "s"(p.ticks), "s"(p.use_bias), "s"(p.sgd_mode),
                 "s"(p.state), "s"(p.mu_state), "s"(p.mu_acc), "s"(p.k), "s"(p.U0), "s"(p.U1), "s"(p.V0), "s"(p.V1), "s"(p.recs));
    // (here and not in the body: the group launch below reads its parameters from a table in memory, where holding them all in
    // scalar registers from the start costs occupancy)
    // (Measured and rejected, round 4: a launch over a third of the slots with a loop over the slots in use, as the group launch does --
    // BPR 198 against 196 M samples/s, FunkSVD, whose slots are nearly all in use, 102 against 161 M.)
    mf_batch_body<ALGO, T, VEC, LPR, KI>(p, batch_local, blockIdx.x);
}

// REPLICA-BATCHED launch: mini-batch `batch_local` of R independent models in one grid (blockIdx.y = model).  A single model's
// epoch is a chain of dependent mini-batches of ~3 MB each -- a launch fills a tenth of the chip, and concurrent replicas on R
// streams still pay one dispatch per model and mini-batch at the command processor.  Here the chain keeps its length but every
// link carries R mini-batches.  The models share nothing but the kernel instance (algorithm, storage type, lanes per row) and the
// number of task slots per mini-batch: factors, hyper-parameters, seeds, optimiser, even k within the instance's range are per
// model (the table row is the model's MfParams, read through the scalar cache: wave-uniform address, nothing stored before it).
// Pointers that arrive as kernel arguments are known to point to global memory; pointers read from a table are generic ("flat")
// to the compiler, which then gathers with flat_load and cannot use the scalar cache for the task header.  The
// assumption below (neither LDS nor scratch) is what the address-space inference needs to use global_load / s_load again.
template <class P> __device__ __forceinline__ P *as_global(P *q) {
    const unsigned long long bits = (unsigned long long)q;
    return (P *)(__attribute__((address_space(1))) P *)bits;
}
template <class T> __device__ __forceinline__ void globalize(MfParams<T> &p) {
    p.indptr = as_global(p.indptr); p.indices = as_global(p.indices); p.data = as_global(p.data);
    p.U0 = as_global(p.U0); p.U1 = as_global(p.U1); p.V0 = as_global(p.V0); p.V1 = as_global(p.V1);
    p.bu0 = as_global(p.bu0); p.bu1 = as_global(p.bu1); p.bi0 = as_global(p.bi0); p.bi1 = as_global(p.bi1);
    p.c1U = as_global(p.c1U); p.c2U = as_global(p.c2U); p.c1V = as_global(p.c1V); p.c2V = as_global(p.c2V);
    p.c1_bu = as_global(p.c1_bu); p.c2_bu = as_global(p.c2_bu); p.c1_bi = as_global(p.c1_bi); p.c2_bi = as_global(p.c2_bi);
    p.mu_state = as_global(p.mu_state); p.mu_acc = as_global(p.mu_acc);
    p.loss_slots = as_global(p.loss_slots); p.state = as_global(p.state);
    p.su = as_global(p.su); p.si = as_global(p.si); p.sj = as_global(p.sj); p.sr = as_global(p.sr);
    p.tasks = as_global(p.tasks); p.recs = as_global(p.recs); p.ticks = as_global(p.ticks); p.used = as_global(p.used);
    p.slot_recs = as_global(p.slot_recs);
}

template <int ALGO, class T, int VEC, int LPR, int KI, bool PLAIN_SGD>
__global__ __launch_bounds__(256, (PLAIN_SGD && ALGO == MI355REC_MF_BPR && sizeof(T) == 4 && LPR == 32) ? 6 : 1) void mf_group_batch_kernel(const MfParams<T> *__restrict__ table, const int batch_local) {
    MfParams<T> p = table[blockIdx.y];
    globalize(p);
    if constexpr (PLAIN_SGD) __builtin_assume(p.sgd_mode == MI355REC_SGD);       // (every member runs plain sgd: see mf_batch_kernel)
    // the grid covers a third of a mini-batch's header slots; with fused / paired tasks fewer than that are in use as a rule
    // (the in-LDS schedule files the count), and a workgroup that finds more walks on: no wavefront is launched for an empty slot
    const int used = p.used ? p.used[batch_local] : p.tasks_per_batch;
    for (int wg = blockIdx.x; wg * 4 < used; wg += gridDim.x) mf_batch_body<ALGO, T, VEC, LPR, KI>(p, batch_local, wg);
}
// Sampler and schedule of every member in ONE launch each (model on the last grid dimension): as 4 x R small launches on R
// streams they took a third of a 32-model epoch.
__device__ __forceinline__ void globalize(FastSchedParams &f) {
    f.su = as_global(f.su); f.si = as_global(f.si); f.sj = as_global(f.sj); f.sr = as_global(f.sr);
    f.touched = as_global(f.touched); f.par = as_global(f.par); f.sorted_slot = as_global(f.sorted_slot); f.qtask = as_global(f.qtask);
    f.used = as_global(f.used); f.tasks = as_global(f.tasks); f.recs = as_global(f.recs); f.slot_recs = as_global(f.slot_recs);
}
template <int ALGO, class T>
__global__ __launch_bounds__(256) void mf_group_sample_kernel(const MfParams<T> *__restrict__ table) {
    MfParams<T> p = table[blockIdx.y];
    globalize(p);
    mf_sample_body<ALGO, T>(p);
}
template <class T>
__global__ void mf_group_epoch_advance_kernel(const MfParams<T> *table, const int n_models) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < n_models) table[m].state->epoch += 1;
}
__global__ __launch_bounds__(SCHED_THREADS) void mf_group_sched_sort_kernel(const FastSchedParams *__restrict__ table) {
    FastSchedParams f = table[blockIdx.y];
    globalize(f);
    mf_sched_sort_body(f, blockIdx.x);
}
__global__ __launch_bounds__(256) void mf_group_sched_emit_kernel(const FastSchedParams *__restrict__ table) {
    FastSchedParams f = table[blockIdx.z];
    globalize(f);
    mf_sched_emit_body(f);
}
__global__ __launch_bounds__(256) void mf_group_sched_finish_kernel(const FastSchedParams *__restrict__ table) {
    FastSchedParams f = table[blockIdx.y];
    globalize(f);
    mf_sched_finish_body(f);
}

// Any k (odd k, k > 64 lanes x 2 chunks): one task per wavefront, one sample at a time, rows re-read for the update.
template <int ALGO, class T>
__global__ __launch_bounds__(256) void mf_batch_generic_kernel(const MfParams<T> p, const int batch_local) {
    constexpr bool BPR = ALGO == MI355REC_MF_BPR;
    constexpr int KMAX_REG = 8;    // k <= 512 keeps the own-row gradient in registers, larger k is rejected at create
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((blockIdx.x * p.wg_stride + p.wg_base) * 4 + (threadIdx.x >> 6));
    const TaskHeader *hp = p.tasks + ((size_t)batch_local * p.tasks_per_batch + min(wv, p.tasks_per_batch - 1));
    const int4 h0 = *reinterpret_cast<const int4 *>(hp);
    const bool active = (h0.y & LEN_MASK) != 0 && wv < p.tasks_per_batch;
    const bool bias = !BPR && p.use_bias;
    __shared__ T s_mu[4];
    T mu_term = (T)0;
    if (active || bias) {
        const long long gb = p.state->batch_base + batch_local;
        T mu_eff = (T)0;
        if (bias) mu_eff = global_bias_at(p, gb, batch_local, wv == 0, lane);
        if (active) {
            const int entry = h0.x, len = h0.y & LEN_MASK, own_par = (unsigned)h0.y >> 31, start = h0.z;
            const int k = p.k;
            T pw1, pw2;
            adam_powers(p, gb + 1, pw1, pw2);
            T acc[KMAX_REG];
#pragma unroll
            for (int c = 0; c < KMAX_REG; ++c) acc[c] = (T)0;
            T bias_acc = (T)0;
            double loss = 0.0;
            for (int idx = 0; idx < len; ++idx) {
                const int4 rec = idx == 0 ? hp->rec0 : p.recs[start + idx];
                const int role = rec.w & 3;
                const T *Wu = ((rec.w >> 2) & 1 ? p.U1 : p.U0) + (size_t)rec.x * k;
                const T *Hi = ((rec.w >> 3) & 1 ? p.V1 : p.V0) + (size_t)rec.y * k;
                const T *Hj = ((rec.w >> 4) & 1 ? p.V1 : p.V0) + (size_t)(BPR ? rec.z : 0) * k;
                T dot = (T)0;
                for (int f = lane; f < k; f += 64) dot += BPR ? Wu[f] * (Hi[f] - Hj[f]) : Wu[f] * Hi[f];
                dot = wave_sum(dot);
                if (BPR) {
                    const T sg = sigmoid_of_minus(dot);
                    if (role == ROLE_U) loss += (double)dot * (double)dot;
#pragma unroll
                    for (int c = 0; c < KMAX_REG; ++c) {
                        const int f = lane + 64 * c;
                        if (f < k) {
                            const T a = Wu[f], b = Hi[f], cc = Hj[f];
                            acc[c] += role == ROLE_U ? sg * (b - cc) - p.user_reg * a
                                                     : (role == ROLE_I ? sg * a - p.positive_reg * b : sg * (-a) - p.negative_reg * cc);
                        }
                    }
                } else {
                    T bu_v = (T)0, bi_v = (T)0;
                    if (bias) {
                        bu_v = ((rec.w >> 2) & 1 ? p.bu1 : p.bu0)[rec.x];
                        bi_v = ((rec.w >> 3) & 1 ? p.bi1 : p.bi0)[rec.y];
                    }
                    const T err = __int_as_float(rec.z) - (dot + (bias ? mu_eff + bu_v + bi_v : (T)0));
                    if (role == ROLE_U) {
                        loss += (double)err * (double)err;
                        if (bias) mu_term += err - p.bias_reg * mu_eff;
                    }
                    if (bias) bias_acc += err - p.bias_reg * (role == ROLE_U ? bu_v : bi_v);
#pragma unroll
                    for (int c = 0; c < KMAX_REG; ++c) {
                        const int f = lane + 64 * c;
                        if (f < k) {
                            const T a = Wu[f], b = Hi[f];
                            acc[c] += role == ROLE_U ? err * b - p.user_reg * a : err * a - p.positive_reg * b;
                        }
                    }
                }
            }
            if (lane == 0 && loss != 0.0) p.loss_slots[wv * 4] += loss;
            const bool is_item = entry >= p.n_users;
            const int row = is_item ? entry - p.n_users : entry;
            const T *Wo = (is_item ? (own_par ? p.V1 : p.V0) : (own_par ? p.U1 : p.U0)) + (size_t)row * k;
            T *Wn = (is_item ? (own_par ? p.V0 : p.V1) : (own_par ? p.U0 : p.U1)) + (size_t)row * k;
            T *c1 = is_item ? p.c1V : p.c1U, *c2 = is_item ? p.c2V : p.c2U;
#pragma unroll
            for (int c = 0; c < KMAX_REG; ++c) {
                const int f = lane + 64 * c;
                if (f < k) {
                    const T step = adapt(p, acc[c] * p.inv_batch, c1, c2, (size_t)row * k + f, pw1, pw2);
                    Wn[f] = Wo[f] + p.lr * step;
                }
            }
            if (bias && lane == 0) {
                const T *bo = is_item ? (own_par ? p.bi1 : p.bi0) : (own_par ? p.bu1 : p.bu0);
                T *bn = is_item ? (own_par ? p.bi0 : p.bi1) : (own_par ? p.bu0 : p.bu1);
                T *b1 = is_item ? p.c1_bi : p.c1_bu, *b2 = is_item ? p.c2_bi : p.c2_bu;
                const T step = adapt(p, bias_acc * p.inv_batch, b1, b2, (size_t)row, pw1, pw2);
                bn[row] = bo[row] + p.lr * step;
            }
            mu_term = lane == 0 ? mu_term : (T)0;     // every lane computed the same terms
        }
    }
    if (bias) {
        if (lane == 0) s_mu[threadIdx.x >> 6] = mu_term;
        __syncthreads();
        if (threadIdx.x == 0) {
            const T sum = (s_mu[0] + s_mu[1]) + (s_mu[2] + s_mu[3]);
            if (sum != (T)0) atomicAdd(&p.mu_acc[(batch_local % 3) * MU_SLOTS + (blockIdx.x & (MU_SLOTS - 1))], sum);
        }
    }
}

// Current version of every row as float32 (the getters of .pyx:685-702), and the global bias after the last batch.
// ---- exact multi-GPU mini-batches -----------------------------------------------------------
// Every rank holds the same factors and the same schedule; the workgroups (4 task slots: a split list's quarters stay together) of
// a mini-batch are dealt round-robin to the ranks -- headers are packed at the front of a batch's slots, so contiguous shares
// would leave the last ranks idle -- and the rows the tasks of rank r own get their new version on rank r only.  PACK copies
// them into the rank's exchange slab (slab row = 4 * (workgroup / world) + slot % 4); after the all-gather the other ranks'
// slabs are copied into the same rows (!PACK), and every rank holds bit-identical factors again.  One wavefront per slot.
template <class T, bool PACK>
__global__ __launch_bounds__(256) void mf_shard_rows_kernel(const MfParams<T> p, const int batch_local, const int rank, const int world,
                                                            const int slots_per_rank, T *slab) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= p.tasks_per_batch) return;
    const int wg = slot >> 2, owner = wg % world;
    if ((owner == rank) != PACK) return;
    const TaskHeader *hd = p.tasks + ((size_t)batch_local * p.tasks_per_batch + slot);
    const int meta = hd->meta;
    if ((meta & LEN_MASK) == 0) return;                                           // empty slot
    if ((meta & META_WIDE) && ((meta >> 28) & 3) != 0) return;                    // quarters 1..3 of a wide list do not write
    const int entry = hd->entry, own_par = (unsigned)meta >> 31;
    const bool is_item = entry >= p.n_users;
    const int row = is_item ? entry - p.n_users : entry;
    T *Wn = (is_item ? (own_par ? p.V0 : p.V1) : (own_par ? p.U0 : p.U1)) + (size_t)row * p.k;
    // slab layout: [rank][slot within the rank][k]; PACK addresses the rank's own slab, !PACK the gathered one
    const int local = (wg / world) * 4 + (slot & 3);
    T *at = slab + ((size_t)(PACK ? 0 : owner) * slots_per_rank + local) * p.k;
    for (int e = lane; e < p.k; e += 64) {
        if (PACK) at[e] = Wn[e];
        else Wn[e] = at[e];
    }
}

template <class T, class O>
__global__ __launch_bounds__(256) void mf_gather_rows_kernel(const T *b0, const T *b1, const unsigned char *par, long long n_rows,
                                                             int k, O *out) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= n_rows * k) return;
    const long long row = t / k;
    out[t] = (O)(par[row] ? b1[t] : b0[t]);
}
template <class T, class O>
__global__ void mf_final_mu_kernel(const MfParams<T> p, O *out) {
    const int lane = threadIdx.x & 63;
    const T mu = global_bias_at(p, p.state->batch_base, 0, false, lane);      // (between streams: the ring stands at a stream's start)
    if (threadIdx.x == 0) out[0] = (O)mu;
}

// AsySVD (.pyx:393-541): batch_size is 1 and every step rewrites all the Y rows of the sampled user's profile, which
// nearly every other profile shares -- consecutive steps are one dependent chain, executed strictly in order by ONE 1024-thread
// workgroup (16 wavefronts across the profile rows, lanes across the factors).  p.U0 is the n_items x k matrix Y
// ("USER_factors" in the reference), p.V0 the item factors X; nothing is double-buffered here.
// What a step may not overlap with its predecessor is the Y / X / bias traffic; everything else is SOFTWARE-PIPELINED one step
// ahead: the next step's sample, its CSR bounds and the profile ids of its first rows are loaded while the current step reduces
// and updates (they come from read-only arrays), the first ASY_ROWS rows of each wavefront stay in registers between the gather
// and the update, and the loads of a step are issued from clamped addresses in one batch (a load inside a conditional is waited
// for where its branch ends).  Per step that leaves: one gather round trip (L2), two LDS reductions, the scalar part, the stores.
constexpr int ASY_KMAX = 256;
constexpr int ASY_CELLS = 12;    // factors per lane that stay in registers between the gather and the update: C chunks of 64 factors x R rows
// C = chunks of 64 factors a row needs (1: k <= 64, 2: k <= 128, 4: k <= 256); R = ASY_CELLS / C profile rows per wavefront stay
// in registers (192 / 96 / 48 rows per step: the mean ML-1M profile has 166; 16 cells spill at float64)
template <class T, int C>
__global__ __launch_bounds__(1024) void mf_asy_kernel(const MfParams<T> p, const long long first, const int count) {
    constexpr int R = ASY_CELLS / C;
    __shared__ T s_part[16][ASY_KMAX];
    __shared__ T s_acc[ASY_KMAX], s_xi[ASY_KMAX];
    __shared__ T s_err, s_pw1, s_pw2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = p.k;
    T *const Ymat = p.U0, *const Xmat = p.V0, *const bu = p.bu0, *const bi = p.bi0;
    double b1p = 0.0, b2p = 0.0, loss = 0.0;
    if (tid == 0) {
        b1p = p.state->beta_1_power;
        b2p = p.state->beta_2_power;
        loss = p.state->asy_loss;
    }
    // this lane's factors of a row: f = lane + 64 c; fc = the same clamped into the row (always a valid address)
    int fc[C];
    bool fok[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        fok[c] = lane + 64 * c < k;
        fc[c] = min(lane + 64 * c, k - 1);
    }
    // the first step's sample, bounds and first row ids
    int u = 0, i = 0, rs = 0, re = 1, rowid[R];
    T rating = (T)0;
#pragma unroll
    for (int m = 0; m < R; ++m) rowid[m] = -1;
    if (count > 0) {
        u = p.su[first];
        i = p.si[first];
        rating = (T)p.sr[first];
        rs = p.indptr[u];
        re = p.indptr[u + 1];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int q = rs + wave + 16 * m;
            const int id = p.indices[min(q, re - 1)];
            rowid[m] = q < re ? id : -1;
        }
    }
    for (int s = 0; s < count; ++s) {
        // ---- gathers of this step, one batch: its first rows of Y, X[i], the biases (lane 0 of wavefront 0)
        T *X = Xmat + (size_t)i * k;
        T yv[R][C];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const T *Y = Ymat + (size_t)max(rowid[m], 0) * k;
#pragma unroll
            for (int c = 0; c < C; ++c) yv[m][c] = Y[fc[c]];
        }
        const T xi_mine = X[min(tid, k - 1)];
        T mu_v = (T)0, bu_v = (T)0, bi_v = (T)0;
        if (p.use_bias) {           // (uniform addresses: every lane may load them)
            mu_v = p.asy_mu[0];
            bu_v = bu[u];
            bi_v = bi[i];
        }
        // ---- the NEXT step's sample (read-only stream)
        const long long tn = first + s + (s + 1 < count ? 1 : 0);
        const int u_n = p.su[tn], i_n = p.si[tn];
        const T rating_n = (T)p.sr[tn];
        // user vector: sum of the Y rows of the profile / sqrt(profile length)   (.pyx:424-441)
        T part[C];
#pragma unroll
        for (int c = 0; c < C; ++c) part[c] = (T)0;
#pragma unroll
        for (int m = 0; m < R; ++m)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                yv[m][c] = rowid[m] >= 0 && fok[c] ? yv[m][c] : (T)0;
                part[c] += yv[m][c];
            }
        for (int q = rs + wave + 16 * R; q < re; q += 16) {          // profiles longer than 64 rows
            const T *Y = Ymat + (size_t)p.indices[q] * k;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const T v = Y[fc[c]];
                part[c] += fok[c] ? v : (T)0;
            }
        }
        if (tid < k) s_xi[tid] = xi_mine;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (fok[c]) s_part[wave][lane + 64 * c] = part[c];
        // (the next step's CSR bounds: requested before the barrier, they arrive during the reduction)
        const int rs_n = p.indptr[u_n], re_n = p.indptr[u_n + 1];
        __syncthreads();
        if (tid < k) {
            T a = (T)0;
            for (int w = 0; w < 16; ++w) a += s_part[w][tid];
            s_acc[tid] = a / root((T)(re - rs));
        }
        __syncthreads();
        if (wave == 0) {
            T dot = (T)0;
            for (int f = lane; f < k; f += 64) dot += s_acc[f] * s_xi[f];
            dot = wave_sum(dot);
            if (lane == 0) {
                T pred = dot;
                if (p.use_bias) pred += mu_v + bu_v + bi_v;
                const T err = rating - pred;
                loss += (double)err * (double)err;
                const T pw1 = (T)(1.0 - b1p), pw2 = (T)(1.0 - b2p);
                if (p.use_bias) {       // global, item, user bias -- in that order (.pyx:458-490)
                    T g = adapt(p, err - p.bias_reg * mu_v, p.asy_c_mu, p.asy_c_mu + 1, 0, pw1, pw2);
                    p.asy_mu[0] = mu_v + p.lr * g;
                    g = adapt(p, err - p.bias_reg * bi_v, p.c1_bi, p.c2_bi, (size_t)i, pw1, pw2);
                    bi[i] = bi_v + p.lr * g;
                    g = adapt(p, err - p.bias_reg * bu_v, p.c1_bu, p.c2_bu, (size_t)u, pw1, pw2);
                    bu[u] = bu_v + p.lr * g;
                }
                s_err = err;
                s_pw1 = pw1;
                s_pw2 = pw2;
                if (p.sgd_mode == MI355REC_ADAM) {
                    b1p *= p.beta_1_d;
                    b2p *= p.beta_2_d;
                }
            }
        }
        // the next step's first row ids (its bounds have arrived)
        int rowid_n[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int q = rs_n + wave + 16 * m;
            const int id = p.indices[min(q, max(re_n - 1, rs_n))];
            rowid_n[m] = q < re_n ? id : -1;
        }
        __syncthreads();
        const T err = s_err, pw1 = s_pw1, pw2 = s_pw2;
        // every Y row of the profile moves against the OLD X[i]   (.pyx:493-511)
#pragma unroll
        for (int m = 0; m < R; ++m) {
            if (rowid[m] < 0) continue;
            const size_t row = (size_t)rowid[m];
            T *Y = Ymat + row * k;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (!fok[c]) continue;
                const int f = lane + 64 * c;
                const T w = yv[m][c];
                const T g = adapt(p, err * s_xi[f] - p.user_reg * w, p.c1U, p.c2U, row * k + f, pw1, pw2);
                Y[f] = w + p.lr * g;
            }
        }
        for (int q = rs + wave + 16 * R; q < re; q += 16) {
            const size_t row = (size_t)p.indices[q];
            T *Y = Ymat + row * k;
            for (int f = lane; f < k; f += 64) {
                const T w = Y[f];
                const T g = adapt(p, err * s_xi[f] - p.user_reg * w, p.c1U, p.c2U, row * k + f, pw1, pw2);
                Y[f] = w + p.lr * g;
            }
        }
        // X[i] moves against the user vector formed BEFORE the Y update   (.pyx:514-531)
        if (tid < k) {
            const T h = s_xi[tid];
            const T g = adapt(p, err * s_acc[tid] - p.item_reg * h, p.c1V, p.c2V, (size_t)i * k + tid, pw1, pw2);
            X[tid] = h + p.lr * g;
        }
        __threadfence_block();
        __syncthreads();
        u = u_n;
        i = i_n;
        rating = rating_n;
        rs = rs_n;
        re = re_n;
#pragma unroll
        for (int m = 0; m < R; ++m) rowid[m] = rowid_n[m];
    }
    if (tid == 0) {
        p.state->beta_1_power = b1p;
        p.state->beta_2_power = b2p;
        p.state->asy_loss = loss;
    }
}

}  // namespace
}  // namespace mi355rec

using namespace mi355rec;

struct mi355rec_mf {
    mi355rec_mf_config cfg{};
    int n_users = 0, n_items = 0, k = 0;
    int n_u_rows = 0;                 // rows of U: n_users, or n_items for AsySVD
    bool f64 = false;                 // storage / arithmetic type of factors and moments
    size_t nnz = 0;
    hipStream_t stream = nullptr;
    StreamTimer timer;
    DeviceBuffer<int> indptr, indices, su, si, sj;
    DeviceBuffer<float> data, sr, stage;
    // T-typed arrays (float or double by `f64`), held as bytes
    DeviceBuffer<unsigned char> U[2], V[2], bu[2], bi[2], c1U, c2U, c1V, c2V, c1_bu, c2_bu, c1_bi, c2_bi;
    DeviceBuffer<unsigned char> mu_state, mu_acc, asy_mu, asy_c_mu;
    DeviceBuffer<unsigned char> par;
    DeviceBuffer<double> loss_slots;
    DeviceBuffer<MfState> state;
    // schedule
    DeviceBuffer<unsigned long long> keys, keys_sorted;
    DeviceBuffer<int> slots, slots_sorted, head, head_scan, task_at, batch_count;
    DeviceBuffer<unsigned char> spar, cub_tmp;
    DeviceBuffer<TaskHeader> tasks;
    DeviceBuffer<unsigned> touched;         // fast schedule: (row, mini-batch) bitmap
    DeviceBuffer<int> sorted_slot, qtask, used;
    bool fast_schedule = false;
    DeviceBuffer<unsigned long long> ticks;
    DeviceBuffer<int4> recs, slot_recs;
    size_t cub_tmp_bytes = 0;
    size_t stream_capacity = 0;      // samples
    long long batch_capacity = 0;    // mini-batches the task arrays can hold
    long long batches_done = 0;      // batches executed since create (device state mirrors this)
    long long last_call_samples = 0; // samples in the stream buffer after the last native call
    mi355rec_stats stats{};
    DispatchTimers dispatch_timers;
    int max_timed = 0;
    // exact multi-GPU mode: this rank's share of every mini-batch and the exchange slabs
    DeviceBuffer<unsigned char> shard_send, shard_recv;
    int shard_rank = -1, shard_world = 0, shard_slots_per_rank = 0;
    long long shard_batches = 0;
    // one native epoch: sampler, schedule, n_batches mini-batch kernels -- in graphs of up to GRAPH_SEGMENT mini-batches each
    std::vector<hipGraphExec_t> epoch_graphs;
    size_t general_capacity = 0;            // samples the radix-sort schedule's buffers hold
    bool graph_failed = false;
    std::vector<double> host_loss;

    void drop_graphs() {
        for (hipGraphExec_t g : epoch_graphs) (void)hipGraphExecDestroy(g);
        epoch_graphs.clear();
    }
    ~mi355rec_mf() {
        if (stream) (void)hipStreamSynchronize(stream);
        drop_graphs();
        timer.destroy();
        dispatch_timers.destroy();
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int bits_for(unsigned long long n_values) {   // bits needed for values 0 .. n_values - 1 (at least 1)
    int b = 1;
    while (b < 63 && (1ull << b) < n_values) ++b;
    return b;
}

int per_sample(const mi355rec_mf *h) { return h->cfg.algorithm == MI355REC_MF_BPR ? 3 : 2; }

long long batches_per_epoch(const mi355rec_mf *h) {
    // .pyx:583 (BPR: n_users / B + 1) and :289 (FunkSVD: nnz / B + 1); ASY_SVD: nnz / 1 + 1 single-sample steps (.pyx:397)
    const long long B = h->cfg.batch_size;
    return (h->cfg.algorithm == MI355REC_MF_BPR ? (long long)h->n_users / B : (long long)h->nnz / B) + 1;
}

template <class T> T *as(const DeviceBuffer<unsigned char> &b) { return reinterpret_cast<T *>(b.ptr); }

bool fast_schedule_fits(const mi355rec_mf *h, long long n_batches);

template <class T>
void fill_params(mi355rec_mf *h, MfParams<T> &p) {
    const auto &c = h->cfg;
    p.n_users = h->n_users; p.n_items = h->n_items; p.k = h->k; p.batch_size = c.batch_size;
    p.use_bias = c.use_bias && c.algorithm != MI355REC_MF_BPR;
    p.sgd_mode = c.sgd_mode;
    p.sample_negatives = c.negative_interactions_quota != 0.0;
    p.tasks_per_batch = per_sample(h) * c.batch_size;
    p.lr = (T)c.learning_rate; p.user_reg = (T)c.user_reg; p.item_reg = (T)c.item_reg; p.bias_reg = (T)c.bias_reg;
    p.positive_reg = (T)c.positive_reg; p.negative_reg = (T)c.negative_reg;
    p.inv_batch = (T)1 / (T)c.batch_size;
    p.quota = (float)c.negative_interactions_quota;
    p.gamma = (T)c.gamma; p.beta_1 = (T)c.beta_1; p.beta_2 = (T)c.beta_2;
    p.one_m_gamma = (T)(1.0 - c.gamma); p.one_m_beta_1 = (T)(1.0 - c.beta_1); p.one_m_beta_2 = (T)(1.0 - c.beta_2);
    p.beta_1_d = c.beta_1; p.beta_2_d = c.beta_2;
    p.seed = c.random_seed;
    p.indptr = h->indptr.ptr; p.indices = h->indices.ptr; p.data = h->data.ptr;
    p.U0 = as<T>(h->U[0]); p.U1 = as<T>(h->U[1]); p.V0 = as<T>(h->V[0]); p.V1 = as<T>(h->V[1]);
    p.bu0 = as<T>(h->bu[0]); p.bu1 = as<T>(h->bu[1]); p.bi0 = as<T>(h->bi[0]); p.bi1 = as<T>(h->bi[1]);
    p.c1U = as<T>(h->c1U); p.c2U = as<T>(h->c2U); p.c1V = as<T>(h->c1V); p.c2V = as<T>(h->c2V);
    p.c1_bu = as<T>(h->c1_bu); p.c2_bu = as<T>(h->c2_bu); p.c1_bi = as<T>(h->c1_bi); p.c2_bi = as<T>(h->c2_bi);
    p.mu_state = reinterpret_cast<MuState<T> *>(h->mu_state.ptr);
    p.mu_acc = as<T>(h->mu_acc);
    p.asy_mu = as<T>(h->asy_mu); p.asy_c_mu = as<T>(h->asy_c_mu);
    p.par = h->par.ptr;
    p.loss_slots = h->loss_slots.ptr; p.state = h->state.ptr;
    p.su = h->su.ptr; p.si = h->si.ptr; p.sj = h->sj.ptr; p.sr = h->sr.ptr;
    p.samples_per_epoch = batches_per_epoch(h) * (long long)c.batch_size;
    p.tasks = h->tasks.ptr; p.recs = h->recs.ptr;
    p.slot_recs = h->slot_recs.ptr;
    p.slot_rec_stride = h->fast_schedule ? 3ll * per_sample(h) * c.batch_size : 0;      // (see ensure_stream_capacity)
    // (only the replica-batched launch sizes its grid below the slot count; it runs whole epochs, i.e. batches_per_epoch mini-batches)
    p.used = h->fast_schedule && fast_schedule_fits(h, batches_per_epoch(h)) ? h->used.ptr : nullptr;
    p.ticks = h->ticks.ptr;
    p.wg_base = 0;
    p.wg_stride = 1;
}

// ---- kernel selection -------------------------------------------------------------------------------------------------
// (Measured and rejected, round 4: rows of 17 .. 32 chunks on 16 lanes with two chunks each instead of 32 lanes with one -- four samples
// per wavefront, twice the bytes in flight per wavefront: the row gathers of a wavefront took 3 840 cycles instead of 2 350, one model
// 140 M samples/s instead of 176 M, the 32-model group 567 M instead of 678 M.)
template <int ALGO, class T, int VEC, int LPR, int KI>
void launch_batch_as(mi355rec_mf *h, const MfParams<T> &p, int grid, int batch_local, hipEvent_t e0, hipEvent_t e1) {
    if (p.sgd_mode == MI355REC_SGD) {
        if (e0) hipExtLaunchKernelGGL((mf_batch_kernel<ALGO, T, VEC, LPR, KI, true>), dim3(grid), dim3(256), 0, h->stream, e0, e1, 0, p, batch_local);
        else hipLaunchKernelGGL((mf_batch_kernel<ALGO, T, VEC, LPR, KI, true>), dim3(grid), dim3(256), 0, h->stream, p, batch_local);   // capturable
        return;
    }
    if (e0) hipExtLaunchKernelGGL((mf_batch_kernel<ALGO, T, VEC, LPR, KI, false>), dim3(grid), dim3(256), 0, h->stream, e0, e1, 0, p, batch_local);
    else hipLaunchKernelGGL((mf_batch_kernel<ALGO, T, VEC, LPR, KI, false>), dim3(grid), dim3(256), 0, h->stream, p, batch_local);
}

template <int ALGO, class T>
void launch_batch(mi355rec_mf *h, const MfParams<T> &p, int batch_local, bool timed, int wg_count = -1) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int grid = wg_count >= 0 ? wg_count : div_up(p.tasks_per_batch, 4);       // (p.wg_base names the first one)
    if (grid == 0) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) h->dispatch_timers.next(e0, e1, h->max_timed);
    const int k = h->k, chunks = k / VEC;
    if (k % VEC == 0 && chunks <= 16) launch_batch_as<ALGO, T, VEC, 16, 1>(h, p, grid, batch_local, e0, e1);
    else if (k % VEC == 0 && chunks <= 32) launch_batch_as<ALGO, T, VEC, 32, 1>(h, p, grid, batch_local, e0, e1);
    else if (k % VEC == 0 && chunks <= 64) launch_batch_as<ALGO, T, VEC, 64, 1>(h, p, grid, batch_local, e0, e1);
    else if (k % VEC == 0 && chunks <= 128) launch_batch_as<ALGO, T, VEC, 64, 2>(h, p, grid, batch_local, e0, e1);
    else if (e0) hipExtLaunchKernelGGL((mf_batch_generic_kernel<ALGO, T>), dim3(grid), dim3(256), 0, h->stream, e0, e1, 0, p, batch_local);
    else hipLaunchKernelGGL((mf_batch_generic_kernel<ALGO, T>), dim3(grid), dim3(256), 0, h->stream, p, batch_local);
}

// instance of the mini-batch kernel a model runs on (must mirror launch_batch); -1: the any-k kernel
int kernel_class(const mi355rec_mf *h) {
    const int vec = h->f64 ? 2 : 4, k = h->k;
    if (k % vec != 0) return -1;
    const int chunks = k / vec;
    return chunks <= 16 ? 0 : (chunks <= 32 ? 1 : (chunks <= 64 ? 2 : (chunks <= 128 ? 3 : -1)));
}

template <int ALGO, class T, int VEC, int LPR, int KI>
void launch_group_as(hipStream_t s, const MfParams<T> *table, dim3 grid, int batch_local, bool plain_sgd, hipEvent_t e0, hipEvent_t e1) {
    // (a 64-VGPR build for 8 wavefronts per SIMD was measured twice and is gone: 20 spilled registers, 422 M against 678 M samples/s)
    if (plain_sgd) {
        if (e0) hipExtLaunchKernelGGL((mf_group_batch_kernel<ALGO, T, VEC, LPR, KI, true>), grid, dim3(256), 0, s, e0, e1, 0, table, batch_local);
        else hipLaunchKernelGGL((mf_group_batch_kernel<ALGO, T, VEC, LPR, KI, true>), grid, dim3(256), 0, s, table, batch_local);
        return;
    }
    if (e0) hipExtLaunchKernelGGL((mf_group_batch_kernel<ALGO, T, VEC, LPR, KI, false>), grid, dim3(256), 0, s, e0, e1, 0, table, batch_local);
    else hipLaunchKernelGGL((mf_group_batch_kernel<ALGO, T, VEC, LPR, KI, false>), grid, dim3(256), 0, s, table, batch_local);
}

template <int ALGO, class T>
void launch_group_batch(hipStream_t s, const MfParams<T> *table, int klass, int wgs, int n_models, int batch_local, bool plain_sgd,
                        hipEvent_t e0, hipEvent_t e1) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const dim3 grid(wgs, n_models);
    switch (klass) {
        case 0: launch_group_as<ALGO, T, VEC, 16, 1>(s, table, grid, batch_local, plain_sgd, e0, e1); break;
        case 1: launch_group_as<ALGO, T, VEC, 32, 1>(s, table, grid, batch_local, plain_sgd, e0, e1); break;
        case 2: launch_group_as<ALGO, T, VEC, 64, 1>(s, table, grid, batch_local, plain_sgd, e0, e1); break;
        default: launch_group_as<ALGO, T, VEC, 64, 2>(s, table, grid, batch_local, plain_sgd, e0, e1); break;
    }
}

template <class T>
void launch_sampler(mi355rec_mf *h, const MfParams<T> &p, hipStream_t on = nullptr) {
    const int grid = div_up(p.samples_per_epoch, 256);
    hipStream_t s = on ? on : h->stream;
    if (h->cfg.algorithm == MI355REC_MF_BPR) hipLaunchKernelGGL((mf_sample_kernel<MI355REC_MF_BPR, T>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((mf_sample_kernel<MI355REC_MF_FUNK_SVD, T>), dim3(grid), dim3(256), 0, s, p);
    hipLaunchKernelGGL(mf_epoch_advance_kernel, dim3(1), dim3(64), 0, s, p.state);
}

int pow2_at_least(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// samples of a task's list a wavefront of the mini-batch kernel works on at a time (must mirror launch_batch)
int samples_in_flight(const mi355rec_mf *h) {
    const int vec = h->f64 ? 2 : 4, k = h->k;
    if (k % vec != 0 || k / vec > 128) return 1;
    const int chunks = k / vec;
    return chunks <= 16 ? 4 : (chunks <= 32 ? 2 : 1);
}

bool fast_schedule_fits(const mi355rec_mf *h, long long n_batches) {
    const int tpb = per_sample(h) * h->cfg.batch_size;
    return !getenv("MI355REC_MF_GENERAL_SCHEDULE") && n_batches <= FAST_MAX_BATCHES && tpb <= FAST_MAX_SLOTS &&
           bits_for((unsigned long long)h->n_users + h->n_items) + bits_for((unsigned long long)std::max(tpb, SCHED_THREADS)) <= 31 &&
           (h->k % (h->f64 ? 2 : 4) == 0 && h->k / (h->f64 ? 2 : 4) <= 128);    // the any-k kernel does not split wide lists
}

// keys, run starts, header slots (words) + one byte per incidence for the once-touched flags
size_t sched_mid_bytes(int np) {
    const size_t middle = std::max(sizeof(unsigned) * (2 * (size_t)np + 1), sched_sort_storage_bytes(np));      // run starts + header slots | sort scratch
    return (middle + 15) & ~(size_t)15;
}
size_t sched_lds_bytes(int np) { return sizeof(unsigned) * (size_t)np + sched_mid_bytes(np) + (size_t)np + 16; }

void set_sched_sort_attribute(const void *kernel, bool (&attr_set)[64]) {
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));                 // the attribute is per DEVICE, not per process
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        MI_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sched_lds_bytes(FAST_MAX_SLOTS)));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
}

FastSchedParams fast_sched_params(mi355rec_mf *h, long long n_samples) {
    FastSchedParams f;
    memset(&f, 0, sizeof(f));                   // (padding bytes included: the group compares tables bytewise)
    f.n_samples = n_samples;
    f.per = per_sample(h);
    f.n_users = h->n_users;
    f.n_entries = h->n_users + h->n_items;
    f.batch_size = h->cfg.batch_size;
    f.tasks_per_batch = f.per * h->cfg.batch_size;
    f.np = std::max(SCHED_THREADS, pow2_at_least(f.tasks_per_batch));
    f.slot_bits = bits_for((unsigned long long)f.np);
    f.entry_bits = std::min(32 - f.slot_bits, bits_for((unsigned long long)f.n_entries) + 1);
    f.mid_bytes = (int)sched_mid_bytes(f.np);
    f.words = FAST_MAX_BATCHES / 32;
    f.group = samples_in_flight(h);
    f.su = h->su.ptr; f.si = h->si.ptr; f.sj = h->sj.ptr; f.sr = h->sr.ptr;
    f.touched = h->touched.ptr; f.par = h->par.ptr;
    f.sorted_slot = h->sorted_slot.ptr; f.qtask = h->qtask.ptr; f.used = h->used.ptr;
    f.tasks = h->tasks.ptr; f.recs = h->recs.ptr; f.slot_recs = h->slot_recs.ptr;
    // fused sample tasks: BPR only; not in the exact multi-GPU mode, whose exchange slabs hold one row per task slot
    f.fuse = h->cfg.algorithm == MI355REC_MF_BPR && h->shard_rank < 0 && !getenv("MI355REC_MF_NO_FUSE");
    return f;
}

void enqueue_fast_schedule(mi355rec_mf *h, long long n_samples, long long n_batches, const FastSchedParams *given = nullptr,
                           hipStream_t on = nullptr) {
    hipStream_t s = on ? on : h->stream;
    const FastSchedParams f = given ? *given : fast_sched_params(h, n_samples);
    const size_t lds = sched_lds_bytes(f.np);
    static bool attr_set[64] = {};
    set_sched_sort_attribute(reinterpret_cast<const void *>(mf_sched_sort_kernel), attr_set);
    hipLaunchKernelGGL(mf_sched_sort_kernel, dim3((unsigned)n_batches), dim3(SCHED_THREADS), lds, s, f);
    hipLaunchKernelGGL(mf_sched_emit_kernel, dim3(div_up(f.tasks_per_batch, 256), (unsigned)n_batches), dim3(256), 0, s, f);
    hipLaunchKernelGGL(mf_sched_finish_kernel, dim3(div_up(f.n_entries, 256)), dim3(256), 0, s, f);
}

// Stream buffers -> tasks (all on the handle's stream, no host synchronisation: capturable).
void enqueue_schedule(mi355rec_mf *h, long long n_samples, long long n_batches) {
    if (h->fast_schedule && fast_schedule_fits(h, n_batches)) {
        enqueue_fast_schedule(h, n_samples, n_batches);
        return;
    }
    hipStream_t s = h->stream;
    const int per = per_sample(h);
    const long long n = n_samples * per;
    SchedParams sp{};
    sp.n_samples = n_samples;
    sp.per = per;
    sp.n_users = h->n_users;
    sp.batch_size = h->cfg.batch_size;
    sp.batch_bits = bits_for((unsigned long long)n_batches);
    sp.tasks_per_batch = per * h->cfg.batch_size;
    sp.su = h->su.ptr; sp.si = h->si.ptr; sp.sj = h->sj.ptr; sp.sr = h->sr.ptr;
    sp.keys = h->keys.ptr; sp.slots = h->slots.ptr;
    sp.keys_sorted = h->keys_sorted.ptr; sp.slots_sorted = h->slots_sorted.ptr;
    sp.head = h->head.ptr; sp.head_scan = h->head_scan.ptr; sp.task_at = h->task_at.ptr;
    sp.spar = h->spar.ptr; sp.par = h->par.ptr; // placement of the headers inside a batch: by rank (reproducible layout) for the exact multi-GPU mode and for batches too large
    // for the in-LDS schedule; by counter for streams of many small batches (see mf_tasks_kernel)
    const bool by_rank = h->shard_rank >= 0 || sp.tasks_per_batch > FAST_MAX_SLOTS;
    sp.batch_count = by_rank ? nullptr : h->batch_count.ptr;
    // (the unsorted keys are dead after the sort: their buffer holds the first-incidence flags and their scan, n ints each)
    sp.slot_flag = reinterpret_cast<int *>(h->keys.ptr);
    sp.slot_rank = reinterpret_cast<int *>(h->keys.ptr) + n;
    sp.tasks = h->tasks.ptr; sp.recs = h->recs.ptr;
    const int end_bit = sp.batch_bits + bits_for((unsigned long long)h->n_users + h->n_items);
    MI_REQUIRE(end_bit <= 64, "sample stream too long for the schedule keys");
    MI_HIP(hipMemsetAsync(h->batch_count.ptr, 0, sizeof(int) * (size_t)n_batches, s));
    MI_HIP(hipMemsetAsync(h->tasks.ptr, 0, sizeof(TaskHeader) * (size_t)n_batches * sp.tasks_per_batch, s));
    hipLaunchKernelGGL(mf_keys_kernel, dim3(div_up(n_samples, 256)), dim3(256), 0, s, sp);
    size_t bytes = h->cub_tmp_bytes;
    MI_HIP(rocprim::radix_sort_pairs(h->cub_tmp.ptr, bytes, h->keys.ptr, h->keys_sorted.ptr, h->slots.ptr,
                                              h->slots_sorted.ptr, (int)n, 0, end_bit, s));
    MI_HIP(hipMemsetAsync(sp.slot_flag, 0, sizeof(int) * (size_t)n, s));
    hipLaunchKernelGGL(mf_heads_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, sp);
    bytes = h->cub_tmp_bytes;
    MI_HIP(rocprim::inclusive_scan(h->cub_tmp.ptr, bytes, h->head.ptr, h->head_scan.ptr, (size_t)n, rocprim::plus<int>(), s));
    bytes = h->cub_tmp_bytes;
    MI_HIP(rocprim::exclusive_scan(h->cub_tmp.ptr, bytes, sp.slot_flag, reinterpret_cast<int *>(h->keys.ptr) + n, 0, (size_t)n, rocprim::plus<int>(), s));
    hipLaunchKernelGGL(mf_tasks_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, sp);
    hipLaunchKernelGGL(mf_recs_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, sp);
}

// Mini-batches per schedule.  A stream the in-LDS schedule cannot hold at once (more than FAST_MAX_BATCHES mini-batches: FunkSVD's
// epoch at the ML-20M shape has 20 001) is scheduled and run FAST_MAX_BATCHES mini-batches at a time, each part a stream of its
// own (its end moves the global batch index and the global-bias ring on): three schedule launches per 256 mini-batch launches,
// against the radix-sort schedule of the whole stream they keep the split of long lists over a workgroup (a list of 10
// held the kernel for 14 000 cycles where the median wavefront took 7 300), headers and records in 16 MB that stay cached instead of
// 1.9 GB, and 1.5 GB of sort buffers are never allocated.  Whole stream: when it fits, when the in-LDS schedule is not available
// for this handle, and in the exact multi-GPU mode (its exchange walks one schedule).
long long schedule_span(const mi355rec_mf *h, long long n_batches) {
    if (h->shard_rank < 0 && n_batches > FAST_MAX_BATCHES && fast_schedule_fits(h, 1) && !getenv("MI355REC_MF_WHOLE_STREAM_SCHEDULE"))
        return FAST_MAX_BATCHES;
    return n_batches;
}

// mini-batches first .. last - 1 of a stream of n_batches, with the schedules and stream ends that fall into that range
template <class T>
void enqueue_stream(mi355rec_mf *h, const MfParams<T> &p, long long n_samples, long long n_batches, bool timed, long long first = 0,
                    long long last = -1) {
    const bool bpr = h->cfg.algorithm == MI355REC_MF_BPR;
    const long long span = schedule_span(h, n_batches), B = h->cfg.batch_size;
    if (last < 0) last = n_batches;
    for (long long b = first; b < last; ++b) {
        const long long part = b / span * span, local = b - part, part_batches = std::min(span, n_batches - part);
        if (local == 0) {
            if (span == n_batches) {
                enqueue_schedule(h, n_samples, n_batches);
            } else {
                FastSchedParams f = fast_sched_params(h, std::min(n_samples - part * B, part_batches * B));
                f.su += part * B; f.si += part * B; f.sj += part * B; f.sr += part * B;
                enqueue_fast_schedule(h, f.n_samples, part_batches, &f);
            }
        }
        if (bpr) launch_batch<MI355REC_MF_BPR, T>(h, p, (int)local, timed);
        else launch_batch<MI355REC_MF_FUNK_SVD, T>(h, p, (int)local, timed);
        if (local + 1 == part_batches) hipLaunchKernelGGL(mf_stream_end_kernel<T>, dim3(1), dim3(64), 0, h->stream, p, part_batches);
    }
}

constexpr int ASY_CHUNK = 1 << 16;   // steps per launch of the ordered AsySVD kernel (keeps single launches short)

template <class T>
void enqueue_asy_steps(mi355rec_mf *h, const MfParams<T> &p, long long n_steps, bool timed) {
    for (long long first = 0; first < n_steps; first += ASY_CHUNK) {
        const int count = (int)std::min<long long>(ASY_CHUNK, n_steps - first);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) h->dispatch_timers.next(e0, e1, h->max_timed);
        auto go = [&](auto kernel) {
            if (e0) hipExtLaunchKernelGGL(kernel, dim3(1), dim3(1024), 0, h->stream, e0, e1, 0, p, first, count);
            else hipLaunchKernelGGL(kernel, dim3(1), dim3(1024), 0, h->stream, p, first, count);
        };
        if (h->k <= 64) go(mf_asy_kernel<T, 1>);
        else if (h->k <= 128) go(mf_asy_kernel<T, 2>);
        else go(mf_asy_kernel<T, 4>);
    }
}

// One native epoch as plain launches (the first `max_timed` mini-batch launches carry per-dispatch events when `timed`) -- or the
// part of it that holds mini-batches first .. last - 1: sampler and schedule go with the first part.
template <class T>
void enqueue_epoch(mi355rec_mf *h, const MfParams<T> &p, bool timed, long long first = 0, long long last = -1) {
    if (first == 0) launch_sampler(h, p);
    if (h->cfg.algorithm == MI355REC_MF_ASY_SVD) {
        enqueue_asy_steps(h, p, p.samples_per_epoch, timed);
        return;
    }
    enqueue_stream(h, p, p.samples_per_epoch, batches_per_epoch(h), timed, first, last);
}

// Capture one epoch into graphs of up to GRAPH_SEGMENT mini-batches (once per handle; re-captured only if the stream buffers are
// re-allocated).  FunkSVD's epoch at the ML-20M shape is 20 001 mini-batches: as plain launches the host's launch rate (8.8 us
// per mini-batch) was the bound, not the chain of kernels.
constexpr long long GRAPH_SEGMENT = 4096, MAX_GRAPH_BATCHES = 64 * GRAPH_SEGMENT;
template <class T>
void ensure_epoch_graph(mi355rec_mf *h, const MfParams<T> &p) {
    if (!h->epoch_graphs.empty() || h->graph_failed) return;
    const long long nb = h->cfg.algorithm == MI355REC_MF_ASY_SVD ? 1 : batches_per_epoch(h);
    for (long long first = 0; first < nb; first += GRAPH_SEGMENT) {
        hipGraph_t g = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            try {
                enqueue_epoch(h, p, false, first, std::min(nb, first + GRAPH_SEGMENT));
            } catch (...) {
                (void)hipStreamEndCapture(h->stream, &g);
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                h->drop_graphs();
                h->graph_failed = true;
                return;
            }
            e = hipStreamEndCapture(h->stream, &g);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (e != hipSuccess) {       // plain launches remain correct, only slower: remember and go on
            (void)hipGetLastError();
            h->drop_graphs();
            h->graph_failed = true;
            return;
        }
        h->epoch_graphs.push_back(exec);
    }
}

// `whole_stream`: the caller schedules the stream in one piece whatever its length (group members, exact multi-GPU mode)
void ensure_stream_capacity(mi355rec_mf *h, size_t n_samples, long long n_batches, bool whole_stream = false) {
    const bool asy = h->cfg.algorithm == MI355REC_MF_ASY_SVD;
    if (h->stream_capacity < n_samples) {
        h->drop_graphs();       // they hold the old buffer addresses
        h->su.alloc(n_samples);
        h->si.alloc(n_samples);
        h->sj.alloc(n_samples);
        h->sr.alloc(n_samples);
        h->stream_capacity = n_samples;
    }
    if (asy) return;
    const bool fast1 = fast_schedule_fits(h, 1);
    // the radix-sort schedule's buffers: only for streams it will actually see
    whole_stream = whole_stream || getenv("MI355REC_MF_WHOLE_STREAM_SCHEDULE");
    const bool general = !fast1 || (whole_stream && !fast_schedule_fits(h, n_batches));
    if (general && h->general_capacity < n_samples) {
        h->drop_graphs();
        const size_t n = n_samples * (size_t)per_sample(h);
        MI_REQUIRE(n < (1ull << 31), "sample stream too long (%zu incidences)", n);
        h->keys.alloc(n); h->keys_sorted.alloc(n);
        h->slots.alloc(n); h->slots_sorted.alloc(n);
        h->head.alloc(n); h->head_scan.alloc(n); h->task_at.alloc(n);
        h->spar.alloc(n);
        size_t sort_bytes = 0, scan_bytes = 0;
        MI_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, h->keys.ptr, h->keys_sorted.ptr, h->slots.ptr,
                                                  h->slots_sorted.ptr, (int)n, 0, 64, h->stream));
        MI_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, h->head.ptr, h->head_scan.ptr, (size_t)n, rocprim::plus<int>(), h->stream));
        h->cub_tmp_bytes = std::max(sort_bytes, scan_bytes) + 256;
        h->cub_tmp.alloc(h->cub_tmp_bytes);
        h->general_capacity = n_samples;
    }
    // headers and records: per mini-batch of a SCHEDULE (a long stream on the in-LDS schedule reuses FAST_MAX_BATCHES of them)
    const long long table_batches = general ? n_batches : std::min<long long>(n_batches, FAST_MAX_BATCHES);
    if (h->batch_capacity < table_batches) {
        h->drop_graphs();
        const size_t tpb = (size_t)per_sample(h) * h->cfg.batch_size;
        h->batch_count.alloc((size_t)table_batches);
        h->tasks.alloc((size_t)(table_batches + 1) * tpb);
        MI_HIP(hipMemsetAsync(h->tasks.ptr, 0, sizeof(TaskHeader) * (size_t)(table_batches + 1) * tpb, h->stream));
        h->recs.alloc((size_t)table_batches * tpb);          // records of batch b start at b * tpb in both schedule paths
        h->fast_schedule = fast1;
        // (the mini-batch kernel reads a slot's pair records whatever the schedule: without the in-LDS schedule one mini-batch of zeros)
        h->slot_recs.alloc_zero(3 * tpb * (size_t)(fast1 ? table_batches : 1), h->stream);
        if (h->fast_schedule) {
            h->sorted_slot.alloc((size_t)table_batches * tpb);
            h->qtask.alloc((size_t)table_batches * tpb);
            h->used.alloc((size_t)table_batches);
            if (!h->touched.ptr) h->touched.alloc_zero(((size_t)h->n_users + h->n_items) * (FAST_MAX_BATCHES / 32), h->stream);
        }
        h->batch_capacity = table_batches;
    }
}

double bytes_per_sample(const mi355rec_mf *h) {
    // ALGORITHMIC lower bound of DESIGN.md section 4: every row a sample touches is read once and written once
    // (3 rows for BPR, 2 for FunkSVD), fp32.
    const double rows = h->cfg.algorithm == MI355REC_MF_BPR ? 3.0 : 2.0;   // (AsySVD: its profile-sized term is added per call)
    return rows * 2.0 * 4.0 * (double)h->k;
}

void begin_call(mi355rec_mf *h) {
    MI_HIP(hipMemsetAsync(h->loss_slots.ptr, 0, sizeof(double) * h->loss_slots.count, h->stream));
    MI_HIP(hipMemsetAsync(&h->state.ptr->asy_loss, 0, sizeof(double), h->stream));
    h->dispatch_timers.reset();
}

void finish_call(mi355rec_mf *h, long long n_samples, long long n_launches) {
    MI_HIP(hipGetLastError());
    h->host_loss.resize(h->loss_slots.count);
    h->loss_slots.download(h->host_loss.data(), h->loss_slots.count, h->stream);
    MI_HIP(hipStreamSynchronize(h->stream));
    double loss = 0;
    for (double v : h->host_loss) loss += v;
    if (h->cfg.algorithm == MI355REC_MF_ASY_SVD) {
        MfState st{};
        MI_HIP(hipMemcpy(&st, h->state.ptr, sizeof(MfState), hipMemcpyDeviceToHost));
        loss = st.asy_loss;
    }
    h->stats.call_ms = h->timer.elapsed_ms();
    h->stats.kernel_ms = h->dispatch_timers.total_ms();
    h->stats.n_timed = h->dispatch_timers.used;
    h->stats.n_launches = n_launches;           // launches of the dominant (mini-batch) kernel
    h->stats.n_units = n_samples;
    h->stats.algorithmic_bytes = bytes_per_sample(h) * (double)n_samples;
    h->stats.algorithmic_flops = 0;
    h->stats.loss = loss;
}

template <class T>
void create_typed(mi355rec_mf *h, const void *U0, const void *V0) {
    hipStream_t s = h->stream;
    const auto &cfg = h->cfg;
    const size_t nu = (size_t)h->n_u_rows * h->k, ni = (size_t)h->n_items * h->k, ts = sizeof(T);
    for (int b = 0; b < 2; ++b) {
        h->U[b].upload(static_cast<const unsigned char *>(U0), nu * ts, s);
        h->V[b].upload(static_cast<const unsigned char *>(V0), ni * ts, s);
        h->bu[b].alloc_zero((size_t)h->n_users * ts, s);
        h->bi[b].alloc_zero((size_t)h->n_items * ts, s);
    }
    if (cfg.sgd_mode != MI355REC_SGD) {
        h->c1U.alloc_zero(nu * ts, s);
        h->c1V.alloc_zero(ni * ts, s);
        h->c1_bu.alloc_zero((size_t)h->n_users * ts, s);
        h->c1_bi.alloc_zero((size_t)h->n_items * ts, s);
        if (cfg.sgd_mode == MI355REC_ADAM) {
            h->c2U.alloc_zero(nu * ts, s);
            h->c2V.alloc_zero(ni * ts, s);
            h->c2_bu.alloc_zero((size_t)h->n_users * ts, s);
            h->c2_bi.alloc_zero((size_t)h->n_items * ts, s);
        }
    }
    h->mu_state.alloc_zero(3 * sizeof(MuState<T>), s);
    h->mu_acc.alloc_zero(3 * MU_SLOTS * ts, s);
    h->asy_mu.alloc_zero(ts, s);
    h->asy_c_mu.alloc_zero(2 * ts, s);
}

template <class T>
void run_epochs_typed(mi355rec_mf *h, int n_epochs) {
    const long long B = h->cfg.batch_size;
    const long long per_epoch = batches_per_epoch(h);
    const bool asy = h->cfg.algorithm == MI355REC_MF_ASY_SVD;
    ensure_stream_capacity(h, (size_t)(per_epoch * B), per_epoch);
    MfParams<T> p{};
    fill_params(h, p);
    begin_call(h);
    // epochs whose mini-batch launches carry timing events run as plain launches, the rest replays the graph
    const long long timed_epochs = h->max_timed > 0 ? std::min<long long>(n_epochs, (h->max_timed + per_epoch - 1) / per_epoch) : 0;
    // MI355REC_NO_GRAPH=1: plain launches only (rocprofv3 on ROCm 7.2 crashes while tracing graph replays)
    bool use_graph = per_epoch <= MAX_GRAPH_BATCHES && n_epochs - timed_epochs > 0 && !getenv("MI355REC_NO_GRAPH") && !asy;
    if (use_graph) {
        ensure_epoch_graph(h, p);
        use_graph = !h->epoch_graphs.empty();
    }
    h->timer.start(h->stream);
    for (long long e = 0; e < n_epochs; ++e) {
        if (e < timed_epochs || !use_graph) enqueue_epoch(h, p, e < timed_epochs);
        else for (hipGraphExec_t g : h->epoch_graphs) MI_HIP(hipGraphLaunch(g, h->stream));
    }
    h->timer.stop(h->stream);
    h->batches_done += per_epoch * n_epochs;
    h->last_call_samples = n_epochs > 0 ? per_epoch * B : 0;
    finish_call(h, per_epoch * n_epochs * B, asy ? n_epochs * ((per_epoch + ASY_CHUNK - 1) / ASY_CHUNK) : per_epoch * n_epochs);
}

template <class T>
void run_samples_typed(mi355rec_mf *h, int64_t n) {
    const long long B = h->cfg.batch_size;
    MfParams<T> p{};
    fill_params(h, p);
    const long long n_batches = (n + B - 1) / B;
    begin_call(h);
    h->timer.start(h->stream);
    if (h->cfg.algorithm == MI355REC_MF_ASY_SVD) {
        enqueue_asy_steps(h, p, n, true);
        h->timer.stop(h->stream);
        h->batches_done += n;
        finish_call(h, n, (n + ASY_CHUNK - 1) / ASY_CHUNK);
        return;
    }
    enqueue_stream(h, p, n, n_batches, true);
    h->timer.stop(h->stream);
    h->batches_done += n_batches;
    finish_call(h, n, n_batches);
}

// O = float (the float32 matrices north_star speaks of) or double (what the reference's getters return, .pyx:685-702: exact when
// the device state is float64)
template <class T, class O>
void get_factors_typed(mi355rec_mf *h, O *U, O *V, O *bu, O *bi, O *mu) {
    hipStream_t s = h->stream;
    MfParams<T> p{};
    fill_params(h, p);
    const size_t nu = (size_t)h->n_u_rows * h->k, ni = (size_t)h->n_items * h->k;
    const size_t need = std::max(std::max(nu, ni), (size_t)std::max(h->n_users, h->n_items)) * (sizeof(O) / sizeof(float));
    if (h->stage.count < need) h->stage.alloc(need);
    O *stage = reinterpret_cast<O *>(h->stage.ptr);
    // rows of U are entries [0, n_u_rows) of `par` for BPR / FunkSVD; AsySVD never flips a buffer (par stays 0)
    const unsigned char *par_u = h->par.ptr, *par_v = h->par.ptr + h->n_users;
    auto gather = [&](const T *b0, const T *b1, const unsigned char *par, size_t rows, int k, O *host) {
        hipLaunchKernelGGL((mf_gather_rows_kernel<T, O>), dim3(div_up((long long)rows * k, 256)), dim3(256), 0, s, b0, b1, par,
                           (long long)rows, k, stage);
        MI_HIP(hipMemcpyAsync(host, stage, rows * k * sizeof(O), hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
    };
    const bool asy = h->cfg.algorithm == MI355REC_MF_ASY_SVD;
    if (U) gather(p.U0, p.U1, asy ? par_v : par_u, h->n_u_rows, h->k, U);   // AsySVD: item-sized, par is all zero anyway
    if (V) gather(p.V0, p.V1, par_v, h->n_items, h->k, V);
    if (bu) gather(p.bu0, p.bu1, par_u, h->n_users, 1, bu);
    if (bi) gather(p.bi0, p.bi1, par_v, h->n_items, 1, bi);
    if (mu) {
        if (asy) {
            T v;
            MI_HIP(hipMemcpyAsync(&v, p.asy_mu, sizeof(T), hipMemcpyDeviceToHost, s));
            MI_HIP(hipStreamSynchronize(s));
            *mu = (O)v;
        } else {
            hipLaunchKernelGGL((mf_final_mu_kernel<T, O>), dim3(1), dim3(64), 0, s, p, stage);
            MI_HIP(hipMemcpyAsync(mu, stage, sizeof(O), hipMemcpyDeviceToHost, s));
            MI_HIP(hipStreamSynchronize(s));
        }
    }
}

}  // namespace

extern "C" int mi355rec_mf_create(mi355rec_mf_t *out, const mi355rec_mf_config *cfg, int32_t n_users, int32_t n_items,
                                  const int32_t *indptr, const int32_t *indices, const float *data, const void *U0,
                                  const void *V0) {
    return guarded([&] {
        MI_REQUIRE(out && cfg && indptr && indices && data && U0 && V0, "NULL argument");
        MI_REQUIRE(n_users > 0 && n_items > 0, "empty URM");
        MI_REQUIRE(cfg->algorithm >= MI355REC_MF_BPR && cfg->algorithm <= MI355REC_MF_ASY_SVD,
                   "Value for 'algorithm_name' not recognized (%d)", cfg->algorithm);
        const bool asy = cfg->algorithm == MI355REC_MF_ASY_SVD;
        MI_REQUIRE(!asy || cfg->batch_size == 1, "Batch size other than 1 not supported for ASY_SVD");
        if (asy && cfg->n_factors > ASY_KMAX)
            fail(MI355REC_E_UNSUPPORTED, "ASY_SVD: n_factors = %d exceeds %d", cfg->n_factors, ASY_KMAX);
        MI_REQUIRE(cfg->sgd_mode >= MI355REC_SGD && cfg->sgd_mode <= MI355REC_ADAM, "Value for 'sgd_mode' not recognized (%d)",
                   cfg->sgd_mode);
        MI_REQUIRE(cfg->n_factors >= 1, "n_factors must be >= 1");
        if (cfg->n_factors > 512) fail(MI355REC_E_UNSUPPORTED, "n_factors = %d exceeds 512", cfg->n_factors);
        MI_REQUIRE(cfg->batch_size >= 1, "batch_size must be >= 1");
        MI_REQUIRE(cfg->precision == MI355REC_F32 || cfg->precision == MI355REC_F64, "precision must be MI355REC_F32 or MI355REC_F64");
        ensure_device();
        std::unique_ptr<mi355rec_mf> h(new mi355rec_mf());
        h->cfg = *cfg;
        h->n_users = n_users;
        h->n_items = n_items;
        h->k = cfg->n_factors;
        h->n_u_rows = asy ? n_items : n_users;
        h->f64 = cfg->precision == MI355REC_F64;
        h->nnz = (size_t)indptr[n_users];
        MI_REQUIRE(h->nnz > 0, "URM has no interactions");
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->timer.init();
        hipStream_t s = h->stream;
        h->indptr.upload(indptr, (size_t)n_users + 1, s);
        h->indices.upload(indices, h->nnz, s);
        h->data.upload(data, h->nnz, s);
        if (h->f64) create_typed<double>(h.get(), U0, V0); else create_typed<float>(h.get(), U0, V0);
        h->par.alloc_zero((size_t)n_users + n_items, s);
        if (getenv("MI355REC_MF_TICKS")) h->ticks.alloc_zero((size_t)per_sample(h.get()) * cfg->batch_size * 8, s);
        h->loss_slots.alloc_zero((size_t)per_sample(h.get()) * cfg->batch_size * 4, s);
        h->state.alloc_zero(1, s);
        {   // Adam's running beta powers start at beta^1 (.pyx:217-218)
            MfState init{};
            init.beta_1_power = cfg->beta_1;
            init.beta_2_power = cfg->beta_2;
            MI_HIP(hipMemcpyAsync(h->state.ptr, &init, sizeof(MfState), hipMemcpyHostToDevice, s));
        }
        MI_HIP(hipStreamSynchronize(s));
        *out = h.release();
    });
}

extern "C" int mi355rec_mf_run_epochs(mi355rec_mf_t h, int32_t n_epochs) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        MI_REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        ensure_device();
        if (h->f64) run_epochs_typed<double>(h, n_epochs); else run_epochs_typed<float>(h, n_epochs);
    });
}

extern "C" int mi355rec_mf_run_samples(mi355rec_mf_t h, const int32_t *u, const int32_t *i, const int32_t *j,
                                       const float *rating, int64_t n) {
    return guarded([&] {
        MI_REQUIRE(h && u && i, "NULL argument");
        const bool bpr = h->cfg.algorithm == MI355REC_MF_BPR;
        MI_REQUIRE(bpr ? j != nullptr : rating != nullptr, "%s", bpr ? "BPR replay needs the negative items" : "FunkSVD / AsySVD replay needs the ratings");
        MI_REQUIRE(n >= 0, "n must be >= 0");
        ensure_device();
        if (n == 0) return;
        const long long B = h->cfg.batch_size;
        const long long per_epoch = batches_per_epoch(h);
        ensure_stream_capacity(h, (size_t)std::max<long long>(n, per_epoch * B), std::max<long long>((n + B - 1) / B, per_epoch));
        hipStream_t s = h->stream;
        MI_HIP(hipMemcpyAsync(h->su.ptr, u, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(h->si.ptr, i, sizeof(int) * n, hipMemcpyHostToDevice, s));
        if (bpr) MI_HIP(hipMemcpyAsync(h->sj.ptr, j, sizeof(int) * n, hipMemcpyHostToDevice, s));
        else MI_HIP(hipMemcpyAsync(h->sr.ptr, rating, sizeof(float) * n, hipMemcpyHostToDevice, s));
        h->last_call_samples = 0;
        if (h->f64) run_samples_typed<double>(h, n); else run_samples_typed<float>(h, n);
    });
}

// ---- exact multi-GPU mini-batches ---------------------------------------------------------------------------------------
namespace {

template <class T>
void shard_begin_typed(mi355rec_mf *h) {
    const long long B = h->cfg.batch_size, per_epoch = batches_per_epoch(h);
    ensure_stream_capacity(h, (size_t)(per_epoch * B), per_epoch, true);
    MfParams<T> p{};
    fill_params(h, p);
    begin_call(h);
    h->timer.start(h->stream);
    launch_sampler(h, p);                                   // the same stream on every rank: the generator is counter based
    enqueue_schedule(h, p.samples_per_epoch, per_epoch);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(h->stream));
}

template <class T>
void shard_batch_typed(mi355rec_mf *h, int b) {
    MfParams<T> p{};
    fill_params(h, p);
    const int wgs = div_up(p.tasks_per_batch, 4);
    p.wg_base = h->shard_rank;
    p.wg_stride = h->shard_world;
    launch_batch<MI355REC_MF_BPR, T>(h, p, b, true, wgs > h->shard_rank ? (wgs - h->shard_rank + h->shard_world - 1) / h->shard_world : 0);
    hipLaunchKernelGGL((mf_shard_rows_kernel<T, true>), dim3(wgs), dim3(256), 0, h->stream, p, b, h->shard_rank, h->shard_world,
                       h->shard_slots_per_rank, as<T>(h->shard_send));
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(h->stream));                // the caller's collective may run on any stream
}

template <class T>
void shard_merge_typed(mi355rec_mf *h, int b) {
    MfParams<T> p{};
    fill_params(h, p);
    hipLaunchKernelGGL((mf_shard_rows_kernel<T, false>), dim3(div_up(p.tasks_per_batch, 4)), dim3(256), 0, h->stream, p, b, h->shard_rank,
                       h->shard_world, h->shard_slots_per_rank, as<T>(h->shard_recv));
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(h->stream));                // the slab may be overwritten by the next exchange
}

template <class T>
void shard_end_typed(mi355rec_mf *h) {
    const long long B = h->cfg.batch_size, per_epoch = batches_per_epoch(h);
    MfParams<T> p{};
    fill_params(h, p);
    hipLaunchKernelGGL(mf_stream_end_kernel<T>, dim3(1), dim3(64), 0, h->stream, p, per_epoch);
    h->timer.stop(h->stream);
    h->batches_done += per_epoch;
    h->last_call_samples = per_epoch * B;
    finish_call(h, per_epoch * B, per_epoch);               // (the loss is this rank's share)
}

// An error inside an exact multi-GPU epoch ends that epoch: the handle leaves the sharded state, so the next call is refused with
// "begin_epoch has not been called" (or a plain epoch may follow) instead of continuing on a half-exchanged mini-batch.
struct ShardEpochGuard {
    mi355rec_mf *h;
    bool ok = false;
    ~ShardEpochGuard() { if (!ok) h->shard_rank = -1; }
};

}  // namespace

extern "C" int mi355rec_mf_shard_begin_epoch(mi355rec_mf_t h, int32_t rank, int32_t world, void **d_send, void **d_recv,
                                             uint64_t *bytes_per_rank, int32_t *n_batches) {
    return guarded([&] {
        MI_REQUIRE(h && d_send && d_recv && bytes_per_rank && n_batches, "NULL argument");
        MI_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d of %d", rank, world);
        if (h->cfg.algorithm != MI355REC_MF_BPR || h->cfg.sgd_mode != MI355REC_SGD)
            fail(MI355REC_E_UNSUPPORTED, "the exact multi-GPU mode covers MF_BPR with sgd (optimiser moments and the FunkSVD global bias are not exchanged)");
        ensure_device();
        const int tpb = per_sample(h) * h->cfg.batch_size;
        const int wgs = div_up(tpb, 4);
        ShardEpochGuard guard{h};
        h->shard_rank = rank;
        h->shard_world = world;
        h->shard_slots_per_rank = div_up(wgs, world) * 4;    // whole workgroups: a wide list's four quarters stay together
        const size_t ts = h->f64 ? sizeof(double) : sizeof(float);
        const size_t per_rank = (size_t)h->shard_slots_per_rank * h->k * ts;
        if (h->shard_send.count < per_rank) h->shard_send.alloc_zero(per_rank, h->stream);
        if (h->shard_recv.count < per_rank * world) h->shard_recv.alloc_zero(per_rank * world, h->stream);
        if (h->f64) shard_begin_typed<double>(h); else shard_begin_typed<float>(h);
        h->shard_batches = batches_per_epoch(h);
        *d_send = h->shard_send.ptr;
        *d_recv = h->shard_recv.ptr;
        *bytes_per_rank = per_rank;
        *n_batches = (int32_t)h->shard_batches;
        guard.ok = true;
    });
}

extern "C" int mi355rec_mf_shard_batch(mi355rec_mf_t h, int32_t batch) {
    return guarded([&] {
        MI_REQUIRE(h && h->shard_rank >= 0, "mi355rec_mf_shard_begin_epoch has not been called");
        MI_REQUIRE(batch >= 0 && batch < h->shard_batches, "mini-batch %d of %lld", batch, h->shard_batches);
        ensure_device();
        ShardEpochGuard guard{h};
        if (h->f64) shard_batch_typed<double>(h, batch); else shard_batch_typed<float>(h, batch);
        guard.ok = true;
    });
}

extern "C" int mi355rec_mf_shard_merge(mi355rec_mf_t h, int32_t batch) {
    return guarded([&] {
        MI_REQUIRE(h && h->shard_rank >= 0, "mi355rec_mf_shard_begin_epoch has not been called");
        MI_REQUIRE(batch >= 0 && batch < h->shard_batches, "mini-batch %d of %lld", batch, h->shard_batches);
        ensure_device();
        ShardEpochGuard guard{h};
        if (h->f64) shard_merge_typed<double>(h, batch); else shard_merge_typed<float>(h, batch);
        guard.ok = true;
    });
}

extern "C" int mi355rec_mf_shard_end_epoch(mi355rec_mf_t h) {
    return guarded([&] {
        MI_REQUIRE(h && h->shard_rank >= 0, "mi355rec_mf_shard_begin_epoch has not been called");
        ensure_device();
        ShardEpochGuard guard{h};                            // (ends the sharded state on success too)
        if (h->f64) shard_end_typed<double>(h); else shard_end_typed<float>(h);
    });
}

// ---- replica-batched epochs: R independent models, one launch per mini-batch index ---------------------------------------------
struct mi355rec_mf_group {
    std::vector<mi355rec_mf *> members;      // not owned
    bool f64 = false;
    int algorithm = 0, klass = 0, tasks_per_batch = 0;
    bool plain_sgd = false;                 // every member runs sgd_mode "sgd"
    long long batches_per_epoch = 0;
    hipStream_t stream = nullptr;
    StreamTimer timer;
    DispatchTimers dispatch_timers;
    int max_timed = 0;
    DeviceBuffer<unsigned char> table;       // MfParams<T>[R]
    std::vector<unsigned char> host_table;
    DeviceBuffer<FastSchedParams> sched_table;   // valid when every member is on the in-LDS schedule (all_fast)
    std::vector<FastSchedParams> host_sched_table;
    bool all_fast = false;
    hipEvent_t fork = nullptr;
    std::vector<hipEvent_t> join;
    hipGraphExec_t graph = nullptr;
    bool graph_failed = false;
    mi355rec_stats stats{};

    ~mi355rec_mf_group() {
        if (stream) (void)hipStreamSynchronize(stream);
        if (graph) (void)hipGraphExecDestroy(graph);
        timer.destroy();
        dispatch_timers.destroy();
        if (fork) (void)hipEventDestroy(fork);
        for (auto e : join) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

// One epoch of every member: samplers and schedules on the members' own streams (parallel branches, also when captured), then
// the shared chain of mini-batch launches on the group's stream.
// sampler + schedule of all members (all on the in-LDS schedule): four launches
template <class T>
void group_enqueue_schedule(mi355rec_mf_group *g, const MfParams<T> *table, const FastSchedParams *sched_table, hipStream_t s) {
    const long long nb = g->batches_per_epoch;
    const int R = (int)g->members.size();
    mi355rec_mf *h0 = g->members[0];
    const FastSchedParams &f0 = g->host_sched_table[0];
    const dim3 sgrid(div_up(nb * (long long)h0->cfg.batch_size, 256), R);
    if (g->algorithm == MI355REC_MF_BPR) hipLaunchKernelGGL((mf_group_sample_kernel<MI355REC_MF_BPR, T>), sgrid, dim3(256), 0, s, table);
    else hipLaunchKernelGGL((mf_group_sample_kernel<MI355REC_MF_FUNK_SVD, T>), sgrid, dim3(256), 0, s, table);
    hipLaunchKernelGGL(mf_group_epoch_advance_kernel<T>, dim3(div_up(R, 64)), dim3(64), 0, s, table, R);
    static bool attr_set[64] = {};
    set_sched_sort_attribute(reinterpret_cast<const void *>(mf_group_sched_sort_kernel), attr_set);
    int max_entries = 0;
    for (const auto &f : g->host_sched_table) max_entries = std::max(max_entries, f.n_entries);
    hipLaunchKernelGGL(mf_group_sched_sort_kernel, dim3((unsigned)nb, R), dim3(SCHED_THREADS), sched_lds_bytes(f0.np), s, sched_table);
    hipLaunchKernelGGL(mf_group_sched_emit_kernel, dim3(div_up(f0.tasks_per_batch, 256), (unsigned)nb, R), dim3(256), 0, s, sched_table);
    hipLaunchKernelGGL(mf_group_sched_finish_kernel, dim3(div_up(max_entries, 256), R), dim3(256), 0, s, sched_table);
}

// the shared chain of mini-batch launches of one epoch, on the group's stream
template <class T>
void group_enqueue_batches(mi355rec_mf_group *g, const MfParams<T> *table, bool timed) {
    const long long nb = g->batches_per_epoch;
    // a third of the slots' workgroups: the kernel loops over the slots in use (all of them when a member is on the general schedule)
    const int wgs = div_up(div_up(g->tasks_per_batch, 4), 3), R = (int)g->members.size();
    for (long long b = 0; b < nb; ++b) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) g->dispatch_timers.next(e0, e1, g->max_timed);
        if (g->algorithm == MI355REC_MF_BPR) launch_group_batch<MI355REC_MF_BPR, T>(g->stream, table, g->klass, wgs, R, (int)b, g->plain_sgd, e0, e1);
        else launch_group_batch<MI355REC_MF_FUNK_SVD, T>(g->stream, table, g->klass, wgs, R, (int)b, g->plain_sgd, e0, e1);
    }
    hipLaunchKernelGGL(mf_group_stream_end_kernel<T>, dim3(div_up(R, 64)), dim3(64), 0, g->stream, table, R, nb);
}

template <class T>
void group_enqueue_epoch(mi355rec_mf_group *g, bool timed) {
    const long long nb = g->batches_per_epoch;
    if (g->all_fast) {
        group_enqueue_schedule<T>(g, reinterpret_cast<const MfParams<T> *>(g->table.ptr), g->sched_table.ptr, g->stream);
    } else {
    MI_HIP(hipEventRecord(g->fork, g->stream));
    for (size_t m = 0; m < g->members.size(); ++m) {
        mi355rec_mf *h = g->members[m];
        MI_HIP(hipStreamWaitEvent(h->stream, g->fork, 0));
        MfParams<T> p{};
        fill_params(h, p);
        launch_sampler(h, p);
        enqueue_schedule(h, p.samples_per_epoch, nb);
        MI_HIP(hipEventRecord(g->join[m], h->stream));
        MI_HIP(hipStreamWaitEvent(g->stream, g->join[m], 0));
    }
    }
    group_enqueue_batches<T>(g, reinterpret_cast<const MfParams<T> *>(g->table.ptr), timed);
}

template <class T>
void group_ensure_graph(mi355rec_mf_group *g) {
    if (g->graph || g->graph_failed) return;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        try {
            group_enqueue_epoch<T>(g, false);
        } catch (...) {
            (void)hipStreamEndCapture(g->stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            g->graph_failed = true;
            return;
        }
        e = hipStreamEndCapture(g->stream, &graph);
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&g->graph, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g->graph = nullptr;
        g->graph_failed = true;
    }
}

template <class T>
void group_run_epochs_typed(mi355rec_mf_group *g, int n_epochs) {
    const long long nb = g->batches_per_epoch;
    const int R = (int)g->members.size();
    std::vector<unsigned char> table(sizeof(MfParams<T>) * (size_t)R);
    for (int m = 0; m < R; ++m) {
        mi355rec_mf *h = g->members[m];
        MI_REQUIRE(h->shard_rank < 0, "member %d is inside an exact multi-GPU epoch", m);
        ensure_stream_capacity(h, (size_t)(nb * h->cfg.batch_size), nb, true);
        MfParams<T> p;
        memset(&p, 0, sizeof(p));
        fill_params(h, p);
        memcpy(table.data() + sizeof(MfParams<T>) * (size_t)m, &p, sizeof(MfParams<T>));
        begin_call(h);
        MI_HIP(hipStreamSynchronize(h->stream));      // (its clears and first-touch memsets run on the member's own stream)
    }
    std::vector<FastSchedParams> sched((size_t)R);
    bool all_fast = !getenv("MI355REC_MF_GROUP_PER_MEMBER_SCHEDULE");
    for (int m = 0; m < R; ++m) {
        mi355rec_mf *h = g->members[m];
        all_fast = all_fast && h->fast_schedule && fast_schedule_fits(h, nb);
        if (all_fast) sched[m] = fast_sched_params(h, nb * (long long)h->cfg.batch_size);
    }
    const bool sched_changed = all_fast != g->all_fast || (all_fast && (g->host_sched_table.size() != sched.size() ||
                               memcmp(g->host_sched_table.data(), sched.data(), sizeof(FastSchedParams) * sched.size()) != 0));
    if (sched_changed) {
        if (g->graph) {
            (void)hipGraphExecDestroy(g->graph);
            g->graph = nullptr;
        }
        MI_HIP(hipStreamSynchronize(g->stream));
        g->all_fast = all_fast;
        g->host_sched_table = sched;
        if (all_fast) {
            if (g->sched_table.count < sched.size()) g->sched_table.alloc(sched.size());
            MI_HIP(hipMemcpy(g->sched_table.ptr, sched.data(), sizeof(FastSchedParams) * sched.size(), hipMemcpyHostToDevice));
        }
    }
    if (table != g->host_table) {                 // a member re-allocated its stream buffers: the graph holds the old addresses
        if (g->graph) {
            (void)hipGraphExecDestroy(g->graph);
            g->graph = nullptr;
        }
        MI_HIP(hipStreamSynchronize(g->stream));
        if (g->table.count < table.size()) g->table.alloc(table.size());
        MI_HIP(hipMemcpy(g->table.ptr, table.data(), table.size(), hipMemcpyHostToDevice));
        g->host_table = table;
    }
    g->dispatch_timers.reset();
    const long long timed_epochs = g->max_timed > 0 ? std::min<long long>(n_epochs, (g->max_timed + nb - 1) / nb) : 0;
    bool use_graph = nb <= GRAPH_SEGMENT && n_epochs - timed_epochs > 0 && !getenv("MI355REC_NO_GRAPH");
    if (use_graph) {
        group_ensure_graph<T>(g);
        use_graph = g->graph != nullptr;
    }
    g->timer.start(g->stream);
    for (long long e = 0; e < n_epochs; ++e) {
        if (e < timed_epochs || !use_graph) group_enqueue_epoch<T>(g, e < timed_epochs);
        else MI_HIP(hipGraphLaunch(g->graph, g->stream));
    }
    g->timer.stop(g->stream);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(g->stream));
    mi355rec_stats &st = g->stats;
    st = mi355rec_stats{};
    st.call_ms = g->timer.elapsed_ms();
    st.kernel_ms = g->dispatch_timers.total_ms();
    st.n_timed = g->dispatch_timers.used;
    st.n_launches = nb * n_epochs;
    for (int m = 0; m < R; ++m) {
        mi355rec_mf *h = g->members[m];
        const long long n = nb * n_epochs * (long long)h->cfg.batch_size;
        h->host_loss.resize(h->loss_slots.count);
        h->loss_slots.download(h->host_loss.data(), h->loss_slots.count, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
        double loss = 0;
        for (double v : h->host_loss) loss += v;
        h->batches_done += nb * n_epochs;
        h->last_call_samples = n_epochs > 0 ? nb * (long long)h->cfg.batch_size : 0;
        h->stats = mi355rec_stats{};
        h->stats.call_ms = st.call_ms;
        h->stats.n_launches = st.n_launches;
        h->stats.n_units = n;
        h->stats.algorithmic_bytes = bytes_per_sample(h) * (double)n;
        h->stats.loss = loss;
        st.n_units += n;
        st.algorithmic_bytes += h->stats.algorithmic_bytes;
        st.loss += loss;
    }
}

}  // namespace

extern "C" int mi355rec_mf_group_create(mi355rec_mf_group_t *out, const mi355rec_mf_t *members, int32_t n_members) {
    return guarded([&] {
        MI_REQUIRE(out && members, "NULL argument");
        MI_REQUIRE(n_members >= 1 && n_members <= 65535, "n_members = %d out of range", n_members);
        ensure_device();
        std::unique_ptr<mi355rec_mf_group> g(new mi355rec_mf_group());
        const mi355rec_mf *first = members[0];
        MI_REQUIRE(first, "member 0 is NULL");
        if (first->cfg.algorithm == MI355REC_MF_ASY_SVD) fail(MI355REC_E_UNSUPPORTED, "ASY_SVD has no mini-batches to share a launch");
        g->f64 = first->f64;
        g->algorithm = first->cfg.algorithm;
        g->klass = kernel_class(first);
        if (g->klass < 0)
            fail(MI355REC_E_UNSUPPORTED, "n_factors = %d runs on the any-k kernel, which has no replica-batched form (use a multiple of %d up to 512)",
                 first->k, first->f64 ? 2 : 4);
        g->tasks_per_batch = per_sample(first) * first->cfg.batch_size;
        g->batches_per_epoch = batches_per_epoch(first);
        for (int m = 0; m < n_members; ++m) {
            mi355rec_mf *h = members[m];
            MI_REQUIRE(h, "member %d is NULL", m);
            for (int o = 0; o < m; ++o) MI_REQUIRE(members[o] != h, "member %d is listed twice", m);
            // what ONE launch must share: the kernel instance and the grid; everything else is per model
            MI_REQUIRE(h->cfg.algorithm == g->algorithm && h->f64 == g->f64 && kernel_class(h) == g->klass,
                       "member %d runs a different kernel instance (algorithm, precision or n_factors class) than member 0", m);
            MI_REQUIRE(per_sample(h) * h->cfg.batch_size == g->tasks_per_batch && batches_per_epoch(h) == g->batches_per_epoch,
                       "member %d has a different batch_size or number of mini-batches per epoch than member 0", m);
            g->members.push_back(h);
        }
        g->plain_sgd = true;
        for (const mi355rec_mf *h : g->members) g->plain_sgd = g->plain_sgd && h->cfg.sgd_mode == MI355REC_SGD;
        MI_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        g->timer.init();
        MI_HIP(hipEventCreateWithFlags(&g->fork, hipEventDisableTiming));
        g->join.resize(n_members, nullptr);
        for (auto &e : g->join) MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        *out = g.release();
    });
}

extern "C" int mi355rec_mf_group_run_epochs(mi355rec_mf_group_t g, int32_t n_epochs) {
    return guarded([&] {
        MI_REQUIRE(g, "NULL handle");
        MI_REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        ensure_device();
        if (g->f64) group_run_epochs_typed<double>(g, n_epochs); else group_run_epochs_typed<float>(g, n_epochs);
    });
}

extern "C" int mi355rec_mf_group_set_profiling(mi355rec_mf_group_t g, int32_t max_timed_launches) {
    return guarded([&] {
        MI_REQUIRE(g, "NULL handle");
        MI_REQUIRE(max_timed_launches >= 0 && max_timed_launches <= 65536, "max_timed_launches out of range");
        ensure_device();
        g->max_timed = max_timed_launches;
        g->dispatch_timers.reserve(max_timed_launches);
    });
}

extern "C" int mi355rec_mf_group_get_stats(mi355rec_mf_group_t g, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(g && stats, "NULL argument");
        *stats = g->stats;
    });
}

extern "C" void mi355rec_mf_group_destroy(mi355rec_mf_group_t g) { delete g; }

extern "C" int mi355rec_mf_get_factors(mi355rec_mf_t h, float *U, float *V, float *bu, float *bi, float *mu) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        if (h->f64) get_factors_typed<double, float>(h, U, V, bu, bi, mu); else get_factors_typed<float, float>(h, U, V, bu, bi, mu);
    });
}

extern "C" int mi355rec_mf_get_factors_f64(mi355rec_mf_t h, double *U, double *V, double *bu, double *bi, double *mu) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        if (h->f64) get_factors_typed<double, double>(h, U, V, bu, bi, mu); else get_factors_typed<float, double>(h, U, V, bu, bi, mu);
    });
}

extern "C" int mi355rec_mf_get_last_samples(mi355rec_mf_t h, int32_t *u, int32_t *i, int32_t *j, float *rating, int64_t cap,
                                            int64_t *n) {
    return guarded([&] {
        MI_REQUIRE(h && n, "NULL argument");
        ensure_device();
        *n = h->last_call_samples;
        const size_t m = (size_t)std::min<long long>(cap, h->last_call_samples);
        hipStream_t s = h->stream;
        if (u) h->su.download(u, m, s);
        if (i) h->si.download(i, m, s);
        if (j && h->cfg.algorithm == MI355REC_MF_BPR) h->sj.download(j, m, s);
        if (rating && h->cfg.algorithm != MI355REC_MF_BPR) h->sr.download(rating, m, s);
        MI_HIP(hipStreamSynchronize(s));
    });
}

extern "C" int mi355rec_mf_set_profiling(mi355rec_mf_t h, int32_t max_timed_launches) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        MI_REQUIRE(max_timed_launches >= 0 && max_timed_launches <= 65536, "max_timed_launches out of range");
        ensure_device();
        h->max_timed = max_timed_launches;
        h->dispatch_timers.reserve(max_timed_launches);
    });
}

extern "C" int mi355rec_mf_get_phase_ticks(mi355rec_mf_t h, uint64_t *out, int64_t cap, int64_t *n) {
    return guarded([&] {
        MI_REQUIRE(h && n, "NULL argument");
        ensure_device();
        *n = (int64_t)h->ticks.count;
        if (out && cap > 0) {
            MI_HIP(hipStreamSynchronize(h->stream));
            MI_HIP(hipMemcpy(out, h->ticks.ptr, sizeof(uint64_t) * std::min<size_t>((size_t)cap, h->ticks.count), hipMemcpyDeviceToHost));
        }
    });
}

extern "C" int mi355rec_mf_get_stats(mi355rec_mf_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_mf_destroy(mi355rec_mf_t h) { delete h; }
