// slim.hip -- SLIM-BPR epoch on MI355X (gfx950).
//
// Replaces SLIM_BPR/Cython/SLIM_BPR_Cython_Epoch.pyx (reference): epochIteration_Cython :212-317 with the dense
// store (:129) and the symmetric triangular store (Triangular_Matrix :1226-1330), sampleBPR_Cython :439-483,
// adaptive_gradient :398-436, get_S :343-391.  The wrapper hard-codes batch_size = 1
// (SLIM_BPR/Cython/SLIM_BPR_Cython.py:140), so an epoch is n_users + 1 STRICTLY ORDERED SGD steps.
//
// Design.  The sample stream does not depend on S, so WHICH earlier step a step has to wait for is
// known before the first step runs, and an epoch is ONE persistent kernel that executes the stream as a dataflow graph.  What
// bounds it is the longest chain of dependent steps (the steps on the busiest item row: 1 214 of 138 494 at the ML-20M shape),
// i.e. the price of ONE hand-off between two steps -- not bytes.  Round 4 rebuilds both stores around that price:
//   dense store       step t owns rows i_t and j_t.  The BUSIEST rows (top-H items by their number of steps in this stream,
//                     chosen on the device) are kept in the LDS of an OWNING workgroup for the whole launch: its 16 wavefronts
//                     take the row's steps in turn, each prefetching everything that does not depend on the row (profile,
//                     the other row's cells once that row's ticket has come up) before its turn; a link of the chain is then
//                     an LDS gather + wavefront reduction + LDS scatter (a few hundred cycles) instead of ticket poll ->
//                     gather -> write-through -> drain -> ticket through L2 (4.8 us per link under load in round 3).
//                     All other steps are run by "cold" wavefronts (one step per wavefront, no barriers), ordered per item by
//                     numbered tickets as before.  Two busy rows in one step talk through a two-word mailbox.
//   symmetric store   cell (r, c) aliases (c, r), so ownership is per CELL and no row can be owned.  Every cell of the packed
//                     lower triangle is an 8-byte GRANULE {float value, tag of the step that wrote it}, written by ONE
//                     write-through store: a reader that knows which step wrote the cell last (`pred`, from a sort of the
//                     stream's (cell, step) pairs) polls THE CELL until the tag matches -- one round trip per link instead of
//                     done-flag poll + gather + drain + flag store, and no step ever drains its stores.  The per-item
//                     optimiser cells travel the same way (two granules per float64).  Steps with profiles of more than 256
//                     entries are run by a whole workgroup (all their granules in flight at once), the others by a wavefront.
// The sample stream of epoch e + 1 is drawn and scheduled on a second stream while the kernel of epoch e runs (StreamSet x 2).
// Steps are claimed from an in-order queue, so a waiting step only ever waits for steps that are already running: no deadlock
// whatever the residency (the owners of the dense store are the exception: they are leased to one launch at a time, see
// OwnerLease).  Exact sequential semantics (1e-5 element-wise against the float64 oracle for all four optimisers, both stores,
// at the full BASELINE config-3 shape).  4-byte gathers / scatters, no dense contraction: no MFMA.
#include "common.h"

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <mutex>
#include "sampling.cuh"
#include "topk.cuh"
#include "wave.cuh"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <memory>
#include <type_traits>

namespace mi355rec {
namespace {

constexpr int LOSS_SLOTS = 1024;
constexpr int FLOW_THREADS = 1024;                    // 16 wavefronts: the turn takers of an owned row, or 16 independent steps
constexpr int FLOW_WAVES = FLOW_THREADS / 64;
constexpr int FLOW_REGS = 4;                          // profile entries per lane whose cells stay in registers between the passes
constexpr int MAX_OWNERS = 192;
constexpr long long SPIN_LIMIT_TICKS = 500000000ll;   // 5 s of the 100 MHz wall clock: a stuck hand-off aborts instead of hanging
constexpr unsigned long long MAIL_EMPTY = ~0ull;

struct alignas(8) Granule { float v; unsigned tag; };

// Everything a step needs before its first gather, in ONE 32-byte load (instead of stream position -> sample -> CSR bounds -> ticket
// numbers: four dependent round trips).  Three uses:
//   dense store, steps in stream order / compacted into the cold queue:  a, b = ticket numbers of items i and j
//   dense store, the list of an owned row (sorted order):  i = the OTHER item, j = role of the owned row (0 positive, 1 negative),
//                                                       a = the other item's ticket number, b = 1 if that row is owned, too
//   symmetric store, stream order:  a, b = the step before this one on item i / j (-1: none), (t, c) = first cell slot (low, high)
struct alignas(32) StepDesc { int rs, L, i, j, a, b, t, c; };

// Steps are handed to single wavefronts: one of them fetches LQ_CHUNK consecutive steps from the global in-order queue (one device
// atomic per chunk: 88 per microsecond is all one word sustains) and the wavefronts of the workgroup pop them one by one from LDS.
constexpr int LQ_CHUNK = 16, LQ_RING = 8;
struct LocalQueue { int next, ready; int base[LQ_RING]; };
constexpr int NO_STEP = 0x7fffffff;

template <class T>
struct SlimParams {
    int n_users, n_items, symmetric, sgd_mode;
    T lr, li_reg, lj_reg, gamma, beta_1, beta_2, one_m_gamma, one_m_beta_1, one_m_beta_2;
    double beta_1_d, beta_2_d;
    unsigned long long seed;
    const int *indptr, *indices;
    T *S;                           // dense store: n_items x n_items
    Granule *G;                     // symmetric store: packed lower triangle of {value, tag of the step that wrote it}
    T *c1, *c2;                     // dense store: per-ITEM optimiser scalars (.pyx:177-181): cache / first moment, second moment
    Granule *oc;                    // symmetric store: the same as [n_items][4] granules (c1 high, c1 low, c2 high, c2 low)
    const int *su, *si, *sj;        // sample stream of the call
    const int *seq;                 // dense: [2 n_steps] ticket numbers of step t on item i_t (2t) and item j_t (2t + 1)
    const int *iprev;               // symmetric: [2 n_steps] the step before t on item i_t / j_t in this call (-1: none)
    const long long *cellptr;       // symmetric: first cell slot of every step (2 per profile entry: row i, row j)
    const int *pred;                // symmetric: per cell slot, the step that touched the cell last (-1: nobody in this call)
    int *ticket;                    // dense: [n_items] steps of this call completed on the item
    int *queue;                     // [0] next step (of the cold list / the short profiles), [1] abort flag, [2] next long profile
    double *loss_slots;             // [LOSS_SLOTS]
    long long epoch;                // RNG counter base
    long long steps_before;         // steps executed before this call (Adam's beta^t, .pyx:313-317)
    int n_steps;
    unsigned tag_base;              // symmetric: step t of this call writes tag tag_base + t + 1
    // dense store, owned rows
    const int *hot_rank;            // [n_items] owner of the item's row, -1: nobody (the row stays in HBM)
    const int *hot_item, *lst_begin, *lst_len;   // [MAX_OWNERS] item, first position and length of its run in the sorted pairs
    const int *n_hot;               // owners in use (decided on the device)
    const StepDesc *desc;           // symmetric store: per step, in stream order
    const int *order;               // symmetric store: the steps with short profiles in stream order, then the others backwards
    int n_short;
    int nap;                        // how much longer a wavefront sleeps between polls once it has polled 24 times in vain
    const StepDesc *cold_desc;      // dense store: the steps with no owned row, in stream order
    const StepDesc *own_desc;       // dense store: per (item, step) pair in sorted order (only the owned items' runs are filled in)
    const int *n_cold;
    unsigned long long *prof;       // optional phase clocks (MI355REC_SLIM_PROF=1), NULL otherwise
    unsigned long long *mail_x, *mail_g;   // [n_steps] steps on TWO owned rows: sum over the negative item's row, sigmoid
};

template <class T> __device__ __forceinline__ T aload(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void astore(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ Granule gload(const Granule *p) {
    return __builtin_bit_cast(Granule, aload(reinterpret_cast<const unsigned long long *>(p)));
}
__device__ __forceinline__ void gstore(Granule *p, float v, unsigned tag) {
    astore(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, Granule{v, tag}));
}

// Triangular_Matrix.get_value/add_value (.pyx:1290-1330): in symmetric mode (r, c) with c > r lives at (c, r)
// and the store is the packed lower triangle, row r starting at r (r + 1) / 2 (:1237-1254): n (n + 1) / 2 cells
__host__ __device__ __forceinline__ size_t packed_cell(int r, int c) {
    if (c > r) { const int t = r; r = c; c = t; }
    return ((size_t)r * ((size_t)r + 1) >> 1) + (size_t)c;
}
template <class P> __device__ __forceinline__ float stored_value(const P &p, int r, int c) {      // for get_S
    return p.symmetric ? p.G[packed_cell(r, c)].v : (float)p.S[(size_t)r * p.n_items + c];
}

// The cell update v +- lr * (g - reg * v) (.pyx:283-309) with every operation rounded on its own, as the reference's scalar x86 code
// does.  A fused multiply-add is a hair more accurate -- and that hair matters to the sparse store: cells that TIE in the reference
// (a value far below the last bit of the increment it is added to: 1e-18 + 0.05) come out one unit in the last place apart with a
// fused add, and the per-row top-K selection then keeps different nodes (found with profiles of 1 850 items at 3 000 items).
template <class T>
__device__ __forceinline__ T cell_plus(T v, T lr, T g, T reg) {
#pragma clang fp contract(off)
    const T a = reg * v;
    const T b = g - a;
    const T c = lr * b;
    return v + c;
}
template <class T>
__device__ __forceinline__ T cell_minus(T v, T lr, T g, T reg) {
#pragma clang fp contract(off)
    const T a = reg * v;
    const T b = g - a;
    const T c = lr * b;
    return v - c;
}

__device__ __forceinline__ float root(float x) { return sqrtf(x); }
__device__ __forceinline__ double root(double x) { return sqrt(x); }
__device__ __forceinline__ float sigmoid_of_minus(float x) { return 1.f / (1.f + __expf(x)); }
__device__ __forceinline__ double sigmoid_of_minus(double x) { return 1.0 / (1.0 + exp(x)); }

// per-ITEM adaptive step (.pyx:398-436) on cells passed by reference; pw1 / pw2 = 1 - beta^t of this step
template <class T, class P>
__device__ __forceinline__ T slim_adapt_cells(const P &p, T g, T pw1, T pw2, T &c1, T &c2) {
    switch (p.sgd_mode) {
        case MI355REC_ADAGRAD:
            c1 = c1 + g * g;
            return g / (root(c1) + (T)1e-8);
        case MI355REC_RMSPROP:
            c1 = c1 * (T)p.gamma + (T)p.one_m_gamma * (g * g);
            return g / (root(c1) + (T)1e-8);
        case MI355REC_ADAM: {
            c1 = c1 * (T)p.beta_1 + (T)p.one_m_beta_1 * g;
            c2 = c2 * (T)p.beta_2 + (T)p.one_m_beta_2 * (g * g);
            return (c1 / pw1) / (root(c2 / pw2) + (T)1e-8);
        }
        default:
            return g;
    }
}
template <class T, class P>
__device__ __forceinline__ void adam_powers(const P &p, int t, T &pw1, T &pw2) {
    pw1 = (T)1;
    pw2 = (T)1;
    if (p.sgd_mode == MI355REC_ADAM) {
        const double tt = (double)(p.steps_before + t + 1);
        pw1 = (T)(1.0 - pow(p.beta_1_d, tt));
        pw2 = (T)(1.0 - pow(p.beta_2_d, tt));
    }
}

template <class T>
__global__ __launch_bounds__(256) void slim_sample_kernel(SlimParams<T> p, int *su, int *si, int *sj) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= p.n_steps) return;
    int u, i, j;
    sample_bpr(p.seed, (unsigned long long)(p.epoch * (long long)p.n_steps + t), p.n_users, p.n_items, p.indptr, p.indices, u, i, j);
    su[t] = u;
    si[t] = i;
    sj[t] = j;
}

// ---- dependencies of the stream ---------------------------------------------------------------------------------------
struct DepParams {
    int n_steps, n_items;
    const int *indptr, *indices, *su, *si, *sj;
    unsigned long long *keys;       // item pass: item << 32 | step;  cell pass: cell << 32 | step
    int *vals;                      // item pass: 2 step + role;      cell pass: cell slot
    const unsigned long long *keys_sorted;
    const int *vals_sorted;
    int *seq, *iprev;
    int *len2;                      // 2 L_u per step
    const long long *cellptr;
    int *pred;
    long long n_cells;
    int step_bits;                  // cell pass: key = cell << step_bits | step (the radix sort walks as few bits as the stream needs)
    int *bad_step;                  // symmetric store: first step whose negative item is in its user's profile (INT_MAX: none)
    unsigned no_cell;               // the diagonal's stand-in: all ones in the cell field (it is read but never written: it orders nothing)
    // owned rows of the dense store
    int *run_start;                 // [n_items] first position of the item's run in the sorted pairs
    unsigned *item_cnt;             // [n_items] steps of the stream on the item (0: memset)
    const unsigned *cnt_sorted;     // item_cnt in descending order ...
    const int *item_by_cnt;         // ... and whose count it is
    int *hot_rank, *hot_item, *lst_begin, *lst_len, *n_hot;
    int max_owners, min_steps;
    unsigned char *cold_flag;       // [n_steps] 1: neither row of the step is owned
    StepDesc *desc, *own_desc;
};

__global__ __launch_bounds__(256) void slim_item_keys_kernel(const DepParams d) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= d.n_steps) return;
    d.keys[2 * t] = ((unsigned long long)d.si[t] << 32) | (unsigned)t;
    d.vals[2 * t] = 2 * t;
    d.keys[2 * t + 1] = ((unsigned long long)d.sj[t] << 32) | (unsigned)t;
    d.vals[2 * t + 1] = 2 * t + 1;
    d.len2[t] = 2 * (d.indptr[d.su[t] + 1] - d.indptr[d.su[t]]);
}

// ticket number = how many earlier steps of the stream touch the same item = position inside the item's run; the step before it
// on the item; per item the run's start and length
__global__ __launch_bounds__(256) void slim_seq_kernel(const DepParams d) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int n2 = 2 * d.n_steps;
    if (q >= n2) return;
    const unsigned long long key = d.keys_sorted[q];
    const unsigned long long first_key = key & 0xFFFFFFFF00000000ull;
    int lo = 0, hi = q;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (d.keys_sorted[mid] < first_key) lo = mid + 1; else hi = mid;
    }
    const int slot = d.vals_sorted[q];
    d.seq[slot] = q - lo;
    d.iprev[slot] = q > lo ? d.vals_sorted[q - 1] >> 1 : -1;
    const int item = (int)(key >> 32);
    if (q == lo) d.run_start[item] = q;
    if (q + 1 == n2 || (int)(d.keys_sorted[q + 1] >> 32) != item) d.item_cnt[item] = (unsigned)(q - lo + 1);
}

// The busiest rows get owners: the first max_owners items of the descending count order that have at least min_steps steps.
__global__ __launch_bounds__(256) void slim_owners_kernel(const DepParams d) {
    const int h = threadIdx.x;
    __shared__ int s_n;
    if (h == 0) s_n = 0;
    __syncthreads();
    if (h < d.max_owners && h < d.n_items && (int)d.cnt_sorted[h] >= d.min_steps) {
        const int item = d.item_by_cnt[h];
        d.hot_rank[item] = h;
        d.hot_item[h] = item;
        d.lst_begin[h] = d.run_start[item];
        d.lst_len[h] = (int)d.cnt_sorted[h];
        atomicAdd(&s_n, 1);            // (the qualifying owners are a prefix of the order)
    }
    __syncthreads();
    if (h == 0) *d.n_hot = s_n;
}

// per step, stream order (dense store: after the owners are known)
__global__ __launch_bounds__(256) void slim_desc_kernel(const DepParams d, const int symmetric) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= d.n_steps) return;
    const int u = d.su[t], i = d.si[t], j = d.sj[t];
    const int rs = d.indptr[u], L = d.indptr[u + 1] - rs;
    StepDesc e;
    e.rs = rs; e.L = L; e.i = i; e.j = j;
    if (symmetric) {
        const long long cp = d.cellptr[t];
        e.a = d.iprev[2 * t]; e.b = d.iprev[2 * t + 1]; e.t = (int)(unsigned)cp; e.c = (int)(cp >> 32);
    } else {
        e.a = d.seq[2 * t]; e.b = d.seq[2 * t + 1]; e.t = t; e.c = 0;
        d.cold_flag[t] = d.hot_rank[i] < 0 && d.hot_rank[j] < 0;
    }
    d.desc[t] = e;
}
// per (item, step) pair in sorted order: the entries of the owned rows' lists
__global__ __launch_bounds__(256) void slim_owner_desc_kernel(const DepParams d) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= 2 * d.n_steps) return;
    if (d.hot_rank[(int)(d.keys_sorted[q] >> 32)] < 0) return;
    const int slot = d.vals_sorted[q], t = slot >> 1, role = slot & 1;
    const int u = d.su[t], other = role ? d.si[t] : d.sj[t];
    StepDesc e;
    e.rs = d.indptr[u]; e.L = d.indptr[u + 1] - e.rs; e.i = other; e.j = role;
    e.a = d.seq[2 * t + (1 - role)]; e.b = d.hot_rank[other] >= 0; e.t = t; e.c = 0;
    d.own_desc[q] = e;
}

// symmetric store: one wavefront per step lists the canonical cells of its two rows
__global__ __launch_bounds__(256) void slim_cell_keys_kernel(const DepParams d) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= d.n_steps) return;
    const int u = d.su[t], i = d.si[t], j = d.sj[t];
    const int rs = d.indptr[u], L = d.indptr[u + 1] - rs;
    const long long cp = d.cellptr[t];
    for (int idx = lane; idx < L; idx += 64) {
        const int s = d.indices[rs + idx];
        // (packed lower triangle, as packed_cell: fits 32 bits up to 92 681 items)
        const unsigned ci = s == i ? d.no_cell : (unsigned)packed_cell(i, s);
        const unsigned cj = s == j ? d.no_cell : (unsigned)packed_cell(j, s);
        // A negative item that the user has seen (the reference's sampler never draws one, .pyx:224-232; a replayed stream might):
        // cell (i, j) IS cell (j, i) in this store, the step would touch it twice and wait for its own tag.  Reported, not run.
        if (s == j) atomicMin(d.bad_step, t);
        d.keys[cp + 2 * idx] = ((unsigned long long)ci << d.step_bits) | (unsigned)t;
        d.vals[cp + 2 * idx] = (int)(cp + 2 * idx);
        d.keys[cp + 2 * idx + 1] = ((unsigned long long)cj << d.step_bits) | (unsigned)t;
        d.vals[cp + 2 * idx + 1] = (int)(cp + 2 * idx + 1);
    }
}

__global__ __launch_bounds__(256) void slim_pred_kernel(const DepParams d) {
    const long long q = blockIdx.x * 256ll + threadIdx.x;
    if (q >= d.n_cells) return;
    const unsigned long long key = d.keys_sorted[q];
    const unsigned cell = (unsigned)(key >> d.step_bits);
    int pred = -1;
    if (q > 0 && cell != d.no_cell) {
        const unsigned long long before = d.keys_sorted[q - 1];
        if ((unsigned)(before >> d.step_bits) == cell) pred = (int)(before & ((1ull << d.step_bits) - 1ull));
    }
    d.pred[d.vals_sorted[q]] = pred;
}

// ---- the stream ---------------------------------------------------------------------------------------------------------
// Every wait is a relaxed poll with a budget: a hand-off that does not arrive within SPIN_LIMIT_TICKS raises the abort flag
// (everybody stops waiting, the call fails) instead of hanging the device.
struct SpinGuard {
    unsigned polls = 0;
    long long t0 = 0;
};
template <class T>
__device__ __forceinline__ bool give_up(const SlimParams<T> &p, SpinGuard &g) {      // wave-uniform answer
    // a waiter that has polled for a while polls less often (p.nap; 0: always every 64 cycles)
    if (p.nap == 0 || g.polls < 24u) __builtin_amdgcn_s_sleep(1);
    else if (p.nap == 1) __builtin_amdgcn_s_sleep(4);
    else if (p.nap == 2) __builtin_amdgcn_s_sleep(12);
    else __builtin_amdgcn_s_sleep(32);
    if ((++g.polls & 127u) != 0) return false;
    int stop = aload(&p.queue[1]);
    const long long now = wall_clock64();
    if (g.t0 == 0) g.t0 = now;
    else if (now - g.t0 > SPIN_LIMIT_TICKS) { astore(&p.queue[1], 1); stop = 1; }
    return __builtin_amdgcn_readfirstlane(stop) != 0;
}

// A ticket says how many steps are still ahead of the waiter on that row, and no step takes less than a microsecond: a waiter
// `ahead` steps away sleeps a quarter of a microsecond per step ahead (at most 8 us) before it looks again.  (Polls are memory-side
// transactions: PMC round 4 counted 5 GB of them per epoch against 0.4 GB of algorithmic bytes.)
__device__ __forceinline__ void nap_by_distance(int ahead) {      // wave-uniform
    ahead = min(ahead - 1, 32);
    for (int n = 0; n < ahead; ++n) __builtin_amdgcn_s_sleep(8);
}

// One lane polls a word for the whole wavefront.
template <class T>
__device__ __forceinline__ bool wave_wait_word(const SlimParams<T> &p, const int *word, int want, int lane) {
    SpinGuard sg;
    for (;;) {
        int v = want;
        if (lane == 0) v = aload(word);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v == want) return true;
        if (p.nap) nap_by_distance(want - v);
        if (give_up(p, sg)) return false;
    }
}
template <class T>
__device__ __forceinline__ bool wave_wait_mail(const SlimParams<T> &p, unsigned long long *word, int lane, double &out) {
    SpinGuard sg;
    for (;;) {
        unsigned long long v = 0;
        if (lane == 0) v = aload(word);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        v = ((unsigned long long)hi << 32) | lo;
        if (v != MAIL_EMPTY) { out = __longlong_as_double((long long)v); return true; }
        if (give_up(p, sg)) return false;
    }
}

// A `volatile T *` into LDS that has lost its address space on the way (a function argument, the address of a __shared__ member) is
// read and written with flat_load / flat_store ... sc0 sc1, each behind an s_waitcnt vmcnt(0): the access goes down the vector-memory
// path, waits for every outstanding global load of the wavefront, and takes several hundred cycles -- found in round 6 on the turn word
// and the optimiser cells of an owned row, i.e. four such round trips inside every turn of the busiest row's chain.  The low 32 bits of
// a generic address inside the shared aperture are the LDS offset: through this cast the same accesses are ds_read / ds_write.
template <class T>
__device__ __forceinline__ __attribute__((address_space(3))) volatile T *as_lds(volatile T *q) {
    return (__attribute__((address_space(3))) volatile T *)(uintptr_t)(unsigned)(unsigned long long)q;
}

__device__ __forceinline__ unsigned long long shader_clock() {   // not reordered against memory operations
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Next step of the in-order queue for this wavefront (NO_STEP: the queue is empty or the launch is being abandoned).  The chunks are
// fetched in the order of their generations, so what a workgroup holds is always a prefix of what it will hold: a step it has not
// handed out yet can only be waited for by steps it has not handed out either.
template <class T, class ReadyPtr, class BasesPtr>
__device__ __forceinline__ int claim_step_on(const SlimParams<T> &p, const int lane, const int k, ReadyPtr ready, BasesPtr bases) {
    const int gen = k / LQ_CHUNK, off = k % LQ_CHUNK;
    SpinGuard sg;
    unsigned spins = 0;
    if (off == 0) {
        while (__builtin_amdgcn_readfirstlane(*ready) != gen)
            if ((++spins & 1023u) == 0 && give_up(p, sg)) return NO_STEP;
        int base = 0;
        if (lane == 0) base = aload(&p.queue[1]) ? NO_STEP : atomicAdd(&p.queue[0], LQ_CHUNK);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane == 0) bases[gen % LQ_RING] = base;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *ready = gen + 1;
        return base;
    }
    while (__builtin_amdgcn_readfirstlane(*ready) <= gen)
        if ((++spins & 1023u) == 0 && give_up(p, sg)) return NO_STEP;
    asm volatile("" ::: "memory");
    const int base = __builtin_amdgcn_readfirstlane(bases[gen % LQ_RING]);
    if (__builtin_amdgcn_readfirstlane(*ready) > gen + LQ_RING) {       // (the ring slot may have been reused: never seen, checked anyway)
        if (lane == 0) astore(&p.queue[1], 1);
        return NO_STEP;
    }
    return base >= NO_STEP - LQ_CHUNK ? NO_STEP : base + off;
}
// LDS_POLLS: the queue's two LDS words read and written as ds_read / ds_write (as_lds) instead of through the generic volatile pointers'
// flat_load ... sc0 sc1.  The dense store's cold steps want it (a poll comes back four times as fast, a claimed step starts sooner:
// epoch 1.77 -> 1.58-1.63 ms); the symmetric store's wavefronts, which ALL pass through here and are bound by the chain behind it,
// do not (10.4 -> 11.1 ms: the faster polls take issue slots from the wavefronts that work) -- both measured in round 6.
template <bool LDS_POLLS, class T>
__device__ __forceinline__ int claim_step(const SlimParams<T> &p, LocalQueue *lq, const int lane) {
    int k = 0;
    if (lane == 0) k = atomicAdd(&lq->next, 1);
    k = __builtin_amdgcn_readfirstlane(k);
    if constexpr (LDS_POLLS) return claim_step_on(p, lane, k, as_lds((volatile int *)&lq->ready), as_lds((volatile int *)lq->base));
    else return claim_step_on(p, lane, k, (volatile int *)&lq->ready, (volatile int *)lq->base);
}

// The sigmoid and the optimiser step of an OWNED row's step sit on the critical path of the whole epoch (the turn of the busiest
// row), so their instruction count matters: the argument is reduced in float64 (x log2(e) = n + f, |f| <= 1/2, exact), 2^f comes
// from v_exp_f32 and the reciprocals from v_rcp_f32 (1 ulp each): relative error of the step ~2e-7, against 1e-5 asked of the
// cells it moves.  Moments stay in float64.
__device__ __forceinline__ double fast_sigmoid_of_minus(double x) {
    const double y = fmin(fmax(x * 1.4426950408889634, -120.0), 120.0);
    const double n = rint(y);
    const float e = ldexpf(__builtin_amdgcn_exp2f((float)(y - n)), (int)n);
    return (double)__builtin_amdgcn_rcpf(1.f + e);
}
template <class P>
__device__ __forceinline__ double hot_adapt(const P &p, double g, double pw1, double pw2, double &c1, double &c2) {
    switch (p.sgd_mode) {
        case MI355REC_ADAGRAD:
            c1 = c1 + g * g;
            return (double)((float)g * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf((float)c1) + 1e-8f));
        case MI355REC_RMSPROP:
            c1 = c1 * (double)p.gamma + (double)p.one_m_gamma * (g * g);
            return (double)((float)g * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf((float)c1) + 1e-8f));
        case MI355REC_ADAM:
            c1 = c1 * (double)p.beta_1 + (double)p.one_m_beta_1 * g;
            c2 = c2 * (double)p.beta_2 + (double)p.one_m_beta_2 * (g * g);
            return (double)((float)(c1 / pw1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf((float)(c2 / pw2)) + 1e-8f));
        default:
            return g;
    }
}

// Profiles longer than the FLOW_REGS x 64 entries whose cells a wavefront keeps in registers are walked in BLOCKS of the same
// size with all loads of a block in flight (round 4's first version took them 64 at a time, one dependent round trip each: 14 % of
// the ML-20M users have more than 256 items, and those steps made up most of every critical section).
constexpr int FLOW_BLOCK = 64 * FLOW_REGS;

// ---- dense store ----------------------------------------------------------------------------------------------------------
// One step on two rows in HBM, run by ONE wavefront: tickets of the two items (lanes 0 and 1 poll), gathers, reduction, the two
// per-item optimiser steps, write-through scatters, drain, tickets passed on.
template <class T>
__device__ __forceinline__ void cold_step(const SlimParams<T> &p, const StepDesc e, const int lane) {
    const int t = e.t, i = e.i, j = e.j, rs = e.rs, L = e.L;
    const size_t n = (size_t)p.n_items;
    const unsigned long long k0 = p.prof ? shader_clock() : 0ull;
    int sv[FLOW_REGS];
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r) sv[r] = p.indices[rs + min(lane + 64 * r, L - 1)];      // L >= 1: users without interactions are never drawn
    const int want = lane == 0 ? e.a : (lane == 1 ? e.b : 0);
    {
        const int *word = &p.ticket[lane == 1 ? j : i];
        SpinGuard sg;
        for (;;) {
            const int v = lane < 2 ? aload(word) : 0;
            if (__all(v == want)) break;
            if (p.nap) nap_by_distance(max(__builtin_amdgcn_readlane(want - v, 0), __builtin_amdgcn_readlane(want - v, 1)));
            if (give_up(p, sg)) return;
        }
    }
    const unsigned long long k1 = p.prof ? shader_clock() : 0ull;
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    // the items' optimiser cells belong to whoever holds the items' tickets: lane 0 looks after item i, lane 1 after item j
    T oc1 = (T)0, oc2 = (T)0;
    if (lane < 2 && p.sgd_mode != MI355REC_SGD) {
        oc1 = aload(&p.c1[lane ? j : i]);
        if (p.sgd_mode == MI355REC_ADAM) oc2 = aload(&p.c2[lane ? j : i]);
    }
    T *Si = p.S + (size_t)i * n, *Sj = p.S + (size_t)j * n;
    T va[FLOW_REGS], vb[FLOW_REGS];
    T x = (T)0;
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r) {       // loads from clamped, always valid addresses, masked afterwards: one wait for all of them
        va[r] = aload(Si + sv[r]);
        vb[r] = aload(Sj + sv[r]);
    }
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r) {
        const bool live = lane + 64 * r < L;
        va[r] = live ? va[r] : (T)0;
        vb[r] = live ? vb[r] : (T)0;
        x += va[r] - vb[r];                                           // x_uij over the profile (.pyx:243-260)
    }
    for (int b0 = FLOW_BLOCK; b0 < L; b0 += FLOW_BLOCK) {             // profiles longer than 256
        int s[FLOW_REGS];
        T a[FLOW_REGS], b[FLOW_REGS];
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) s[r] = p.indices[rs + min(b0 + lane + 64 * r, L - 1)];
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            a[r] = aload(Si + s[r]);
            b[r] = aload(Sj + s[r]);
        }
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r)
            if (b0 + lane + 64 * r < L) x += a[r] - b[r];
    }
    x = wave_sum(x);
    const T g = sigmoid_of_minus(x);                                  // .pyx:263
    T pw1, pw2;
    adam_powers(p, t, pw1, pw2);
    const T step = slim_adapt_cells(p, g, pw1, pw2, oc1, oc2);        // item i on lane 0, item j on lane 1 (.pyx:267-268)
    if (lane < 2 && p.sgd_mode != MI355REC_SGD) {
        astore(&p.c1[lane ? j : i], oc1);
        if (p.sgd_mode == MI355REC_ADAM) astore(&p.c2[lane ? j : i], oc2);
    }
    const T gi = __shfl(step, 0), gj = __shfl(step, 1);
    if (lane == 0) atomicAdd(&p.loss_slots[t & (LOSS_SLOTS - 1)], (double)x * (double)x);
    // the two rows move (.pyx:271-309); write-through stores
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r) {
        if (lane + 64 * r < L) {
            const int s = sv[r];
            if (s != i) astore(Si + s, cell_plus(va[r], p.lr, gi, p.li_reg));
            if (s != j) astore(Sj + s, cell_minus(vb[r], p.lr, gj, p.lj_reg));
        }
    }
    for (int b0 = FLOW_BLOCK; b0 < L; b0 += FLOW_BLOCK) {
        int s[FLOW_REGS];
        T a[FLOW_REGS], b[FLOW_REGS];
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) s[r] = p.indices[rs + min(b0 + lane + 64 * r, L - 1)];
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            a[r] = aload(Si + s[r]);
            b[r] = aload(Sj + s[r]);
        }
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            if (b0 + lane + 64 * r < L) {
                if (s[r] != i) astore(Si + s[r], cell_plus(a[r], p.lr, gi, p.li_reg));
                if (s[r] != j) astore(Sj + s[r], cell_minus(b[r], p.lr, gj, p.lj_reg));
            }
        }
    }
    // publish: drain the write-through stores, then pass the tickets on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < 2) astore(&p.ticket[lane ? j : i], want + 1);
    if (p.prof && lane == 0) {
        unsigned long long *o = p.prof + 8 * MAX_OWNERS;
        atomicAdd(&o[0], 1ull);
        atomicAdd(&o[1], k1 - k0);                   // profile ids + ticket wait
        atomicAdd(&o[2], shader_clock() - k1);       // gathers ... tickets passed on
    }
}

// The steps of one OWNED row, in stream order, by the 16 wavefronts of the owning workgroup in turn.  `row` is the item's row of S
// in LDS (float32), `oc` its two optimiser cells (float64).  Entry k of the row's list is step t with the row in role 0 (the
// positive item) or 1 (the negative item); the OTHER row of the step is
//   in HBM    -> this wavefront does that row's half of the step as well: waits for its ticket, gathers its cells and sums them
//                BEFORE its turn, writes them back and passes the ticket on AFTER its turn;
//   owned too -> the two owners exchange two scalars through the step's mailbox (the negative item's owner sends its sum, the
//                positive item's owner answers with the sigmoid), both inside their turns.
// A turn: LDS gather, wavefront reduction, sigmoid, the item's optimiser step, LDS scatter, turn counter + 1 -- and nothing that
// leaves the compute unit: the profile's ids (up to OWN_IDS x 64 of them, two 16-bit ids per register: a row that fits the LDS has
// fewer than 65 536 columns) are in registers before the turn starts.
constexpr int OWN_IDS = 16;

template <class T>
__device__ __forceinline__ void owned_row(const SlimParams<T> &p, const int h, float *row, volatile int *turn_generic, volatile double *oc_generic,
                                          const int lane, const int wave) {
    auto turn = as_lds(turn_generic);
    auto oc = as_lds(oc_generic);
    const int item = p.hot_item[h], first = p.lst_begin[h], len = p.lst_len[h];
    const size_t n = (size_t)p.n_items;
    const float lr = (float)p.lr, li_reg = (float)p.li_reg, lj_reg = (float)p.lj_reg;
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = wave; k < len; k += FLOW_WAVES) {
        const unsigned long long k0 = p.prof ? shader_clock() : 0ull;
        const StepDesc e = p.own_desc[first + k];
        const int t = e.t, role = e.j, other = e.i, rs = e.rs, L = e.L;
        unsigned ids[OWN_IDS / 2];
#pragma unroll
        for (int r = 0; r < OWN_IDS; r += 2) {
            ids[r / 2] = 0;
            if (64 * r < L) {         // (wave-uniform: chunks the profile does not reach are not fetched)
                const unsigned lo = (unsigned)p.indices[rs + min(lane + 64 * r, L - 1)];
                const unsigned hi = (unsigned)p.indices[rs + min(lane + 64 * (r + 1), L - 1)];
                ids[r / 2] = lo | (hi << 16);
            }
        }
        auto id_of = [&](int r) -> int { return (int)((ids[r / 2] >> (16 * (r & 1))) & 0xffffu); };
        const bool mail = e.b != 0;
        // ---- before the turn: the other row's half ------------------------------------------------------------------------
        T *So = p.S + (size_t)other * n;
        T vo[FLOW_REGS];
        T oc1 = (T)0, oc2 = (T)0;
        double xo = 0.0;
        int want = 0;
        unsigned long long k1 = k0;
        if (!mail) {
            want = e.a;
            if (!wave_wait_word(p, &p.ticket[other], want, lane)) return;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            if (p.prof) k1 = shader_clock();
            if (lane == 0 && p.sgd_mode != MI355REC_SGD) {
                oc1 = aload(&p.c1[other]);
                if (p.sgd_mode == MI355REC_ADAM) oc2 = aload(&p.c2[other]);
            }
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) vo[r] = aload(So + id_of(r));
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {
                vo[r] = lane + 64 * r < L ? vo[r] : (T)0;
                xo += (double)vo[r];
            }
#pragma unroll
            for (int blk = 1; blk < OWN_IDS / FLOW_REGS; ++blk) {         // entries 256 .. 1023: ids in registers
                if (blk * FLOW_BLOCK < L) {
                    T a[FLOW_REGS];
#pragma unroll
                    for (int r = 0; r < FLOW_REGS; ++r) a[r] = aload(So + id_of(blk * FLOW_REGS + r));
#pragma unroll
                    for (int r = 0; r < FLOW_REGS; ++r)
                        if (blk * FLOW_BLOCK + lane + 64 * r < L) xo += (double)a[r];
                }
            }
            for (int b0 = 64 * OWN_IDS; b0 < L; b0 += FLOW_BLOCK) {
                T a[FLOW_REGS];
#pragma unroll
                for (int r = 0; r < FLOW_REGS; ++r) a[r] = aload(So + p.indices[rs + min(b0 + lane + 64 * r, L - 1)]);
#pragma unroll
                for (int r = 0; r < FLOW_REGS; ++r)
                    if (b0 + lane + 64 * r < L) xo += (double)a[r];
            }
            xo = wave_sum(xo);
        }
        double pw1, pw2;
        adam_powers(p, t, pw1, pw2);
        const unsigned long long k2 = p.prof ? shader_clock() : 0ull;
        // ---- the turn -------------------------------------------------------------------------------------------------------
        {
            SpinGuard sg;
            unsigned spins = 0;
            while (__builtin_amdgcn_readfirstlane(*turn) != k)
                if ((++spins & 1023u) == 0 && give_up(p, sg)) return;
        }
        __builtin_amdgcn_s_setprio(3);
        asm volatile("" ::: "memory");
        const unsigned long long k3 = p.prof ? shader_clock() : 0ull;
        double xr = 0.0;
        float vr[FLOW_REGS];            // the cells of the first 256 entries stay in registers between the sum and the update
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) vr[r] = lane + 64 * r < L ? row[id_of(r)] : 0.f;
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) xr += (double)vr[r];
        if (L > FLOW_BLOCK) {           // (14 % of the ML-20M users)
#pragma unroll
            for (int r = FLOW_REGS; r < OWN_IDS; ++r)
                if (64 * r < L) xr += lane + 64 * r < L ? (double)row[id_of(r)] : 0.0;
            for (int idx = lane + 64 * OWN_IDS; idx < L; idx += 64) xr += (double)row[p.indices[rs + idx]];      // (0.7 %)
        }
        xr = wave_sum(xr);
        const unsigned long long k3a = p.prof ? shader_clock() : 0ull;
        double g, x = 0.0;
        if (!mail) {
            x = role ? xo - xr : xr - xo;                             // x_uij = sum over S[i, .] - sum over S[j, .]
            g = fast_sigmoid_of_minus(x);
        } else if (role) {      // this row is the step's negative item: send the sum, wait for the sigmoid
            if (lane == 0) astore(&p.mail_x[t], (unsigned long long)__double_as_longlong(xr));
            if (!wave_wait_mail(p, &p.mail_g[t], lane, g)) { __builtin_amdgcn_s_setprio(0); return; }
        } else {
            if (!wave_wait_mail(p, &p.mail_x[t], lane, xo)) { __builtin_amdgcn_s_setprio(0); return; }
            x = xr - xo;
            g = fast_sigmoid_of_minus(x);
            if (lane == 0) astore(&p.mail_g[t], (unsigned long long)__double_as_longlong(g));
        }
        double c1 = oc[0], c2 = oc[1];
        const double gr = hot_adapt(p, g, pw1, pw2, c1, c2);
        if (lane == 0) { oc[0] = c1; oc[1] = c2; }
        const unsigned long long k3b = p.prof ? shader_clock() : 0ull;
        // (the row's cells are float32: their update in float32 arithmetic adds ~1e-7 of the INCREMENT to the rounding of the sum)
        const float reg = role ? lj_reg : li_reg, grf = (float)gr;
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            const int s = id_of(r);
            if (lane + 64 * r < L && s != item) row[s] = role ? cell_minus(vr[r], lr, grf, reg) : cell_plus(vr[r], lr, grf, reg);
        }
        if (L > FLOW_BLOCK) {
#pragma unroll
            for (int r = FLOW_REGS; r < OWN_IDS; ++r) {
                if (64 * r < L) {
                    const int s = id_of(r);
                    if (lane + 64 * r < L && s != item) row[s] = role ? cell_minus(row[s], lr, grf, reg) : cell_plus(row[s], lr, grf, reg);
                }
            }
            for (int idx = lane + 64 * OWN_IDS; idx < L; idx += 64) {
                const int s = p.indices[rs + idx];
                if (s != item) row[s] = role ? cell_minus(row[s], lr, grf, reg) : cell_plus(row[s], lr, grf, reg);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the row's new cells are in LDS before the next wavefront is let in
        if (lane == 0) *turn = k + 1;
        __builtin_amdgcn_s_setprio(0);
        const unsigned long long k4 = p.prof ? shader_clock() : 0ull;
        // ---- after the turn: the other row moves, its ticket is passed on -------------------------------------------------------
        if (!mail) {
            T po1, po2;
            adam_powers(p, t, po1, po2);
            const T go = slim_adapt_cells(p, (T)g, po1, po2, oc1, oc2);
            if (lane == 0 && p.sgd_mode != MI355REC_SGD) {
                astore(&p.c1[other], oc1);
                if (p.sgd_mode == MI355REC_ADAM) astore(&p.c2[other], oc2);
            }
            const T go_all = __shfl(go, 0);                            // (lane 0 holds the item's optimiser cells)
            // (the other row is the negative item when this one is the positive: .pyx:296-309)
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {
                const int s = id_of(r);
                if (lane + 64 * r < L && s != other)
                    astore(So + s, role ? cell_plus(vo[r], p.lr, go_all, p.li_reg) : cell_minus(vo[r], p.lr, go_all, p.lj_reg));
            }
#pragma unroll
            for (int blk = 1; blk < OWN_IDS / FLOW_REGS; ++blk) {
                if (blk * FLOW_BLOCK < L) {
                    T a[FLOW_REGS];
#pragma unroll
                    for (int r = 0; r < FLOW_REGS; ++r) a[r] = aload(So + id_of(blk * FLOW_REGS + r));
#pragma unroll
                    for (int r = 0; r < FLOW_REGS; ++r) {
                        const int s = id_of(blk * FLOW_REGS + r);
                        if (blk * FLOW_BLOCK + lane + 64 * r < L && s != other)
                            astore(So + s, role ? cell_plus(a[r], p.lr, go_all, p.li_reg) : cell_minus(a[r], p.lr, go_all, p.lj_reg));
                    }
                }
            }
            for (int b0 = 64 * OWN_IDS; b0 < L; b0 += FLOW_BLOCK) {
                int s[FLOW_REGS];
                T a[FLOW_REGS];
#pragma unroll
                for (int r = 0; r < FLOW_REGS; ++r) {
                    s[r] = p.indices[rs + min(b0 + lane + 64 * r, L - 1)];
                    a[r] = aload(So + s[r]);
                }
#pragma unroll
                for (int r = 0; r < FLOW_REGS; ++r)
                    if (b0 + lane + 64 * r < L && s[r] != other)
                        astore(So + s[r], role ? cell_plus(a[r], p.lr, go_all, p.li_reg) : cell_minus(a[r], p.lr, go_all, p.lj_reg));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) astore(&p.ticket[other], want + 1);
        }
        if (lane == 0 && !(mail && role)) atomicAdd(&p.loss_slots[t & (LOSS_SLOTS - 1)], x * x);
        if (p.prof) {
            acc[0] += 1; acc[1] += k1 - k0; acc[2] += k2 - k1; acc[3] += k3 - k2; acc[4] += k4 - k3; acc[5] += shader_clock() - k4;
            acc[6] += k3a - k3; acc[7] += k3b - k3a;
        }
    }
    if (p.prof && lane == 0) {       // entries | descriptor + ticket wait | gather + sum | wait for the turn | the turn | other row's write-back
        unsigned long long *o = p.prof + 8 * h;
        for (int c = 0; c < 8; ++c) atomicAdd(&o[c], acc[c]);
    }
}

// Workgroups 0 .. n_hot - 1 own a row each; the others (and an owner once its list is done) run the cold list.
template <class T>
__global__ __launch_bounds__(FLOW_THREADS) void slim_dense_flow_kernel(const SlimParams<T> p, const int owners) {
    extern __shared__ __attribute__((aligned(16))) float flow_lds[];
    __shared__ int s_turn;
    __shared__ LocalQueue s_queue;
    __shared__ double s_oc[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_queue.next = 0; s_queue.ready = 0; }
    const int n_hot = owners ? *p.n_hot : 0;
    if ((int)blockIdx.x < n_hot) {
        const int h = blockIdx.x, item = p.hot_item[h];
        T *Sr = p.S + (size_t)item * p.n_items;
        for (int c = tid; c < p.n_items; c += FLOW_THREADS) flow_lds[c] = (float)Sr[c];
        if (tid == 0) {
            s_turn = 0;
            s_oc[0] = p.sgd_mode != MI355REC_SGD ? (double)p.c1[item] : 0.0;
            s_oc[1] = p.sgd_mode == MI355REC_ADAM ? (double)p.c2[item] : 0.0;
        }
        __syncthreads();
        owned_row(p, h, flow_lds, &s_turn, s_oc, lane, wave);
        __syncthreads();
        for (int c = tid; c < p.n_items; c += FLOW_THREADS) Sr[c] = (T)flow_lds[c];
        if (tid == 0) {
            if (p.sgd_mode != MI355REC_SGD) p.c1[item] = (T)s_oc[0];
            if (p.sgd_mode == MI355REC_ADAM) p.c2[item] = (T)s_oc[1];
        }
    }
    __syncthreads();
    const int n_cold = *p.n_cold;
    for (;;) {          // in-order queue: everything a step can wait for is already running
        const int q = claim_step<true>(p, &s_queue, lane);
        if (q >= n_cold) break;
        cold_step(p, p.cold_desc[q], lane);
    }
}

// ---- symmetric store ------------------------------------------------------------------------------------------------------
// One step by ONE wavefront.  Every cell is a granule {value, tag of the step that wrote it}; the step knows which step wrote each
// of its cells last (`pred`), so it loads all its granules at once and re-loads only those whose tag is not there yet.  Its own
// stores carry its tag: nothing is drained, no flag is raised.  The optimiser cells of the two items travel as granules, too
// (lane 0: item i, lane 1: item j; float64 as two float32 halves, each with its own tag).  Between the arrival of a step's last
// tag and its stores sits the chain of the whole epoch (3 839 links at the ML-20M shape): sigmoid and optimiser step use the
// short forms of the owned rows' turns.
__device__ __forceinline__ bool tag_ok(int pred, unsigned tag, unsigned tag_base) { return pred < 0 || tag == tag_base + (unsigned)pred + 1u; }

// one block of FLOW_BLOCK profile entries: ids, last writers, granules of both rows -- fetched, then polled until every tag is there
struct SymBlock {
    int s[FLOW_REGS], pa[FLOW_REGS], pb[FLOW_REGS];
    Granule ga[FLOW_REGS], gb[FLOW_REGS];
};
__device__ __forceinline__ bool sym_fetch(const SlimParams<double> &p, const StepDesc &e, const long long cp, const int b0, const int lane,
                                          const bool poll, SymBlock &k, unsigned &repolls) {
    const int i = e.i, j = e.j, rs = e.rs, L = e.L;
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r) {
        const int at = min(b0 + lane + 64 * r, L - 1);
        k.s[r] = p.indices[rs + at];
        if (poll) {
            const int2 pp = *reinterpret_cast<const int2 *>(p.pred + cp + 2 * at);
            k.pa[r] = pp.x;
            k.pb[r] = pp.y;
        }
    }
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r) {
        const bool live = b0 + lane + 64 * r < L;
        k.pa[r] = live && poll ? k.pa[r] : -1;
        k.pb[r] = live && poll ? k.pb[r] : -1;
        k.ga[r] = gload(p.G + packed_cell(i, k.s[r]));
        k.gb[r] = gload(p.G + packed_cell(j, k.s[r]));
    }
    if (!poll) return true;
    SpinGuard sg;
    for (;;) {
        bool pending = false;
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            if (!tag_ok(k.pa[r], k.ga[r].tag, p.tag_base)) { pending = true; k.ga[r] = gload(p.G + packed_cell(i, k.s[r])); }
            if (!tag_ok(k.pb[r], k.gb[r].tag, p.tag_base)) { pending = true; k.gb[r] = gload(p.G + packed_cell(j, k.s[r])); }
        }
        if (!__any(pending)) return true;
        ++repolls;
        if (give_up(p, sg)) return false;
    }
}

__device__ __forceinline__ void sym_step(const SlimParams<double> &p, const int t, const int lane) {
    const StepDesc e = p.desc[t];
    const int i = e.i, j = e.j, L = e.L;
    const long long cp = (long long)(((unsigned long long)(unsigned)e.c << 32) | (unsigned)e.t);
    const unsigned long long k0 = p.prof ? shader_clock() : 0ull;
    unsigned repolls = 0;
    const unsigned my_tag = p.tag_base + (unsigned)t + 1u;
    const bool adaptive = p.sgd_mode != MI355REC_SGD, adam = p.sgd_mode == MI355REC_ADAM;
    // optimiser granules of the lane's item (requested first: they are polled last)
    Granule *oc = p.oc + 4 * (size_t)(lane == 1 ? j : i);
    const int ip = lane < 2 && adaptive ? (lane ? e.b : e.a) : -1;
    Granule o[4] = {{0.f, 0u}, {0.f, 0u}, {0.f, 0u}, {0.f, 0u}};
    if (lane < 2 && adaptive) {
        o[0] = gload(oc);
        o[1] = gload(oc + 1);
        if (adam) {
            o[2] = gload(oc + 2);
            o[3] = gload(oc + 3);
        }
    }
    SymBlock k;                                                       // the first block stays in registers for the second pass
    if (!sym_fetch(p, e, cp, 0, lane, true, k, repolls)) return;
    double x = 0.0;
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r)
        if (lane + 64 * r < L) x += (double)k.ga[r].v - (double)k.gb[r].v;                // x_uij over the profile (.pyx:243-260)
    for (int b0 = FLOW_BLOCK; b0 < L; b0 += FLOW_BLOCK) {                                 // profiles longer than 256
        SymBlock m;
        if (!sym_fetch(p, e, cp, b0, lane, true, m, repolls)) return;
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r)
            if (b0 + lane + 64 * r < L) x += (double)m.ga[r].v - (double)m.gb[r].v;
    }
    {
        SpinGuard sg;
        for (;;) {
            bool pending = false;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if ((c < 2 || adam) && !tag_ok(ip, o[c].tag, p.tag_base)) { pending = true; o[c] = gload(oc + c); }
            if (!__any(pending)) break;
            ++repolls;
            if (give_up(p, sg)) return;
        }
    }
    const unsigned long long k1 = p.prof ? shader_clock() : 0ull;
    x = wave_sum(x);
    const double g = fast_sigmoid_of_minus(x);                                            // .pyx:263
    double pw1, pw2;
    adam_powers(p, t, pw1, pw2);
    double c1 = (double)o[0].v + (double)o[1].v, c2 = (double)o[2].v + (double)o[3].v;
    const double step = hot_adapt(p, g, pw1, pw2, c1, c2);                                // item i on lane 0, item j on lane 1 (.pyx:267-268)
    const double gi = __shfl(step, 0), gj = __shfl(step, 1);
    // the two rows move (.pyx:271-309): one write-through store per cell, value and tag together
#pragma unroll
    for (int r = 0; r < FLOW_REGS; ++r) {
        if (lane + 64 * r < L) {
            if (k.s[r] != i) gstore(p.G + packed_cell(i, k.s[r]), (float)cell_plus((double)k.ga[r].v, p.lr, gi, p.li_reg), my_tag);
            if (k.s[r] != j) gstore(p.G + packed_cell(j, k.s[r]), (float)cell_minus((double)k.gb[r].v, p.lr, gj, p.lj_reg), my_tag);
        }
    }
    if (lane < 2 && adaptive) {
        const float h1 = (float)c1;
        gstore(oc, h1, my_tag);
        gstore(oc + 1, (float)(c1 - (double)h1), my_tag);
        if (adam) {
            const float h2 = (float)c2;
            gstore(oc + 2, h2, my_tag);
            gstore(oc + 3, (float)(c2 - (double)h2), my_tag);
        }
    }
    for (int b0 = FLOW_BLOCK; b0 < L; b0 += FLOW_BLOCK) {
        // (nobody can have written these cells since they were read above: a later step waits for THIS step's tag on them)
        SymBlock m;
        sym_fetch(p, e, cp, b0, lane, false, m, repolls);
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            if (b0 + lane + 64 * r < L) {
                if (m.s[r] != i) gstore(p.G + packed_cell(i, m.s[r]), (float)cell_plus((double)m.ga[r].v, p.lr, gi, p.li_reg), my_tag);
                if (m.s[r] != j) gstore(p.G + packed_cell(j, m.s[r]), (float)cell_minus((double)m.gb[r].v, p.lr, gj, p.lj_reg), my_tag);
            }
        }
    }
    if (lane == 0) atomicAdd(&p.loss_slots[t & (LOSS_SLOTS - 1)], x * x);
    if (p.prof && lane == 0) {       // steps | descriptor .. all tags there | the rest | polling rounds that found a tag missing
        atomicAdd(&p.prof[0], 1ull);
        atomicAdd(&p.prof[1], k1 - k0);
        atomicAdd(&p.prof[2], shader_clock() - k1);
        atomicAdd(&p.prof[3], (unsigned long long)repolls);
    }
}

// A step with a LONG profile (more than FLOW_BLOCK entries) by a whole workgroup: every wavefront takes a block, so the granules
// of up to 4 096 entries are in flight together and stay in registers for the stores.  One wavefront would fetch them block after
// block, twice -- and steps with long profiles touch the most cells, so they sit on the chain of the epoch more often than
// their 14 % share of the steps: with them at one round trip per block the chain of the ML-20M shape weighs 13.5 ms, without 7.8 ms
// (scratch: chain2.c on a stream of the bench's epoch).  Every wavefront adds the sixteen partial sums in the same order and does
// the (cheap) scalar part itself: one barrier per step.
__device__ __forceinline__ bool sym_step_wide(const SlimParams<double> &p, const int t, const int lane, const int wave, double *s_x, int *s_bad) {
    const StepDesc e = p.desc[t];
    const int i = e.i, j = e.j, L = e.L;
    const long long cp = (long long)(((unsigned long long)(unsigned)e.c << 32) | (unsigned)e.t);
    const unsigned long long k0 = p.prof ? shader_clock() : 0ull;
    unsigned repolls = 0;
    const unsigned my_tag = p.tag_base + (unsigned)t + 1u;
    const bool adaptive = p.sgd_mode != MI355REC_SGD, adam = p.sgd_mode == MI355REC_ADAM;
    Granule *oc = p.oc + 4 * (size_t)(lane == 1 ? j : i);
    const int ip = lane < 2 && adaptive ? (lane ? e.b : e.a) : -1;
    Granule o[4] = {{0.f, 0u}, {0.f, 0u}, {0.f, 0u}, {0.f, 0u}};
    if (lane < 2 && adaptive) {
        o[0] = gload(oc);
        o[1] = gload(oc + 1);
        if (adam) {
            o[2] = gload(oc + 2);
            o[3] = gload(oc + 3);
        }
    }
    constexpr int ROUND = FLOW_WAVES * FLOW_BLOCK;
    const int first = wave * FLOW_BLOCK;
    bool ok = true;
    SymBlock k;
    double x = 0.0;
    if (first < L) {
        ok = sym_fetch(p, e, cp, first, lane, true, k, repolls);
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r)
            if (first + lane + 64 * r < L) x += (double)k.ga[r].v - (double)k.gb[r].v;
    }
    for (int b0 = first + ROUND; ok && b0 < L; b0 += ROUND) {                             // profiles longer than 4 096
        SymBlock m;
        ok = sym_fetch(p, e, cp, b0, lane, true, m, repolls);
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r)
            if (b0 + lane + 64 * r < L) x += (double)m.ga[r].v - (double)m.gb[r].v;
    }
    if (ok) {
        SpinGuard sg;
        for (;;) {
            bool pending = false;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if ((c < 2 || adam) && !tag_ok(ip, o[c].tag, p.tag_base)) { pending = true; o[c] = gload(oc + c); }
            if (!__any(pending)) break;
            ++repolls;
            if (give_up(p, sg)) { ok = false; break; }
        }
    }
    x = wave_sum(x);
    if (lane == 0) {
        s_x[wave] = x;
        if (!ok) *s_bad = 1;
    }
    __syncthreads();
    if (*s_bad) return false;                                                             // (the same answer in every wavefront)
    const unsigned long long k1 = p.prof ? shader_clock() : 0ull;
    x = 0.0;
#pragma unroll
    for (int w = 0; w < FLOW_WAVES; ++w) x += s_x[w];
    const double g = fast_sigmoid_of_minus(x);
    double pw1, pw2;
    adam_powers(p, t, pw1, pw2);
    double c1 = (double)o[0].v + (double)o[1].v, c2 = (double)o[2].v + (double)o[3].v;
    const double step = hot_adapt(p, g, pw1, pw2, c1, c2);
    const double gi = __shfl(step, 0), gj = __shfl(step, 1);
    if (first < L) {
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            if (first + lane + 64 * r < L) {
                if (k.s[r] != i) gstore(p.G + packed_cell(i, k.s[r]), (float)cell_plus((double)k.ga[r].v, p.lr, gi, p.li_reg), my_tag);
                if (k.s[r] != j) gstore(p.G + packed_cell(j, k.s[r]), (float)cell_minus((double)k.gb[r].v, p.lr, gj, p.lj_reg), my_tag);
            }
        }
    }
    if (wave == 0 && lane < 2 && adaptive) {
        const float h1 = (float)c1;
        gstore(oc, h1, my_tag);
        gstore(oc + 1, (float)(c1 - (double)h1), my_tag);
        if (adam) {
            const float h2 = (float)c2;
            gstore(oc + 2, h2, my_tag);
            gstore(oc + 3, (float)(c2 - (double)h2), my_tag);
        }
    }
    for (int b0 = first + ROUND; b0 < L; b0 += ROUND) {
        SymBlock m;
        sym_fetch(p, e, cp, b0, lane, false, m, repolls);
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            if (b0 + lane + 64 * r < L) {
                if (m.s[r] != i) gstore(p.G + packed_cell(i, m.s[r]), (float)cell_plus((double)m.ga[r].v, p.lr, gi, p.li_reg), my_tag);
                if (m.s[r] != j) gstore(p.G + packed_cell(j, m.s[r]), (float)cell_minus((double)m.gb[r].v, p.lr, gj, p.lj_reg), my_tag);
            }
        }
    }
    if (wave == 0 && lane == 0) {
        atomicAdd(&p.loss_slots[t & (LOSS_SLOTS - 1)], x * x);
        if (p.prof) {                // long profiles | claim .. barrier passed | the rest | polling rounds of wavefront 0
            atomicAdd(&p.prof[4], 1ull);
            atomicAdd(&p.prof[5], k1 - k0);
            atomicAdd(&p.prof[6], shader_clock() - k1);
            atomicAdd(&p.prof[7], (unsigned long long)repolls);
        }
    }
    return true;
}

// Workgroups 0 .. long_wgs - 1 take the steps with long profiles, one step per workgroup, the others (and those once that queue is
// empty) the steps with short profiles, one per wavefront.  Both queues hand out steps in stream order and every workgroup of
// the grid is resident: the oldest step that has not run is either running or the next one of its queue, and the consumers of
// that queue that are busy are busy with older steps.
__global__ __launch_bounds__(FLOW_THREADS) void slim_sym_flow_kernel(const SlimParams<double> p, const int long_wgs) {
    __shared__ LocalQueue s_queue;
    __shared__ double s_x[FLOW_WAVES];
    __shared__ int s_next, s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_queue.next = 0; s_queue.ready = 0; s_bad = 0; }
    __syncthreads();
    if ((int)blockIdx.x < long_wgs) {
        const int n_long = p.n_steps - p.n_short;
        for (;;) {
            if (tid == 0) s_next = aload(&p.queue[1]) ? NO_STEP : atomicAdd(&p.queue[2], 1);
            __syncthreads();
            const int q = s_next;
            __syncthreads();
            if (q >= n_long) break;
            if (!sym_step_wide(p, p.order[p.n_steps - 1 - q], lane, wave, s_x, &s_bad)) break;
        }
    }
    for (;;) {          // in-order queue: everything a step can wait for is already running
        const int q = claim_step<false>(p, &s_queue, lane);
        if (q >= p.n_short) break;
        sym_step(p, p.order[q], lane);
    }
}

// Fallback (symmetric store with more than 92 681 items: packed cell ids no longer fit the 32-bit sort key): one workgroup runs
// the steps one after the other (plain accesses: one compute unit, one L1).
__global__ __launch_bounds__(1024) void slim_ordered_kernel(const SlimParams<double> p) {
    __shared__ double s_part[16];
    __shared__ double s_g[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = 0; t < p.n_steps; ++t) {
        const int u = p.su[t], i = p.si[t], j = p.sj[t];
        const int rs = p.indptr[u], re = p.indptr[u + 1];
        double x = 0.0;
        for (int q = rs + tid; q < re; q += 1024) {
            const int s = p.indices[q];
            x += (double)p.G[packed_cell(i, s)].v - (double)p.G[packed_cell(j, s)].v;
        }
        x = wave_sum(x);
        if (lane == 0) s_part[wave] = x;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < 16; ++w) tot += s_part[w];
            const double g = sigmoid_of_minus(tot);
            double pw1, pw2;
            adam_powers(p, t, pw1, pw2);
            for (int e = 0; e < 2; ++e) {                                 // item i first, then j (.pyx:267-268)
                Granule *oc = p.oc + 4 * (size_t)(e ? j : i);
                double c1 = (double)oc[0].v + (double)oc[1].v, c2 = (double)oc[2].v + (double)oc[3].v;
                s_g[e] = slim_adapt_cells(p, g, pw1, pw2, c1, c2);
                const float h1 = (float)c1, h2 = (float)c2;
                oc[0].v = h1; oc[1].v = (float)(c1 - (double)h1);
                oc[2].v = h2; oc[3].v = (float)(c2 - (double)h2);
            }
            p.loss_slots[t & (LOSS_SLOTS - 1)] += tot * tot;
        }
        __syncthreads();
        const double gi = s_g[0], gj = s_g[1];
        for (int q = rs + tid; q < re; q += 1024) {
            const int s = p.indices[q];
            if (s != i) {
                Granule *c = &p.G[packed_cell(i, s)];
                c->v = (float)cell_plus((double)c->v, p.lr, gi, p.li_reg);
            }
            if (s != j) {
                Granule *c = &p.G[packed_cell(j, s)];
                c->v = (float)cell_minus((double)c->v, p.lr, gj, p.lj_reg);
            }
        }
        __threadfence_block();       // the next step of this workgroup must read what this one wrote
        __syncthreads();
    }
}

// get_S (.pyx:343-391): row r of S with the diagonal zeroed (symmetric store mirrored), then the per-row top-K.
template <class T, int THREADS>
__global__ __launch_bounds__(THREADS) void slim_topk_kernel(const SlimParams<T> p, int topK, int n_pad, int *out_idx,
                                                            float *out_val) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *acc = smem;
    uint32_t *aux = reinterpret_cast<uint32_t *>(smem + n_pad);
    __shared__ SelectScratch sc;
    __shared__ uint32_t s_npos, s_nneg, s_ncand;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int r = blockIdx.x; r < p.n_items; r += gridDim.x) {
        if (tid == 0) { s_npos = 0; s_nneg = 0; s_ncand = 0; }
        __syncthreads();
        uint32_t npos = 0, nneg = 0;
        for (int c = tid; c < p.n_items; c += THREADS) {
            const float v = c == r ? 0.f : stored_value(p, r, c);
            acc[c] = v;
            npos += v > 0.f;
            nneg += v < 0.f;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            npos += __shfl_down(npos, off);
            nneg += __shfl_down(nneg, off);
        }
        if (lane == 0) {
            if (npos) atomicAdd(&s_npos, npos);
            if (nneg) atomicAdd(&s_nneg, nneg);
        }
        __syncthreads();
        // symmetric store: Triangular_Matrix.get_scipy_csr ranks the FULL row (zeros compete, :1384-1404);
        // dense store: similarityMatrixTopK ranks the non-zero cells only (Base/Recommender_utils.py:100-104)
        block_topk_emit<THREADS>(acc, p.n_items, topK, s_npos, s_nneg, p.symmetric ? TOPK_ZEROS_COMPETE : TOPK_NONZERO, aux, sc, &s_ncand,
                                 out_idx + (size_t)r * topK, out_val + (size_t)r * topK);
        __syncthreads();
    }
}

template <class T>
__global__ void slim_dense_kernel(const SlimParams<T> p, float *out) {
    const size_t n = (size_t)p.n_items;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n * n; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / n), c = (int)(e % n);
        out[e] = r == c ? 0.f : stored_value(p, r, c);
    }
}

// ---- the sparse-tree store's semantics on the dense array (Sparse_Matrix_Tree_CSR, .pyx:582-1030) --------------------------
// A cell "has a node" once add_value has written it.  Cells without a node hold the bit pattern of -0.0: it reads as zero,
// any update a + lr * g of a step turns it into an ordinary value, and no arithmetic of the epoch produces it again (an
// update that is exactly -0.0 would; that needs a gradient that underflowed to zero).
constexpr unsigned long long NO_NODE = 0x8000000000000000ull;
constexpr int PRUNE_THREADS = 256;

__global__ __launch_bounds__(256) void slim_no_nodes_kernel(unsigned long long *S, size_t n_cells) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n_cells; e += (size_t)gridDim.x * blockDim.x) S[e] = NO_NODE;
}

// unsigned key in the order of the doubles
__device__ __forceinline__ unsigned long long order_key(unsigned long long bits) {
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
}

// topK_selection_from_list on every row (.pyx:957-1030), as rebalance_tree(TopK) :785-805 and get_scipy_csr(TopK) :740-780 apply
// it: a row with fewer than TopK nodes is left alone, otherwise the TopK largest values stay; among equal values the HIGHER
// columns stay (glibc's qsort is a stable merge sort and compare_struct_on_data :553-568 never answers "equal", so ties keep
// their column order and the last TopK of the sorted array are taken).  Dropped nodes are freed: no node, value zero.
//
// One workgroup per row.  The row is streamed once to count its nodes (most of the work: n_items^2 * 8 bytes per call, HBM
// bound).  A row that has to be cut is read again (from L2) and its nodes are packed, in column order, into LDS, where an
// 8-bit radix select finds the TopK-th value; rows with more than PRUNE_CAP nodes run the same select over the row itself.
// The select stops as soon as the bucket holding the TopK-th value is wanted whole.
constexpr int PRUNE_CAP = 2048;

struct PruneShared {
    unsigned hist[256];
    unsigned wave_count[PRUNE_THREADS / 64];
    unsigned keep, bucket;
    unsigned long long prefix;
    unsigned long long keys[PRUNE_CAP];
    int cols[PRUNE_CAP];
};

// the row itself as the select's source: position = column
struct RowSource {
    unsigned long long *row;
    int n;
    __device__ __forceinline__ int size() const { return n; }
    __device__ __forceinline__ bool key(int at, unsigned long long &k) const {
        const unsigned long long b = row[at];
        k = order_key(b);
        return b != NO_NODE;
    }
    __device__ __forceinline__ void drop(int at) const { row[at] = NO_NODE; }
};
// the packed nodes in LDS (column order)
struct PackedSource {
    unsigned long long *row;
    const unsigned long long *keys;
    const int *cols;
    int len;
    __device__ __forceinline__ int size() const { return len; }
    __device__ __forceinline__ bool key(int at, unsigned long long &k) const { k = keys[at]; return true; }
    __device__ __forceinline__ void drop(int at) const { row[cols[at]] = NO_NODE; }
};

template <class Src>
__device__ __forceinline__ void select_and_drop(const Src src, const int topK, PruneShared &sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int size = src.size();
    // radix select, most significant byte first: after a pass the wanted value's leading bytes are `prefix`, `keep` of the
    // keys that share them stay (all keys above them stay anyway)
    unsigned keep = (unsigned)topK, bucket = 0;
    unsigned long long prefix = 0;
    int shift = 64;
    while (shift > 0) {
        shift -= 8;
        sh.hist[tid] = 0;                                           // (PRUNE_THREADS == 256)
        __syncthreads();
        for (int at = tid; at < size; at += PRUNE_THREADS) {
            unsigned long long k;
            if (!src.key(at, k)) continue;
            if (shift == 56 || (k >> (shift + 8)) == prefix) atomicAdd(&sh.hist[(unsigned)(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (wave == 0) {                                            // lane l owns digits 4l .. 4l+3; suffix sums from the top digit down
            const unsigned h0 = sh.hist[4 * lane], h1 = sh.hist[4 * lane + 1], h2 = sh.hist[4 * lane + 2], h3 = sh.hist[4 * lane + 3];
            const unsigned own = h0 + h1 + h2 + h3;
            unsigned incl = own;                                    // sum over lanes >= this one
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_down(incl, off);
                if (lane + off < 64) incl += v;
            }
            const unsigned long long reach = __ballot(incl >= keep);
            const int owner = 63 - __builtin_clzll(reach);          // the highest lane whose suffix reaches `keep`
            if (lane == owner) {
                unsigned above = incl - own;                        // keys in higher digits
                int d = 3;
                unsigned hd = h3;
                if (above + hd < keep) { above += hd; d = 2; hd = h2; }
                if (d == 2 && above + hd < keep) { above += hd; d = 1; hd = h1; }
                if (d == 1 && above + hd < keep) { above += hd; d = 0; hd = h0; }
                sh.prefix = (prefix << 8) | (unsigned long long)(4 * lane + d);
                sh.keep = keep - above;
                sh.bucket = hd;
            }
        }
        __syncthreads();
        prefix = sh.prefix;
        keep = sh.keep;
        bucket = sh.bucket;
        if (bucket == keep) break;                                  // the whole bucket stays: nothing left to split
    }
    // keys whose leading bytes are below `prefix` go; of the `bucket` keys equal to it the `keep` highest columns stay
    const unsigned drop_ties = bucket - keep;                       // > 0 only after all 8 passes: equal VALUES
    unsigned ties_before = 0;                                       // ties in lower columns (only tracked when some must go)
    for (int at0 = 0; at0 < size; at0 += PRUNE_THREADS) {
        const int at = at0 + tid;
        bool tie = false, drop = false;
        if (at < size) {
            unsigned long long k;
            if (src.key(at, k)) {
                k >>= shift;
                drop = k < prefix;
                tie = k == prefix;
            }
        }
        if (drop_ties) {
            const unsigned long long m = __ballot(tie);
            __syncthreads();
            if (lane == 0) sh.wave_count[wave] = (unsigned)__builtin_popcountll(m);
            __syncthreads();
            unsigned before = ties_before, total = 0;
#pragma unroll
            for (int w = 0; w < PRUNE_THREADS / 64; ++w) {
                if (w < wave) before += sh.wave_count[w];
                total += sh.wave_count[w];
            }
            before += (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (tie && before < drop_ties) drop = true;             // the first (lowest-column) `drop_ties` ties go
            ties_before += total;
        }
        if (drop) src.drop(at);
    }
    __syncthreads();
}

// with_diag: get_S gives the diagonal a node holding zero first (.pyx:350-351).
__global__ __launch_bounds__(PRUNE_THREADS) void slim_prune_kernel(unsigned long long *S, int n, int topK, int with_diag) {
    __shared__ PruneShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // every wavefront owns one contiguous quarter of the row (whole 64-column groups)
    const int seg = ((n + PRUNE_THREADS - 1) / PRUNE_THREADS) * 64;
    const int c_begin = min(wave * seg, n), c_end = min(c_begin + seg, n);
    for (int r = blockIdx.x; r < n; r += gridDim.x) {
        unsigned long long *row = S + (size_t)r * n;
        if (with_diag && tid == 0) row[r] = 0ull;               // a node holding +0.0
        __syncthreads();
        unsigned mine = 0;
        int c = c_begin + lane;
        for (; c + 192 < c_end; c += 256) {                     // four independent loads in flight per lane
            const unsigned long long b0 = row[c], b1 = row[c + 64], b2 = row[c + 128], b3 = row[c + 192];
            mine += (b0 != NO_NODE) + (b1 != NO_NODE) + (b2 != NO_NODE) + (b3 != NO_NODE);
        }
        for (; c < c_end; c += 64) mine += row[c] != NO_NODE;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) sh.wave_count[wave] = mine;
        __syncthreads();
        unsigned len = 0, base = 0;
#pragma unroll
        for (int w = 0; w < PRUNE_THREADS / 64; ++w) {
            if (w < wave) base += sh.wave_count[w];
            len += sh.wave_count[w];
        }
        __syncthreads();
        if (topK <= 0 || len <= (unsigned)topK) continue;        // (len == TopK: the selection keeps everything)
        if (len <= (unsigned)PRUNE_CAP) {
            // pack (key, column) in column order: wavefront w writes from `base`, lanes by ballot rank
            for (int c0 = c_begin; c0 < c_end; c0 += 64) {
                const int cc = c0 + lane;
                const unsigned long long b = cc < c_end ? row[cc] : NO_NODE;
                const bool node = b != NO_NODE;
                const unsigned long long m = __ballot(node);
                if (node) {
                    const unsigned at = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    sh.keys[at] = order_key(b);
                    sh.cols[at] = cc;
                }
                base += (unsigned)__builtin_popcountll(m);
            }
            __syncthreads();
            select_and_drop(PackedSource{row, sh.keys, sh.cols, (int)len}, topK, sh);
        } else {
            select_and_drop(RowSource{row, n}, topK, sh);
        }
    }
}

// from_linked_list_to_python_list (.pyx:862-875) for every row after the selection: the non-zero nodes in column order.
__global__ __launch_bounds__(PRUNE_THREADS) void slim_list_kernel(const unsigned long long *S, int n, int width, int *out_idx, float *out_val) {
    __shared__ unsigned s_wave[PRUNE_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = blockIdx.x; r < n; r += gridDim.x) {
        const unsigned long long *row = S + (size_t)r * n;
        unsigned at = 0;
        for (int c0 = 0; c0 < n; c0 += PRUNE_THREADS) {
            const int c = c0 + tid;
            double v = 0.0;
            if (c < n) {
                const unsigned long long b = row[c];
                if (b != NO_NODE) v = __longlong_as_double((long long)b);
            }
            const bool listed = v != 0.0;
            const unsigned long long m = __ballot(listed);
            if (lane == 0) s_wave[wave] = (unsigned)__builtin_popcountll(m);
            __syncthreads();
            unsigned pos = at, total = 0;
#pragma unroll
            for (int w = 0; w < PRUNE_THREADS / 64; ++w) {
                if (w < wave) pos += s_wave[w];
                total += s_wave[w];
            }
            pos += (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (listed && pos < (unsigned)width) {
                out_idx[(size_t)r * width + pos] = c;
                out_val[(size_t)r * width + pos] = (float)v;
            }
            at += total;
            __syncthreads();
        }
        for (unsigned q = min(at, (unsigned)width) + tid; q < (unsigned)width; q += PRUNE_THREADS) {
            out_idx[(size_t)r * width + q] = -1;
            out_val[(size_t)r * width + q] = 0.f;
        }
    }
}

}  // namespace
}  // namespace mi355rec

using namespace mi355rec;

// What the schedule of ONE stream of steps hands to the launch that runs it.  There are two: the sample stream of an epoch does not
// depend on S, so while the dataflow kernel of epoch e runs, epoch e + 1 is drawn and scheduled on a second HIP stream (the item
// sort, the 40 M-pair cell sort of the symmetric store: 0.35 / 3.7 ms of a 2.1 / 18.9 ms epoch at the ML-20M shape) -- also across
// calls: the reference's fit loop asks for one epoch per call.
struct StreamSet {
    DeviceBuffer<int> su, si, sj, seq, iprev, len2, item_vals, pred, run_start;
    DeviceBuffer<unsigned long long> item_keys;          // (item, step) pairs in sorted order: the owned rows' lists are runs of it
    DeviceBuffer<long long> cellptr;
    DeviceBuffer<StepDesc> desc;                         // symmetric store: one descriptor per step ...
    DeviceBuffer<int> order, n_short_dev;                // ... and the two queues: short profiles in stream order, then the long ones backwards
    int n_short = 0;
    DeviceBuffer<unsigned> item_cnt;
    size_t capacity = 0, pred_capacity = 0;
    long long epoch = -1;                                // the native epoch this set is scheduled for (-1: none / a replayed stream)
    int n = 0;
    long long n_cells = 0;
};

struct mi355rec_slim {
    mi355rec_slim_config cfg{};
    int n_users = 0, n_items = 0;
    bool f64 = false;           // arithmetic type; storage type of the dense store and its optimiser cells
    size_t nnz = 0;
    hipStream_t stream = nullptr, side = nullptr;       // `side`: schedules of the next epoch
    StreamTimer call_timer;
    DispatchTimers dispatch_timers;
    StreamSet set[2];
    int cur = 0;                                        // the set whose stream ran last
    bool last_native = false;                           // ... and whether that was an epoch of the on-device sampler
    DeviceBuffer<int> indptr, indices, ticket, queue;
    // scratch of the schedules (one schedule at a time)
    DeviceBuffer<int> vals, vals_sorted;
    DeviceBuffer<unsigned long long> keys, keys_sorted;
    DeviceBuffer<unsigned char> sched_tmp;
    size_t sort_capacity = 0;
    DeviceBuffer<unsigned char> S, c1, c2;              // dense store: S, c1, c2 are float or double by `f64`
    DeviceBuffer<Granule> G, oc;                        // symmetric store: packed triangle of granules, [n_items][4] optimiser granules
    DeviceBuffer<double> loss_slots;
    // dense store, filled at launch time on the main stream: owned rows, descriptors, the cold queue
    DeviceBuffer<int> hot_rank, hot_tables, counters, iota, item_by_cnt;     // hot_tables: item | first | length
    DeviceBuffer<StepDesc> desc, cold_desc, own_desc;
    DeviceBuffer<unsigned> cnt_sorted;
    DeviceBuffer<unsigned char> cold_flag, launch_tmp;
    DeviceBuffer<unsigned long long> mail;              // [2][launch_capacity]
    size_t launch_capacity = 0;
    DeviceBuffer<unsigned long long> prof;              // MI355REC_SLIM_PROF=1: phase clocks of the last launch
    long long steps_done = 0, epochs_done = 0;
    bool aborted = false;                               // a launch gave up on a hand-off: S holds part of an epoch, every later call refuses
    unsigned tag_base = 0;                              // symmetric store: tags handed out so far
    int last_owners = 0, last_cold = 0;                 // owned rows / steps on rows in HBM of the last dense launch (diagnostics)
    std::vector<double> h_loss;
    mi355rec_stats stats{};

    ~mi355rec_slim() {
        if (stream) (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);
        call_timer.destroy();
        dispatch_timers.destroy();
        ReleaseScope::forget(side);
        ReleaseScope::forget(stream);
        if (side) (void)hipStreamDestroy(side);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int env_int(const char *name, int fallback) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : fallback;
}

template <class T>
void fill_params(mi355rec_slim *h, const StreamSet &st, SlimParams<T> &p) {
    const auto &c = h->cfg;
    p.n_users = h->n_users; p.n_items = h->n_items; p.symmetric = c.symmetric; p.sgd_mode = c.sgd_mode;
    p.lr = (T)c.learning_rate; p.li_reg = (T)c.li_reg; p.lj_reg = (T)c.lj_reg;
    p.gamma = (T)c.gamma; p.beta_1 = (T)c.beta_1; p.beta_2 = (T)c.beta_2;
    p.one_m_gamma = (T)(1.0 - c.gamma); p.one_m_beta_1 = (T)(1.0 - c.beta_1); p.one_m_beta_2 = (T)(1.0 - c.beta_2);
    p.beta_1_d = c.beta_1; p.beta_2_d = c.beta_2;
    p.seed = c.random_seed;
    p.indptr = h->indptr.ptr; p.indices = h->indices.ptr;
    p.S = reinterpret_cast<T *>(h->S.ptr); p.c1 = reinterpret_cast<T *>(h->c1.ptr); p.c2 = reinterpret_cast<T *>(h->c2.ptr);
    p.G = h->G.ptr; p.oc = h->oc.ptr;
    p.su = st.su.ptr; p.si = st.si.ptr; p.sj = st.sj.ptr;
    p.seq = st.seq.ptr; p.iprev = st.iprev.ptr; p.cellptr = st.cellptr.ptr; p.pred = st.pred.ptr;
    p.ticket = h->ticket.ptr; p.queue = h->queue.ptr;
    p.loss_slots = h->loss_slots.ptr;
    p.epoch = h->epochs_done;
    p.steps_before = h->steps_done;
    p.n_steps = 0;
    p.tag_base = h->tag_base;
    p.hot_rank = h->hot_rank.ptr;
    p.hot_item = h->hot_tables.ptr; p.lst_begin = h->hot_tables.ptr + MAX_OWNERS; p.lst_len = h->hot_tables.ptr + 2 * MAX_OWNERS;
    p.n_hot = h->counters.ptr; p.n_cold = h->counters.ptr + 1;
    p.desc = c.symmetric ? st.desc.ptr : h->desc.ptr; p.cold_desc = h->cold_desc.ptr; p.own_desc = h->own_desc.ptr;
    p.order = st.order.ptr; p.n_short = st.n_short;
    p.nap = env_int("MI355REC_SLIM_NAP", 1);
    p.prof = h->prof.ptr;
    p.mail_x = h->mail.ptr; p.mail_g = h->mail.ptr + h->launch_capacity;
}

bool flow_supported(const mi355rec_slim *h) { return !(h->cfg.symmetric && h->n_items > 92681); }      // (cell ids are 32-bit sort keys)

// the sparse store cuts an epoch into segments with a pruning pass between them: those are scheduled one after the other
bool schedules_ahead(const mi355rec_slim *h) { return !h->cfg.train_with_sparse_weights && flow_supported(h) && !getenv("MI355REC_SLIM_NO_PRESCHED"); }

void ensure_set_capacity(mi355rec_slim *h, StreamSet &st, size_t n) {
    if (st.capacity >= n) return;
    MI_HIP(hipStreamSynchronize(h->stream));
    MI_HIP(hipStreamSynchronize(h->side));
    st.su.alloc(n); st.si.alloc(n); st.sj.alloc(n);
    st.seq.alloc(2 * n); st.iprev.alloc(2 * n); st.len2.alloc(n + 1); st.cellptr.alloc(n + 1);
    st.item_keys.alloc(std::max<size_t>(2 * n, 1024)); st.item_vals.alloc(std::max<size_t>(2 * n, 1024));
    if (h->cfg.symmetric) {
        st.desc.alloc(n);
        st.order.alloc(n);
        if (!st.n_short_dev.ptr) st.n_short_dev.alloc(2);      // [1]: first step with a seen negative item
    }
    if (!st.run_start.ptr) {
        st.run_start.alloc((size_t)h->n_items);
        st.item_cnt.alloc((size_t)h->n_items);
    }
    st.capacity = n;
    st.epoch = -1;
    st.n = 0;
}

void ensure_launch_capacity(mi355rec_slim *h, size_t n) {      // dense store
    if (h->cfg.symmetric || h->launch_capacity >= n) return;
    MI_HIP(hipStreamSynchronize(h->stream));
    h->desc.alloc(n);
    h->cold_desc.alloc(n);
    h->own_desc.alloc(2 * n);
    h->cold_flag.alloc(n);
    h->mail.alloc(2 * n);
    h->launch_capacity = n;
}

void ensure_tmp(DeviceBuffer<unsigned char> &tmp, size_t bytes, hipStream_t s) {
    if (tmp.count < bytes) {
        MI_HIP(hipStreamSynchronize(s));                // (earlier work of the stream may still use the old block)
        tmp.alloc(bytes + (bytes >> 2) + 256);
    }
}

void ensure_sort_capacity(mi355rec_slim *h, size_t n, hipStream_t s) {
    if (h->sort_capacity >= n) return;
    MI_HIP(hipStreamSynchronize(s));
    h->keys.alloc(n); h->vals.alloc(n);
    if (h->cfg.symmetric) { h->keys_sorted.alloc(n); h->vals_sorted.alloc(n); }      // (the item pass sorts into the set's own arrays)
    h->sort_capacity = n;
}

int bits_for(unsigned long long n_values) {
    int b = 1;
    while (b < 63 && (1ull << b) < n_values) ++b;
    return b;
}

void sort_pairs(mi355rec_slim *h, unsigned long long *keys_out, int *vals_out, size_t n, int end_bit, hipStream_t s) {
    size_t bytes = 0;
    MI_HIP(rocprim::radix_sort_pairs(nullptr, bytes, h->keys.ptr, keys_out, h->vals.ptr, vals_out, std::max(n, h->sort_capacity), 0, end_bit, s));
    ensure_tmp(h->sched_tmp, bytes, s);                 // (sized for the scratch arrays' capacity: it does not grow from epoch to epoch)
    bytes = h->sched_tmp.count;
    MI_HIP(rocprim::radix_sort_pairs(h->sched_tmp.ptr, bytes, h->keys.ptr, keys_out, h->vals.ptr, vals_out, n, 0, end_bit, s));
}

// Owned rows need a LEASE on compute units.  An owner's steps are static (its row's list), so every owner has to be resident
// together with at least one workgroup that runs the other steps.  That holds for launches whose workgroups all fit the device
// together (one workgroup per compute unit: a row takes most of its LDS) -- and not for persistent kernels that compete for the
// compute units, where the workgroups of each would wait for owners that the others keep out.  The pool below has one slot per
// compute unit; a launch takes what it is allowed (all of them, or MI355REC_SLIM_CUS of them when several models train side by
// side) and sizes its grid by what it got.  A launch that gets fewer than 32 slots runs every step from the in-order queue, which
// makes progress under any residency.
// ... of ONE process: workgroups of another process's persistent kernel are as much in the owners' way as another thread's.  The
// reference's search trains its candidates in a multiprocessing.Pool (ParameterTuning/run_parameter_search.py:498-503): two fits on
// one GPU each used to launch one owner workgroup per compute unit, neither set became resident, and both aborted after the 5 s spin
// budget with S half-updated.  Owners therefore also need the device's OWNER GATE: an advisory file lock (flock, released by the
// kernel when its holder dies) on a per-user, per-device file; the process that holds it runs owners, everybody else runs every
// step from the in-order queue -- slower, never stuck.  Taken with the first lease of a process, given back with the last.
struct OwnerGate {
    std::mutex lock;
    int fd = -1, holders = 0;
    // The device's lock file: per user and per PCI bus id, in MI355REC_LOCK_DIR, else XDG_RUNTIME_DIR (a directory only the user can
    // write), else /tmp; never through a symbolic link (the name is predictable).  -1: no lock file.
    static int open_lock_file() {
        int dev = 0;
        char bus[64] = "unknown";
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev);
        for (char *c = bus; *c; ++c)
            if (*c == ':' || *c == '.' || *c == '/') *c = '_';
        const char *dir = getenv("MI355REC_LOCK_DIR");
        if (!dir || !*dir) dir = getenv("XDG_RUNTIME_DIR");
        if (!dir || !*dir) dir = "/tmp";
        char path[512];
        snprintf(path, sizeof(path), "%s/mi355rec_slim_owners_%u_%s.lock", dir, (unsigned)getuid(), bus);
        return open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
    }
    bool acquire() {
        std::lock_guard<std::mutex> g(lock);
        if (holders > 0) { ++holders; return true; }
        if (getenv("MI355REC_SLIM_NO_OWNER_GATE")) { ++holders; return true; }
        const int f = open_lock_file();
        if (f < 0) return false;                       // no lock file, no owners: the queue-only mode is always correct
        if (flock(f, LOCK_EX | LOCK_NB) != 0) {
            close(f);
            return false;                              // another process trains with owners on this device
        }
        fd = f;
        holders = 1;
        return true;
    }
    // Blocking variant for the symmetric store's launch, which has no queue-only mode to fall back to: its long-profile workgroups wait
    // for steps that only its short-profile workgroups run, so TWO such kernels on one device (two handles in threads, two processes)
    // may leave each other's short workgroups non-resident and spin into the 5 s budget.  Symmetric launches of a device
    // therefore run one at a time: `serial` inside the process, the file lock across processes (held from launch to finish).
    // The file lock is polled (LOCK_NB) OUTSIDE the gate's mutex against a deadline (MI355REC_SLIM_GATE_WAIT_S, default 600 s): a
    // process that hangs with the lock held makes this launch fail with a message, not block for ever -- and never blocks the
    // acquire / release of the dense store's launches of this process.
    std::mutex serial;
    bool acquire_blocking() {
        serial.lock();
        {
            std::lock_guard<std::mutex> g(lock);
            if (holders > 0 || getenv("MI355REC_SLIM_NO_OWNER_GATE")) { ++holders; return true; }
        }
        const int f = open_lock_file();
        bool locked = false;
        if (f >= 0) {
            const double wait_s = getenv("MI355REC_SLIM_GATE_WAIT_S") ? atof(getenv("MI355REC_SLIM_GATE_WAIT_S")) : 600.0;
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(wait_s);
            for (;;) {
                if (flock(f, LOCK_EX | LOCK_NB) == 0) { locked = true; break; }
                if (errno != EWOULDBLOCK && errno != EINTR) break;          // (no usable lock file: in-process serialisation only)
                if (std::chrono::steady_clock::now() >= deadline) {
                    close(f);
                    serial.unlock();
                    fail(MI355REC_E_HIP, "SLIM-BPR (symmetric store): another process has held this device's launch lock for more than %.0f s", wait_s);
                }
                usleep(2000);
            }
            if (!locked) close(f);
        }
        std::lock_guard<std::mutex> g(lock);
        if (holders > 0) {                                 // (a dense launch of this process took the gate meanwhile: its lock serves)
            if (locked) { (void)flock(f, LOCK_UN); close(f); }
            ++holders;
            return true;
        }
        if (locked) fd = f;
        holders = 1;
        return true;
    }
    void release_blocking() {
        release();
        serial.unlock();
    }
    void release() {
        std::lock_guard<std::mutex> g(lock);
        if (holders > 0 && --holders == 0 && fd >= 0) {
            (void)flock(fd, LOCK_UN);
            close(fd);
            fd = -1;
        }
    }
};
OwnerGate &owner_gate() {
    static OwnerGate *g = new OwnerGate();
    return *g;
}

std::atomic<int> g_owner_slots{-1};
struct OwnerLease {
    int slots = 0;
    bool gated = false;
    void take(bool wanted, int want) {
        if (!wanted) return;
        if (!owner_gate().acquire()) return;           // (slots stays 0: queue-only launch)
        gated = true;
        int expected = -1;
        g_owner_slots.compare_exchange_strong(expected, multiprocessor_count());      // first use: one slot per compute unit
        int have = g_owner_slots.load();
        while (have >= 32) {
            const int n = std::min(have, want);
            if (g_owner_slots.compare_exchange_weak(have, have - n)) { slots = n; return; }
        }
    }
    void give_back() {
        if (slots) g_owner_slots.fetch_add(slots);
        slots = 0;
        if (gated) owner_gate().release();
        gated = false;
    }
    ~OwnerLease() { give_back(); }
};

template <class Kernel>
int blocks_per_cu(Kernel k, size_t lds) {
    int per_cu = 0;
    MI_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, FLOW_THREADS, lds));
    return std::max(1, std::min(per_cu, 2));            // (2048 threads per compute unit)
}

DepParams dep_params(mi355rec_slim *h, StreamSet &st, int n, int first) {
    DepParams d{};
    d.n_steps = n; d.n_items = h->n_items;
    d.indptr = h->indptr.ptr; d.indices = h->indices.ptr; d.su = st.su.ptr + first; d.si = st.si.ptr + first; d.sj = st.sj.ptr + first;
    d.keys = h->keys.ptr; d.vals = h->vals.ptr; d.keys_sorted = st.item_keys.ptr; d.vals_sorted = st.item_vals.ptr;
    d.seq = st.seq.ptr; d.iprev = st.iprev.ptr; d.len2 = st.len2.ptr; d.cellptr = st.cellptr.ptr; d.pred = st.pred.ptr;
    d.run_start = st.run_start.ptr; d.item_cnt = st.item_cnt.ptr; d.cnt_sorted = h->cnt_sorted.ptr; d.item_by_cnt = h->item_by_cnt.ptr;
    d.hot_rank = h->hot_rank.ptr;
    d.hot_item = h->hot_tables.ptr; d.lst_begin = h->hot_tables.ptr + MAX_OWNERS; d.lst_len = h->hot_tables.ptr + 2 * MAX_OWNERS;
    d.n_hot = h->counters.ptr;
    d.cold_flag = h->cold_flag.ptr;
    d.desc = h->cfg.symmetric ? st.desc.ptr : h->desc.ptr; d.own_desc = h->own_desc.ptr;
    return d;
}

struct ShortProfile {
    const int *len2;
    __device__ bool operator()(const int t) const { return len2[t] <= 2 * FLOW_BLOCK; }
};

template <class T>
void draw_epoch(mi355rec_slim *h, StreamSet &st, long long epoch, int n, hipStream_t s) {
    SlimParams<T> p{};
    fill_params(h, st, p);
    p.epoch = epoch;
    p.n_steps = n;
    hipLaunchKernelGGL(slim_sample_kernel<T>, dim3(div_up(n, 256)), dim3(256), 0, s, p, st.su.ptr, st.si.ptr, st.sj.ptr);
    st.epoch = -1;
    st.n = 0;
}

// Everything about steps first .. first + n - 1 of st.su / si / sj that does not depend on S, on stream `s`: ticket numbers and
// previous steps per item (a sort of the 2n (item, step) pairs), cell slots, and -- symmetric store -- the last writer of every
// cell (a sort of the (cell, step) pairs) and the step descriptors.  Returns with `s` drained (the number of cells sizes the sort).
void schedule_stream(mi355rec_slim *h, StreamSet &st, int n, int first, hipStream_t s) {
    if (!flow_supported(h)) {
        st.n = n;
        st.n_cells = 0;
        return;
    }
    ensure_sort_capacity(h, std::max<size_t>(2 * (size_t)n, 1024), s);
    DepParams d = dep_params(h, st, n, first);
    MI_HIP(hipMemsetAsync(st.item_cnt.ptr, 0, sizeof(unsigned) * (size_t)h->n_items, s));
    hipLaunchKernelGGL(slim_item_keys_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, d);
    sort_pairs(h, st.item_keys.ptr, st.item_vals.ptr, 2 * (size_t)n, 32 + bits_for((unsigned long long)h->n_items), s);
    hipLaunchKernelGGL(slim_seq_kernel, dim3(div_up(2 * n, 256)), dim3(256), 0, s, d);
    // profile lengths -> cell slots (also the algorithmic byte count of the call)
    MI_HIP(hipMemsetAsync(st.len2.ptr + n, 0, sizeof(int), s));
    size_t bytes = 0;
    MI_HIP(rocprim::exclusive_scan(nullptr, bytes, st.len2.ptr, st.cellptr.ptr, 0ll, (size_t)(n + 1), rocprim::plus<long long>(), s));
    ensure_tmp(h->sched_tmp, bytes, s);
    bytes = h->sched_tmp.count;
    MI_HIP(rocprim::exclusive_scan(h->sched_tmp.ptr, bytes, st.len2.ptr, st.cellptr.ptr, 0ll, (size_t)(n + 1), rocprim::plus<long long>(), s));
    long long n_cells = 0;
    int n_short = n;
    if (h->cfg.symmetric) {
        // the two queues of the symmetric store's kernel: profiles of up to FLOW_BLOCK entries (a wavefront each), longer ones (a workgroup each)
        const ShortProfile is_short{st.len2.ptr};
        bytes = 0;
        MI_HIP(rocprim::partition(nullptr, bytes, rocprim::counting_iterator<int>(0), st.order.ptr, st.n_short_dev.ptr, (size_t)n, is_short, s));
        ensure_tmp(h->sched_tmp, bytes, s);
        bytes = h->sched_tmp.count;
        MI_HIP(rocprim::partition(h->sched_tmp.ptr, bytes, rocprim::counting_iterator<int>(0), st.order.ptr, st.n_short_dev.ptr, (size_t)n, is_short, s));
        MI_HIP(hipMemcpyAsync(&n_short, st.n_short_dev.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    MI_HIP(hipMemcpyAsync(&n_cells, st.cellptr.ptr + n, sizeof(long long), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    st.n_cells = n_cells;
    st.n_short = n_short;
    if (h->cfg.symmetric) {
        // per cell the step that touched it last: sort the (cell, step) pairs of the stream
        MI_REQUIRE(n_cells < (1ll << 31), "stream too long for the cell sort (%lld cells)", n_cells);
        // (a buffer that grows is handed back to the block cache, which waits for the device -- and so for the dataflow kernel this
        // schedule is meant to run behind: sized once, 25 % above the expected 2 nnz n / n_users cells of n uniformly drawn users)
        const size_t roomy = std::max((size_t)(2.5 * (double)h->nnz * (double)n / (double)h->n_users) + 1024, (size_t)n_cells + (size_t)(n_cells >> 2));
        if (h->sort_capacity < (size_t)n_cells) ensure_sort_capacity(h, roomy, s);
        if (st.pred_capacity < (size_t)n_cells) {
            MI_HIP(hipStreamSynchronize(h->stream));
            st.pred.alloc(roomy);
            st.pred_capacity = st.pred.count;
        }
        d = dep_params(h, st, n, first);
        d.keys_sorted = h->keys_sorted.ptr; d.vals_sorted = h->vals_sorted.ptr;
        d.n_cells = n_cells;
        d.step_bits = bits_for((unsigned long long)n);
        const int cell_bits = bits_for((unsigned long long)h->n_items * ((unsigned long long)h->n_items + 1) / 2 + 1);
        d.no_cell = (unsigned)((1ull << cell_bits) - 1ull);
        d.bad_step = st.n_short_dev.ptr + 1;
        MI_HIP(hipMemsetAsync(d.bad_step, 0x7F, sizeof(int), s));
        hipLaunchKernelGGL(slim_cell_keys_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, d);
        sort_pairs(h, h->keys_sorted.ptr, h->vals_sorted.ptr, (size_t)n_cells, cell_bits + d.step_bits, s);
        hipLaunchKernelGGL(slim_pred_kernel, dim3(div_up(n_cells, 256)), dim3(256), 0, s, d);
        hipLaunchKernelGGL(slim_desc_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, d, 1);
        MI_HIP(hipGetLastError());
        int bad = 0;
        MI_HIP(hipMemcpyAsync(&bad, d.bad_step, sizeof(int), hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        if (bad < n)
            fail(MI355REC_E_INVALID, "sample %d: the negative item is in the user's profile (the symmetric store holds cell (i, j) and cell (j, i) "
                 "as one cell; the reference's sampler never draws a seen item)", first + bad);
    }
    st.n = n;
}

struct Launched {
    OwnerLease lease;
    bool dense = false, profile = false, serial = false;
    hipStream_t stream = nullptr;          // set once the dataflow kernel is enqueued
    void end_serial() {
        if (serial) owner_gate().release_blocking();
        serial = false;
    }
    // (error paths end here with the kernel possibly still running: it is waited for BEFORE the gate and the leased compute units go
    // back -- finish_stream has already drained the stream on the regular path, where this costs nothing)
    ~Launched() {
        if (stream && (serial || lease.slots > 0)) (void)hipStreamSynchronize(stream);
        end_serial();
    }
};

// The dataflow kernel of a scheduled stream, enqueued on the handle's main stream.
template <class T>
void launch_stream(mi355rec_slim *h, StreamSet &st, int n, int first, Launched &L) {
    hipStream_t s = h->stream;
    L.stream = s;
    SlimParams<T> p{};
    const bool sym = h->cfg.symmetric != 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    L.profile = getenv("MI355REC_SLIM_PROF") != nullptr;
    if (L.profile) {
        if (!h->prof.ptr) h->prof.alloc(8 * (MAX_OWNERS + 1));
        MI_HIP(hipMemsetAsync(h->prof.ptr, 0, sizeof(unsigned long long) * h->prof.count, s));
    }
    MI_HIP(hipMemsetAsync(h->queue.ptr, 0, sizeof(int) * 4, s));
    if (getenv("MI355REC_SLIM_INJECT_ABORT")) {                      // (test hook: the abort flag is up before the first step polls it)
        const int one = 1;
        MI_HIP(hipMemcpyAsync(h->queue.ptr + 1, &one, sizeof(int), hipMemcpyHostToDevice, s));
        MI_HIP(hipStreamSynchronize(s));
    }
    if (!flow_supported(h)) {
        if constexpr (std::is_same<T, double>::value) {
            fill_params(h, st, p);
            p.su += first; p.si += first; p.sj += first;
            p.n_steps = n;
            h->dispatch_timers.next(e0, e1, 1 << 30);
            hipExtLaunchKernelGGL(slim_ordered_kernel, dim3(1), dim3(1024), 0, s, e0, e1, 0, p);
        }
    } else if (sym) {
        if constexpr (std::is_same<T, double>::value) {
            fill_params(h, st, p);
            p.su += first; p.si += first; p.sj += first;
            p.n_steps = n;
            h->tag_base += (unsigned)n;                  // (wraps after 4 G steps; a tag is compared only with the tag of a step of the same call)
            // steps in flight = wavefronts of the grid: more of them only adds pollers once the chain of dependent steps is the bound
            // (every workgroup has to be resident: at most what the device holds at once)
            // ... leaving some compute units to the schedule of the next epoch (the kernel is bound by its chain of dependent
            // steps, not by the number of steps in flight: 512 of them were as fast as 8 192)
            const int per_cu = blocks_per_cu(slim_sym_flow_kernel, 0);
            const int spare = schedules_ahead(h) ? std::max(0, std::min(multiprocessor_count() / 2, env_int("MI355REC_SLIM_SYM_SPARE_CUS", 64))) : 0;
            const int fit = (multiprocessor_count() - spare) * per_cu;
            const int most = std::max(2, std::min(env_int("MI355REC_SLIM_SYM_WGS", fit), fit));
            // a quarter of them for the long profiles (14 % of the steps at the ML-20M shape, a workgroup each)
            const int n_long = n - st.n_short;
            const int long_wgs = std::min(n_long, std::max(1, std::min(most - 1, env_int("MI355REC_SLIM_SYM_LONG_WGS", most / 4))));
            const int grid = long_wgs + std::max(1, std::min(div_up(st.n_short, FLOW_WAVES), most - long_wgs));
            h->dispatch_timers.next(e0, e1, 1 << 30);
            L.serial = owner_gate().acquire_blocking();              // one symmetric dataflow kernel per device at a time
            hipExtLaunchKernelGGL(slim_sym_flow_kernel, dim3(grid), dim3(FLOW_THREADS), 0, s, e0, e1, 0, p, long_wgs);
        }
    } else {
        L.dense = true;
        ensure_launch_capacity(h, st.capacity);
        DepParams d = dep_params(h, st, n, first);
        // the busiest rows of this stream get owners (if this launch gets compute units leased and a row fits the LDS)
        const size_t row_bytes = ((size_t)h->n_items * sizeof(float) + 15) & ~(size_t)15;
        auto kernel = slim_dense_flow_kernel<T>;
        int max_owners = std::min(MAX_OWNERS, env_int("MI355REC_SLIM_OWNERS", 128));
        const bool wanted = max_owners > 0 && row_bytes + 4096 <= 160 * 1024 && !h->cfg.train_with_sparse_weights;
        L.lease.take(wanted, std::max(32, std::min(multiprocessor_count(), env_int("MI355REC_SLIM_CUS", multiprocessor_count()))));
        const bool owners = L.lease.slots > 0;
        const size_t lds = owners ? row_bytes : 0;
        if (lds > 48 * 1024) MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // with owners: one workgroup per leased compute unit (they must all be resident); without: whatever fits
        const int grid = owners ? L.lease.slots : multiprocessor_count() * blocks_per_cu(kernel, 0);
        max_owners = std::min(max_owners, grid / 2);
        MI_HIP(hipMemsetAsync(h->hot_rank.ptr, 0xFF, sizeof(int) * (size_t)h->n_items, s));
        MI_HIP(hipMemsetAsync(h->counters.ptr, 0, sizeof(int) * 2, s));
        size_t bytes = 0;
        if (owners) {
            MI_HIP(rocprim::radix_sort_pairs_desc(nullptr, bytes, st.item_cnt.ptr, h->cnt_sorted.ptr, h->iota.ptr, h->item_by_cnt.ptr,
                                                  (size_t)h->n_items, 0, 32, s));
            ensure_tmp(h->launch_tmp, bytes, s);
            bytes = h->launch_tmp.count;
            MI_HIP(rocprim::radix_sort_pairs_desc(h->launch_tmp.ptr, bytes, st.item_cnt.ptr, h->cnt_sorted.ptr, h->iota.ptr, h->item_by_cnt.ptr,
                                                  (size_t)h->n_items, 0, 32, s));
            d.max_owners = max_owners;
            d.min_steps = std::max(2, env_int("MI355REC_SLIM_OWNER_MIN_STEPS", 24));
            hipLaunchKernelGGL(slim_owners_kernel, dim3(1), dim3(256), 0, s, d);
            MI_HIP(hipMemsetAsync(h->mail.ptr, 0xFF, sizeof(unsigned long long) * 2 * h->launch_capacity, s));
        }
        hipLaunchKernelGGL(slim_desc_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, d, 0);
        if (owners) hipLaunchKernelGGL(slim_owner_desc_kernel, dim3(div_up(2 * n, 256)), dim3(256), 0, s, d);
        bytes = 0;
        MI_HIP(rocprim::select(nullptr, bytes, h->desc.ptr, h->cold_flag.ptr, h->cold_desc.ptr, h->counters.ptr + 1, (size_t)n, s));
        ensure_tmp(h->launch_tmp, bytes, s);
        bytes = h->launch_tmp.count;
        MI_HIP(rocprim::select(h->launch_tmp.ptr, bytes, h->desc.ptr, h->cold_flag.ptr, h->cold_desc.ptr, h->counters.ptr + 1, (size_t)n, s));
        MI_HIP(hipMemsetAsync(h->ticket.ptr, 0, sizeof(int) * (size_t)h->n_items, s));
        fill_params(h, st, p);
        p.su += first; p.si += first; p.sj += first;
        p.n_steps = n;
        h->dispatch_timers.next(e0, e1, 1 << 30);
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(FLOW_THREADS), lds, s, e0, e1, 0, p, owners ? 1 : 0);
    }
    h->stats.n_launches += 1;
    MI_HIP(hipGetLastError());
}

// Waits for the launch, gives the leased compute units back, fails if the kernel gave up on a hand-off.
void finish_stream(mi355rec_slim *h, Launched &L, int n) {
    hipStream_t s = h->stream;
    int flags[2] = {0, 0}, counters[2] = {0, 0};
    MI_HIP(hipMemcpyAsync(flags, h->queue.ptr, sizeof(flags), hipMemcpyDeviceToHost, s));
    if (L.dense) MI_HIP(hipMemcpyAsync(counters, h->counters.ptr, sizeof(counters), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    L.lease.give_back();
    L.end_serial();
    if (L.dense) {
        h->last_owners = counters[0];
        h->last_cold = counters[1];
    }
    if (flags[1]) {
        // Steps of the epoch have been applied, others not, and nothing records which: the model is neither the one before the call nor a
        // trained one.  There is no copy to roll back to (S is n_items^2 cells), so the handle says so from now on instead of training on.
        h->aborted = true;
        fail(MI355REC_E_HIP, "SLIM-BPR: a hand-off between steps did not arrive (dataflow kernel aborted); S holds part of an epoch -- this handle "
                             "refuses further calls, create a new one");
    }
    if (L.profile) {
        std::vector<unsigned long long> c(h->prof.count);
        MI_HIP(hipMemcpy(c.data(), h->prof.ptr, sizeof(unsigned long long) * c.size(), hipMemcpyDeviceToHost));
        if (h->cfg.symmetric) {
            fprintf(stderr, "[slim prof] symmetric: %llu steps by a wavefront, mean cycles: until all tags were there %.0f, rest %.0f; %.2f extra polling rounds per step\n",
                    c[0], (double)c[1] / std::max(1ull, c[0]), (double)c[2] / std::max(1ull, c[0]), (double)c[3] / std::max(1ull, c[0]));
            fprintf(stderr, "[slim prof] symmetric: %llu steps by a workgroup, mean cycles: until all tags were there %.0f, rest %.0f; %.2f extra polling rounds per step\n",
                    c[4], (double)c[5] / std::max(1ull, c[4]), (double)c[6] / std::max(1ull, c[4]), (double)c[7] / std::max(1ull, c[4]));
        } else {
            for (int o = 0; o < h->last_owners; o += std::max(1, h->last_owners / 8)) {
                const unsigned long long *q = c.data() + 8 * o;
                const double e = (double)std::max(1ull, q[0]);
                fprintf(stderr, "[slim prof] owner %3d: %5llu entries, mean cycles: ticket %.0f, gather %.0f, wait for turn %.0f, turn %.0f (row sum %.0f, sigmoid + optimiser %.0f, "
                        "row update + release %.0f), write-back %.0f\n", o, q[0], q[1] / e, q[2] / e, q[3] / e, q[4] / e, q[6] / e, q[7] / e, (q[4] - q[6] - q[7]) / e, q[5] / e);
            }
            const unsigned long long *q = c.data() + 8 * MAX_OWNERS;
            fprintf(stderr, "[slim prof] cold: %llu steps, mean cycles: ids + tickets %.0f, gather .. tickets passed on %.0f\n", q[0],
                    (double)q[1] / std::max(1ull, q[0]), (double)q[2] / std::max(1ull, q[0]));
        }
    }
    h->steps_done += n;
}

// Steps first .. first + n - 1 of a set, scheduled and run on the main stream, one thing after the other.
template <class T>
void run_stream(mi355rec_slim *h, StreamSet &st, int n, double &sum_profile, int first = 0) {
    schedule_stream(h, st, n, first, h->stream);
    sum_profile += 0.5 * (double)st.n_cells;
    Launched L;
    launch_stream<T>(h, st, n, first, L);
    finish_stream(h, L, n);
    st.epoch = -1;
}

void prune_rows(mi355rec_slim *h, int with_diag) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    h->dispatch_timers.next(e0, e1, 1 << 30);
    hipExtLaunchKernelGGL(slim_prune_kernel, dim3(std::min(h->n_items, multiprocessor_count() * 8)), dim3(PRUNE_THREADS), 0, h->stream,
                          e0, e1, 0, reinterpret_cast<unsigned long long *>(h->S.ptr), h->n_items, h->cfg.topK, with_diag);
    h->stats.n_launches += 1;
    MI_HIP(hipGetLastError());
}

// One epoch of n steps of a set, scheduled and run on the main stream.  Sparse store: the stream is cut after every step whose
// index is a positive multiple of n / 5 -- `numCurrentBatch % (totalNumberOfBatch/5) == 0 and numCurrentBatch != 0` with C integer
// division (.pyx:320-324; the module sets cdivision) -- and the rows are pruned there.
template <class T>
void run_epoch_stream(mi355rec_slim *h, StreamSet &st, int n, double &sum_profile) {
    if (!h->cfg.train_with_sparse_weights || n < 5) {
        run_stream<T>(h, st, n, sum_profile);
        return;
    }
    const int every = n / 5;
    int first = 0;
    while (first < n) {
        // steps first .. cut (inclusive) run, then the rows are pruned if `cut` is a rebalance point
        const int cut = std::max(1, (first + every - 1) / every) * every;      // next multiple of `every` at or after `first`, never step 0
        const int last = std::min(cut, n - 1);
        run_stream<T>(h, st, last - first + 1, sum_profile, first);
        if (cut <= n - 1) prune_rows(h, 0);
        first = last + 1;
    }
}

void require_consistent(const mi355rec_slim *h) {
    if (h->aborted) fail(MI355REC_E_HIP, "SLIM-BPR: an earlier call on this handle was aborted inside an epoch; its model is inconsistent -- create a new handle");
}

void begin_call(mi355rec_slim *h) {
    require_consistent(h);
    MI_HIP(hipMemsetAsync(h->loss_slots.ptr, 0, sizeof(double) * LOSS_SLOTS, h->stream));
    h->dispatch_timers.reset();
    h->stats = mi355rec_stats{};
    h->call_timer.start(h->stream);
}

void end_call(mi355rec_slim *h, long long n_steps, double sum_profile) {
    h->call_timer.stop(h->stream);
    h->h_loss.resize(LOSS_SLOTS);
    h->loss_slots.download(h->h_loss.data(), LOSS_SLOTS, h->stream);
    MI_HIP(hipStreamSynchronize(h->stream));
    double loss = 0;
    for (double v : h->h_loss) loss += v;
    h->stats.call_ms = h->call_timer.elapsed_ms();
    h->stats.kernel_ms = h->dispatch_timers.total_ms();
    h->stats.n_timed = h->dispatch_timers.used;
    h->stats.n_units = n_steps;
    // ALGORITHMIC bytes, SURVEY.md section 8(d): 20 * L_u per step (profile ids + 2 gathered rows read + 2 rows written)
    h->stats.algorithmic_bytes = 20.0 * sum_profile;
    h->stats.loss = loss;
}

template <class T>
void run_epochs_typed(mi355rec_slim *h, int n_epochs) {
    const int n = h->n_users + 1;                            // totalNumberOfBatch with batch_size 1 (.pyx:215)
    ensure_set_capacity(h, h->set[0], (size_t)n);
    ensure_set_capacity(h, h->set[1], (size_t)n);
    begin_call(h);
    double sum_profile = 0;
    const bool ahead = schedules_ahead(h);
    for (int e = 0; e < n_epochs; ++e) {
        const long long epoch = h->epochs_done;
        StreamSet &st = h->set[h->cur ^ 1];
        if (!ahead) {
            draw_epoch<T>(h, st, epoch, n, h->stream);
            run_epoch_stream<T>(h, st, n, sum_profile);
        } else {
            if (!(st.epoch == epoch && st.n == n)) {            // (first epoch of the handle, or a replayed stream took the set)
                draw_epoch<T>(h, st, epoch, n, h->side);
                schedule_stream(h, st, n, 0, h->side);
                st.epoch = epoch;
            }
            sum_profile += 0.5 * (double)st.n_cells;
            Launched L;
            launch_stream<T>(h, st, n, 0, L);
            // ... and while it runs: the next epoch's stream, into the other set (kept for the next call if this was the last)
            StreamSet &nx = h->set[h->cur];
            try {
                draw_epoch<T>(h, nx, epoch + 1, n, h->side);
                schedule_stream(h, nx, n, 0, h->side);
                nx.epoch = epoch + 1;
            } catch (...) {
                // The look-ahead failed (allocation, a HIP error) while this epoch's kernel is in flight: finish the epoch first --
                // drain, return the lease, advance the counters (S has moved) -- and drop the half-made set; only then report
                // (the lease went back with owners still resident and the epoch would have been applied twice).
                nx.epoch = -1;
                nx.n = 0;
                finish_stream(h, L, n);
                h->cur ^= 1;
                h->epochs_done += 1;
                h->last_native = true;
                throw;
            }
            finish_stream(h, L, n);
        }
        h->cur ^= 1;
        h->epochs_done += 1;
        h->last_native = true;
    }
    end_call(h, (long long)n * n_epochs, sum_profile);
}

// W = similarityMatrixTopK(get_S(), k) on the device (Base/Recommender_utils.py:55-122 applied to the per-row selection of .pyx:343-391;
// SLIM_BPR_Cython.py:186-197 does this at every validation, and on the host the column step alone -- 9 000 over-full columns ranked one
// by one -- was 0.09 s at ML-20M size against 2 ms for the epoch it follows).  From the (row, K) slabs: one radix sort of the non-zero
// entries by (column, value descending, row descending) ranks every column -- the host function's stable ascending sort drops the first
// len - k entries of a column, i.e. of equal values it keeps the HIGHEST rows --, entries ranked below k are dropped, a second sort by
// (row, column) puts the survivors into canonical CSR order.  Items are 16-bit here (n_items <= 65 535), like everywhere on this path.
__global__ void slim_w_rank_keys_kernel(const int *idx, const float *val, size_t n_slots, int topK, unsigned long long *key, int *slot) {
    const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (q >= n_slots) return;
    const int col = idx[q];
    const float v = val[q];
    slot[q] = (int)q;
    if (col < 0 || v == 0.f) {
        key[q] = ~0ull;                                   // (padding and zeros sort behind every column)
        return;
    }
    const unsigned row = (unsigned)(q / (size_t)topK);
    key[q] = ((unsigned long long)(unsigned)col << 48) | ((unsigned long long)(~float_key(v)) << 16) | (unsigned long long)(0xFFFFu - row);
}
// position of every column's first entry in the ranked order (binary search on the column field; n_items + 1 entries, the last = the
// number of real entries)
__global__ void slim_w_col_start_kernel(const unsigned long long *ranked, size_t n_slots, int n_items, int *start) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_items) return;
    size_t lo = 0, hi = n_slots;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        const unsigned long long k = ranked[mid];
        const bool before = k != ~0ull && (int)(k >> 48) < c;
        if (before) lo = mid + 1; else hi = mid;
    }
    start[c] = (int)lo;
}
// the survivors' (row, column) keys and values, in ranked order; everything else sorts behind them
__global__ void slim_w_keep_kernel(const unsigned long long *ranked, const int *slot, const int *col_start, const float *val, size_t n_slots, int topK,
                                   int k_cols, unsigned *key2, float *val2) {
    const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (q >= n_slots) return;
    const unsigned long long k = ranked[q];
    key2[q] = ~0u;
    val2[q] = 0.f;
    if (k == ~0ull) return;
    const int col = (int)(k >> 48);
    if ((int)q - col_start[col] >= k_cols) return;
    const int s = slot[q];
    key2[q] = ((unsigned)(s / topK) << 16) | (unsigned)col;
    val2[q] = val[s];
}
__global__ void slim_w_csr_kernel(const unsigned *key2, size_t n_slots, int n_items, int *indptr, int *indices) {
    const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (q < n_slots && key2[q] != ~0u) indices[q] = (int)(key2[q] & 0xFFFFu);
    if (q <= (size_t)n_items) {                          // first entry whose row is >= q
        size_t lo = 0, hi = n_slots;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            const bool before = key2[mid] != ~0u && (key2[mid] >> 16) < (unsigned)q;
            if (before) lo = mid + 1; else hi = mid;
        }
        indptr[q] = (int)lo;
    }
}

template <class T>
void topk_slabs_on_device(mi355rec_slim *h, int topK, DeviceBuffer<int> &d_idx, DeviceBuffer<float> &d_val);

template <class T>
void get_w_csr_typed(mi355rec_slim *h, int topK, int *nbr_idx, float *nbr_val, int *indptr, int *indices, float *data, long long *nnz) {
    MI_REQUIRE(h->n_items <= 65535, "n_items = %d: the device-side column selection packs items into 16 bits", h->n_items);
    hipStream_t s = h->stream;
    DeviceBuffer<int> d_idx, slot, slot_sorted, col_start, d_indptr, d_indices;
    DeviceBuffer<float> d_val, val2, val2_sorted;
    DeviceBuffer<unsigned long long> key, key_sorted;
    DeviceBuffer<unsigned> key2, key2_sorted;
    DeviceBuffer<unsigned char> tmp;
    topk_slabs_on_device<T>(h, topK, d_idx, d_val);
    const size_t n_slots = (size_t)h->n_items * topK;
    MI_REQUIRE(n_slots < (size_t)INT32_MAX, "too many slab entries");
    key.alloc(n_slots); key_sorted.alloc(n_slots); slot.alloc(n_slots); slot_sorted.alloc(n_slots);
    const unsigned grid = (unsigned)((n_slots + 255) / 256);
    hipLaunchKernelGGL(slim_w_rank_keys_kernel, dim3(grid), dim3(256), 0, s, d_idx.ptr, d_val.ptr, n_slots, topK, key.ptr, slot.ptr);
    size_t bytes = 0;
    MI_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.ptr, key_sorted.ptr, slot.ptr, slot_sorted.ptr, n_slots, 0, 64, s));
    tmp.alloc(bytes + 16);
    MI_HIP(rocprim::radix_sort_pairs(tmp.ptr, bytes, key.ptr, key_sorted.ptr, slot.ptr, slot_sorted.ptr, n_slots, 0, 64, s));
    col_start.alloc((size_t)h->n_items + 1);
    hipLaunchKernelGGL(slim_w_col_start_kernel, dim3((unsigned)((h->n_items + 256) / 256)), dim3(256), 0, s, key_sorted.ptr, n_slots, h->n_items, col_start.ptr);
    key2.alloc(n_slots); key2_sorted.alloc(n_slots); val2.alloc(n_slots); val2_sorted.alloc(n_slots);
    hipLaunchKernelGGL(slim_w_keep_kernel, dim3(grid), dim3(256), 0, s, key_sorted.ptr, slot_sorted.ptr, col_start.ptr, d_val.ptr, n_slots, topK, topK,
                       key2.ptr, val2.ptr);
    size_t bytes2 = 0;
    MI_HIP(rocprim::radix_sort_pairs(nullptr, bytes2, key2.ptr, key2_sorted.ptr, val2.ptr, val2_sorted.ptr, n_slots, 0, 32, s));
    if (bytes2 + 16 > tmp.count) tmp.alloc(bytes2 + 16);
    MI_HIP(rocprim::radix_sort_pairs(tmp.ptr, bytes2, key2.ptr, key2_sorted.ptr, val2.ptr, val2_sorted.ptr, n_slots, 0, 32, s));
    d_indptr.alloc((size_t)h->n_items + 1); d_indices.alloc(n_slots);
    hipLaunchKernelGGL(slim_w_csr_kernel, dim3(std::max(grid, (unsigned)((h->n_items + 256) / 256))), dim3(256), 0, s, key2_sorted.ptr, n_slots, h->n_items,
                       d_indptr.ptr, d_indices.ptr);
    MI_HIP(hipGetLastError());
    d_indptr.download(indptr, (size_t)h->n_items + 1, s);
    MI_HIP(hipStreamSynchronize(s));
    const size_t n_kept = (size_t)indptr[h->n_items];
    *nnz = (long long)n_kept;
    if (n_kept) {
        d_indices.download(indices, n_kept, s);
        val2_sorted.download(data, n_kept, s);
    }
    if (nbr_idx) d_idx.download(nbr_idx, n_slots, s);
    if (nbr_val) d_val.download(nbr_val, n_slots, s);
    MI_HIP(hipStreamSynchronize(s));
}

template <class T>
void topk_slabs_on_device(mi355rec_slim *h, int topK, DeviceBuffer<int> &d_idx, DeviceBuffer<float> &d_val) {
    const int n_pad = (h->n_items + 3) & ~3;
    const size_t lds = (size_t)n_pad * 4 + (size_t)AUX_WORDS * 4;
    if (lds + 2048 > 160 * 1024)
        fail(MI355REC_E_UNSUPPORTED, "n_items = %d: a row of S does not fit the 160 KiB LDS for the top-K selection", h->n_items);
    const size_t n_out = (size_t)h->n_items * topK;
    d_idx.alloc(n_out);
    d_val.alloc(n_out);
    SlimParams<T> p{};
    fill_params(h, h->set[h->cur], p);
    auto k = slim_topk_kernel<T, 1024>;
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / (lds + 2048))));
    hipLaunchKernelGGL(k, dim3(std::min(h->n_items, multiprocessor_count() * per_cu)), dim3(1024), lds, h->stream, p, topK,
                       n_pad, d_idx.ptr, d_val.ptr);
    MI_HIP(hipGetLastError());
}

template <class T>
void get_topk_typed(mi355rec_slim *h, int topK, int *nbr_idx, float *nbr_val) {
    DeviceBuffer<int> d_idx;
    DeviceBuffer<float> d_val;
    topk_slabs_on_device<T>(h, topK, d_idx, d_val);
    const size_t n_out = (size_t)h->n_items * topK;
    d_idx.download(nbr_idx, n_out, h->stream);
    d_val.download(nbr_val, n_out, h->stream);
    MI_HIP(hipStreamSynchronize(h->stream));
}

template <class T>
void get_dense_typed(mi355rec_slim *h, float *S) {
    const size_t n2 = (size_t)h->n_items * h->n_items;
    DeviceBuffer<float> out;
    out.alloc(n2);
    SlimParams<T> p{};
    fill_params(h, h->set[h->cur], p);
    hipLaunchKernelGGL(slim_dense_kernel<T>, dim3((unsigned)std::min<size_t>((n2 + 255) / 256, 8192)), dim3(256), 0, h->stream, p, out.ptr);
    MI_HIP(hipGetLastError());
    out.download(S, n2, h->stream);
    MI_HIP(hipStreamSynchronize(h->stream));
}

}  // namespace

extern "C" int mi355rec_slim_create(mi355rec_slim_t *out, const mi355rec_slim_config *cfg, int32_t n_users, int32_t n_items,
                                    const int32_t *indptr, const int32_t *indices) {
    return guarded([&] {
        MI_REQUIRE(out && cfg && indptr && indices, "NULL argument");
        MI_REQUIRE(n_users > 0 && n_items > 0, "empty URM");
        MI_REQUIRE(cfg->sgd_mode >= MI355REC_SGD && cfg->sgd_mode <= MI355REC_ADAM, "Value for 'sgd_mode' not recognized (%d)",
                   cfg->sgd_mode);
        MI_REQUIRE(cfg->precision == MI355REC_F32 || cfg->precision == MI355REC_F64, "precision must be MI355REC_F32 or MI355REC_F64");
        ensure_device();
        std::unique_ptr<mi355rec_slim> h(new mi355rec_slim());
        h->cfg = *cfg;
        if (cfg->train_with_sparse_weights) {
            MI_REQUIRE(cfg->precision == MI355REC_F64, "train_with_sparse_weights needs precision MI355REC_F64 (the selections compare values)");
            MI_REQUIRE(cfg->topK >= 0, "topK must be >= 0 (0 = False)");
            h->cfg.symmetric = 0;                                // .pyx:112-113
        }
        h->n_users = n_users;
        h->n_items = n_items;
        // the symmetric store keeps float32 values inside 8-byte granules and computes in float64 whatever `precision` says
        h->f64 = cfg->precision == MI355REC_F64 || h->cfg.symmetric;
        h->nnz = (size_t)indptr[n_users];
        MI_REQUIRE(h->nnz > 0, "URM has no interactions");
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        MI_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        h->call_timer.init();
        h->dispatch_timers.reserve(64);
        hipStream_t s = h->stream;
        const size_t ts = h->f64 ? sizeof(double) : sizeof(float);
        h->indptr.upload(indptr, (size_t)n_users + 1, s);
        h->indices.upload(indices, h->nnz, s);
        if (h->cfg.symmetric) {
            // Triangular_Matrix :1237-1254 the packed lower triangle
            h->G.alloc_zero((size_t)n_items * ((size_t)n_items + 1) / 2, s);
            h->oc.alloc_zero(4 * (size_t)n_items, s);
        } else {
            h->S.alloc_zero((size_t)n_items * n_items * ts, s);     // .pyx:129 dense n x n
            if (h->cfg.train_with_sparse_weights)                        // .pyx:124: an empty tree per row
                hipLaunchKernelGGL(slim_no_nodes_kernel, dim3(multiprocessor_count() * 8), dim3(256), 0, s,
                                   reinterpret_cast<unsigned long long *>(h->S.ptr), (size_t)n_items * n_items);
            h->c1.alloc_zero((size_t)n_items * ts, s);
            h->c2.alloc_zero((size_t)n_items * ts, s);
            h->ticket.alloc_zero((size_t)n_items, s);
            h->hot_rank.alloc((size_t)n_items);
            h->hot_tables.alloc_zero(3 * MAX_OWNERS, s);
            h->counters.alloc_zero(2, s);
            h->cnt_sorted.alloc((size_t)n_items);
            h->item_by_cnt.alloc((size_t)n_items);
            std::vector<int> iota((size_t)n_items);
            for (int i = 0; i < n_items; ++i) iota[(size_t)i] = i;
            h->iota.upload(iota.data(), iota.size(), s);
            MI_HIP(hipStreamSynchronize(s));                         // (iota is read from host memory)
        }
        h->queue.alloc_zero(4, s);
        h->loss_slots.alloc_zero(LOSS_SLOTS, s);
        MI_HIP(hipStreamSynchronize(s));
        *out = h.release();
    });
}

extern "C" int mi355rec_slim_run_epochs(mi355rec_slim_t h, int32_t n_epochs) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        MI_REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        ensure_device();
        ReleaseScope scope(h->stream, h->side);
        if (h->f64) run_epochs_typed<double>(h, n_epochs); else run_epochs_typed<float>(h, n_epochs);
    });
}

extern "C" int mi355rec_slim_run_samples(mi355rec_slim_t h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n) {
    return guarded([&] {
        MI_REQUIRE(h && u && i && j, "NULL argument");
        MI_REQUIRE(n >= 0 && n < (1ll << 30), "n out of range");
        ensure_device();
        if (n == 0) return;
        ReleaseScope scope(h->stream, h->side);
        StreamSet &st = h->set[h->cur ^ 1];                  // (a stream scheduled ahead for the next native epoch, if any, is given up)
        ensure_set_capacity(h, st, (size_t)n);
        st.epoch = -1;
        hipStream_t s = h->stream;
        for (int64_t t = 0; t < n; ++t)
            MI_REQUIRE(u[t] >= 0 && u[t] < h->n_users && i[t] >= 0 && i[t] < h->n_items && j[t] >= 0 && j[t] < h->n_items && i[t] != j[t],
                       "sample %lld out of range (or its positive item is its negative item)", (long long)t);
        MI_HIP(hipMemcpyAsync(st.su.ptr, u, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(st.si.ptr, i, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(st.sj.ptr, j, sizeof(int) * n, hipMemcpyHostToDevice, s));
        begin_call(h);
        double sum_profile = 0;
        // (the n samples are ONE epoch of n steps: that is what the sparse store's rebalance rule counts against)
        if (h->f64) run_epoch_stream<double>(h, st, (int)n, sum_profile); else run_epoch_stream<float>(h, st, (int)n, sum_profile);
        h->cur ^= 1;
        h->last_native = false;
        end_call(h, n, sum_profile);
    });
}

extern "C" int mi355rec_slim_get_last_samples(mi355rec_slim_t h, int32_t *u, int32_t *i, int32_t *j, int64_t cap, int64_t *n) {
    return guarded([&] {
        MI_REQUIRE(h && n, "NULL argument");
        ensure_device();
        const int64_t have = h->epochs_done > 0 && h->last_native ? h->n_users + 1 : 0;
        *n = have;
        const size_t m = (size_t)std::min<int64_t>(cap, have);
        const StreamSet &st = h->set[h->cur];
        if (u && m) st.su.download(u, m, h->stream);
        if (i && m) st.si.download(i, m, h->stream);
        if (j && m) st.sj.download(j, m, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
    });
}

extern "C" int mi355rec_slim_get_S_topk(mi355rec_slim_t h, int32_t topK, int32_t *nbr_idx, float *nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && nbr_idx && nbr_val, "NULL argument");
        require_consistent(h);
        MI_REQUIRE(topK >= 1, "topK must be >= 1 (use mi355rec_slim_get_S_dense for the full matrix)");
        ensure_device();
        topK = std::min(topK, h->n_items);
        if (topK > MAX_TOPK) fail(MI355REC_E_UNSUPPORTED, "topK = %d exceeds the in-LDS selection limit of %d", topK, MAX_TOPK);
        if (h->f64) get_topk_typed<double>(h, topK, nbr_idx, nbr_val); else get_topk_typed<float>(h, topK, nbr_idx, nbr_val);
    });
}

extern "C" int mi355rec_slim_get_W_csr(mi355rec_slim_t h, int32_t topK, int32_t *nbr_idx, float *nbr_val, int32_t *indptr, int32_t *indices,
                                       float *data, int64_t *nnz) {
    return guarded([&] {
        MI_REQUIRE(h && indptr && indices && data && nnz, "NULL argument");
        require_consistent(h);
        MI_REQUIRE(topK >= 1, "topK must be >= 1");
        ensure_device();
        ReleaseScope scope(h->stream, h->side);
        topK = std::min(topK, h->n_items);
        if (topK > MAX_TOPK) fail(MI355REC_E_UNSUPPORTED, "topK = %d exceeds the in-LDS selection limit of %d", topK, MAX_TOPK);
        long long n = 0;
        if (h->f64) get_w_csr_typed<double>(h, topK, nbr_idx, nbr_val, indptr, indices, data, &n);
        else get_w_csr_typed<float>(h, topK, nbr_idx, nbr_val, indptr, indices, data, &n);
        *nnz = (int64_t)n;
    });
}

extern "C" int mi355rec_slim_get_S_sparse(mi355rec_slim_t h, int32_t *nbr_idx, float *nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && nbr_idx && nbr_val, "NULL argument");
        require_consistent(h);
        MI_REQUIRE(h->cfg.train_with_sparse_weights && h->cfg.topK >= 1, "handle was not created with train_with_sparse_weights and topK >= 1");
        ensure_device();
        const int width = h->cfg.topK;
        const size_t n_out = (size_t)h->n_items * width;
        DeviceBuffer<int> d_idx;
        DeviceBuffer<float> d_val;
        d_idx.alloc(n_out);
        d_val.alloc(n_out);
        h->dispatch_timers.reset();
        prune_rows(h, 1);
        hipLaunchKernelGGL(slim_list_kernel, dim3(std::min(h->n_items, multiprocessor_count() * 8)), dim3(PRUNE_THREADS), 0, h->stream,
                           reinterpret_cast<const unsigned long long *>(h->S.ptr), h->n_items, width, d_idx.ptr, d_val.ptr);
        MI_HIP(hipGetLastError());
        d_idx.download(nbr_idx, n_out, h->stream);
        d_val.download(nbr_val, n_out, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
    });
}

extern "C" int mi355rec_slim_get_S_dense(mi355rec_slim_t h, float *S) {
    return guarded([&] {
        MI_REQUIRE(h && S, "NULL argument");
        require_consistent(h);
        ensure_device();
        if (h->f64) get_dense_typed<double>(h, S); else get_dense_typed<float>(h, S);
    });
}

extern "C" int mi355rec_slim_get_stats(mi355rec_slim_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" int mi355rec_slim_schedule_info(mi355rec_slim_t h, int32_t *n_owned_rows, int32_t *n_cold_steps) {
    return guarded([&] {
        MI_REQUIRE(h && n_owned_rows && n_cold_steps, "NULL argument");
        *n_owned_rows = h->last_owners;
        *n_cold_steps = h->last_cold;
    });
}

extern "C" void mi355rec_slim_destroy(mi355rec_slim_t h) {
    if (!h) return;
    ReleaseScope scope(h->stream, h->side);
    delete h;
}
