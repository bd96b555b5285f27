// slim.hip -- SLIM-BPR epoch on MI355X (gfx950).
//
// Replaces SLIM_BPR/Cython/SLIM_BPR_Cython_Epoch.pyx (reference): epochIteration_Cython :212-317 with the dense
// store (:129) and the symmetric triangular store (Triangular_Matrix :1226-1330), sampleBPR_Cython :439-483,
// adaptive_gradient :398-436, get_S :343-391.  The wrapper hard-codes batch_size = 1
// (SLIM_BPR/Cython/SLIM_BPR_Cython.py:140), so an epoch is n_users + 1 STRICTLY ORDERED SGD steps.
//
// Design.  S is a dense n_items x n_items matrix in HBM (the symmetric store uses the lower triangle
// of the same array); float32 for plain sgd, float64 (like the reference) for adagrad / rmsprop / adam.  The sample stream
// does not depend on S, so WHICH earlier step a step has to wait for is known before the first step runs:
//   dense store       step t owns rows i_t and j_t (and the optimiser cells of items i_t, j_t).  Per item, steps take
//                     numbered tickets in stream order (a device sort of the 2n (item, step) pairs gives every step its two
//                     ticket numbers);
//   symmetric store   cell (r, c) aliases (c, r), so ownership is per CELL: the cells every step touches are sorted by
//                     (cell, step) and each one is handed the step that touched it last (`pred`).
// Round 1 turned the row dependencies into ~1250 host-scheduled level launches per epoch (dense, 10.6 us per level) and ran
// the symmetric store on ONE workgroup.  Here ONE persistent kernel executes the whole stream as a dataflow graph: workgroups
// pull steps from an in-order queue; a step waits until its two tickets come up (and, symmetric, until the last writer of
// each of its cells is done), does its 2 L_u gathers, the two per-item optimiser steps, its 2 L_u scattered writes, drains
// them, and passes the tickets on.  Cells are only ever accessed with agent-scope atomic loads / write-through stores
// (L2 / MALL coherent across XCDs); waiting is a relaxed poll of ONE word per lane.  Because the queue is in order, a
// waiting step only ever waits for steps that are already running: no deadlock, whatever the residency.
// Exact sequential semantics; the critical path is the chain of steps on the most popular item, not launches.
// 4-byte gathers / scatters, no dense contraction: no MFMA.
#include "common.h"
#include "sampling.cuh"
#include "topk.cuh"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <memory>

namespace mi355rec {
namespace {

constexpr int LOSS_SLOTS = 1024;
constexpr int FLOW_THREADS = 256;
constexpr int FLOW_REGS = 4;                          // profile entries per thread whose cells stay in registers between the passes
constexpr unsigned NO_CELL = 0xFFFFFFFFu;             // (the diagonal is read but never written: it orders nothing)
constexpr long long SPIN_LIMIT_TICKS = 2000000000ll;  // 20 s of the 100 MHz wall clock: a stuck hand-off aborts instead of hanging

template <class T>
struct SlimParams {
    int n_users, n_items, symmetric, sgd_mode;
    T lr, li_reg, lj_reg, gamma, beta_1, beta_2, one_m_gamma, one_m_beta_1, one_m_beta_2;
    double beta_1_d, beta_2_d;
    unsigned long long seed;
    const int *indptr, *indices;
    T *S;
    T *c1, *c2;                     // per-ITEM optimiser scalars (.pyx:177-181): cache / first moment, second moment
    const int *su, *si, *sj;        // sample stream of the call
    const int *seq;                 // [2 n_steps] ticket numbers of step t on item i_t (2t) and item j_t (2t + 1)
    const long long *cellptr;       // symmetric: first cell slot of every step (2 per profile entry: row i, row j)
    const int *pred;                // symmetric: per cell slot, the step that touched the cell last (-1: nobody in this call)
    int *ticket;                    // [n_items] steps of this call completed on the item
    int *done;                      // symmetric: [n_steps]
    int *queue;                     // [0] next step, [1] abort flag
    double *loss_slots;             // [LOSS_SLOTS]
    long long epoch;                // RNG counter base
    long long steps_before;         // steps executed before this call (Adam's beta^t, .pyx:313-317)
    int n_steps, use_tickets;
};

template <class T> __device__ __forceinline__ T aload(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void astore(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <class P> __device__ __forceinline__ size_t cell_at(const P &p, int r, int c) {
    // Triangular_Matrix.get_value/add_value (.pyx:1290-1330): in symmetric mode (r, c) with c > r lives at (c, r)
    // and the store is the packed lower triangle, row r starting at r (r + 1) / 2 (:1237-1254): n (n + 1) / 2 cells
    if (p.symmetric) {
        if (c > r) { const int t = r; r = c; c = t; }
        return ((size_t)r * ((size_t)r + 1) >> 1) + (size_t)c;
    }
    return (size_t)r * p.n_items + c;
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

// per-ITEM adaptive step (.pyx:398-436); pw1 / pw2 = 1 - beta^t of this step.  The cells travel between workgroups with the
// item's ticket, hence the agent-scope accesses.
template <class T>
__device__ __forceinline__ T slim_adapt_cells(const SlimParams<T> &p, T g, int item, T pw1, T pw2, T c1, T c2) {
    // c1 / c2: the item's optimiser cells as the owner of the item's ticket read them
    switch (p.sgd_mode) {
        case MI355REC_ADAGRAD: {
            const T c = c1 + g * g;
            astore(&p.c1[item], c);
            return g / (root(c) + (T)1e-8);
        }
        case MI355REC_RMSPROP: {
            const T c = c1 * p.gamma + p.one_m_gamma * (g * g);
            astore(&p.c1[item], c);
            return g / (root(c) + (T)1e-8);
        }
        case MI355REC_ADAM: {
            const T m1 = c1 * p.beta_1 + p.one_m_beta_1 * g;
            const T m2 = c2 * p.beta_2 + p.one_m_beta_2 * (g * g);
            astore(&p.c1[item], m1);
            astore(&p.c2[item], m2);
            return (m1 / pw1) / (root(m2 / pw2) + (T)1e-8);
        }
        default:
            return g;
    }
}

template <class T>
__device__ __forceinline__ T slim_adapt(const SlimParams<T> &p, T g, int item, T pw1, T pw2) {
    T c1 = (T)0, c2 = (T)0;
    if (p.sgd_mode != MI355REC_SGD) c1 = aload(&p.c1[item]);
    if (p.sgd_mode == MI355REC_ADAM) c2 = aload(&p.c2[item]);
    return slim_adapt_cells(p, g, item, pw1, pw2, c1, c2);
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
    int *seq;
    int *len2;                      // 2 L_u per step
    const long long *cellptr;
    int *pred;
    long long n_cells;
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

// ticket number = how many earlier steps of the stream touch the same item = position inside the item's run
__global__ __launch_bounds__(256) void slim_seq_kernel(const DepParams d) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= 2 * d.n_steps) return;
    const unsigned long long first_key = d.keys_sorted[q] & 0xFFFFFFFF00000000ull;
    int lo = 0, hi = q;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (d.keys_sorted[mid] < first_key) lo = mid + 1; else hi = mid;
    }
    d.seq[d.vals_sorted[q]] = q - lo;
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
        // (packed lower triangle, as cell_at: fits 32 bits up to 92 681 items)
        const unsigned ci = s == i ? NO_CELL : (unsigned)(((size_t)max(i, s) * ((size_t)max(i, s) + 1) >> 1) + (size_t)min(i, s));
        const unsigned cj = s == j ? NO_CELL : (unsigned)(((size_t)max(j, s) * ((size_t)max(j, s) + 1) >> 1) + (size_t)min(j, s));
        d.keys[cp + 2 * idx] = ((unsigned long long)ci << 32) | (unsigned)t;
        d.vals[cp + 2 * idx] = (int)(cp + 2 * idx);
        d.keys[cp + 2 * idx + 1] = ((unsigned long long)cj << 32) | (unsigned)t;
        d.vals[cp + 2 * idx + 1] = (int)(cp + 2 * idx + 1);
    }
}

__global__ __launch_bounds__(256) void slim_pred_kernel(const DepParams d) {
    const long long q = blockIdx.x * 256ll + threadIdx.x;
    if (q >= d.n_cells) return;
    const unsigned long long key = d.keys_sorted[q];
    const unsigned cell = (unsigned)(key >> 32);
    int pred = -1;
    if (q > 0 && cell != NO_CELL) {
        const unsigned long long before = d.keys_sorted[q - 1];
        if ((unsigned)(before >> 32) == cell) pred = (int)(before & 0xFFFFFFFFull);
    }
    d.pred[d.vals_sorted[q]] = pred;
}

// ---- the stream ---------------------------------------------------------------------------------------------------------
// relaxed poll of one word; a hand-off that does not arrive within SPIN_LIMIT_TICKS raises the abort flag (everybody stops
// waiting, the call fails) instead of hanging the device
template <class T>
__device__ __forceinline__ void wait_for(const SlimParams<T> &p, const int *word, int want, bool exact) {
    unsigned polls = 0;
    long long t0 = 0;
    for (;;) {
        const int v = aload(word);
        if (exact ? v == want : v != 0) return;
        __builtin_amdgcn_s_sleep(2);
        if ((++polls & 255u) == 0) {
            if (aload(&p.queue[1])) return;
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > SPIN_LIMIT_TICKS) { astore(&p.queue[1], 1); return; }
        }
    }
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// BATCHED (experimental, MI355REC_SLIM_BATCHED=1, symmetric store only; NOT in the test suite): the `done` words of all last
// writers of a thread's cells are requested under one wait, only the stragglers are polled, then all cells are gathered under one
// wait -- instead of poll, gather, poll, gather with up to 3 x FLOW_REGS dependent round trips per step.  Round 3: equal results on
// small matrices, but the ML-1M-shape symmetric epoch stalled (twice; with `done[0]` and with `done[t]` as the dummy word of lanes
// without a predecessor) and the cause was not found -- kept so that the next session with a GPU can debug it.
template <class T, bool SYM, bool BATCHED = false>
__global__ __launch_bounds__(FLOW_THREADS) void slim_flow_kernel(const SlimParams<T> p) {
    __shared__ int s_t;
    __shared__ T s_part[FLOW_THREADS / 64];
    __shared__ T s_g[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (;;) {
        if (tid == 0) s_t = atomicAdd(&p.queue[0], 1);         // in-order queue: everything this step can wait for is already running
        __syncthreads();
        const int t = s_t;
        if (t >= p.n_steps) break;
        const int u = p.su[t], i = p.si[t], j = p.sj[t];
        const int rs = p.indptr[u], L = p.indptr[u + 1] - rs;
        const long long cp = SYM ? p.cellptr[t] : 0;
        // profile entries and (symmetric) the last writers of their cells do not depend on anybody: fetch them before waiting
        int sv[FLOW_REGS], pa[FLOW_REGS], pb[FLOW_REGS];
        if constexpr (BATCHED) {
            // loads from clamped, always valid addresses, masked afterwards: with a load inside a conditional the compiler waits for
            // each one where its branch ends
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {
                const int idx = tid + r * FLOW_THREADS, at = min(idx, L - 1);        // L >= 1: users without interactions are never drawn
                sv[r] = p.indices[rs + at];
                const int2 pp = *reinterpret_cast<const int2 *>(p.pred + cp + 2 * at);
                pa[r] = pp.x;
                pb[r] = pp.y;
            }
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {
                const bool live = tid + r * FLOW_THREADS < L;
                pa[r] = live ? pa[r] : -1;
                pb[r] = live ? pb[r] : -1;
            }
        } else {
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            const int idx = tid + r * FLOW_THREADS;
            sv[r] = idx < L ? p.indices[rs + idx] : 0;
            pa[r] = SYM && idx < L ? p.pred[cp + 2 * idx] : -1;
            pb[r] = SYM && idx < L ? p.pred[cp + 2 * idx + 1] : -1;
        }
        }
        if (p.use_tickets) {
            if (tid < 2) wait_for(p, &p.ticket[tid ? j : i], p.seq[2 * t + tid], true);
            __syncthreads();
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        // the items' optimiser cells belong to whoever holds the items' tickets: requested now, next to the gathers, instead of
        // after the gradient is known (one dependent round trip less per step for adagrad / rmsprop / adam)
        T oc1_i = (T)0, oc1_j = (T)0, oc2_i = (T)0, oc2_j = (T)0;
        if (tid == 0 && p.sgd_mode != MI355REC_SGD) {
            oc1_i = aload(&p.c1[i]);
            oc1_j = aload(&p.c1[j]);
            if (p.sgd_mode == MI355REC_ADAM) {
                oc2_i = aload(&p.c2[i]);
                oc2_j = aload(&p.c2[j]);
            }
        }
        // x_uij over the profile (.pyx:243-260)
        T x = (T)0;
        T va[FLOW_REGS], vb[FLOW_REGS];
        if constexpr (BATCHED) {
            int fa[FLOW_REGS], fb[FLOW_REGS];
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {       // (lanes without a predecessor read this step's own word)
                fa[r] = aload(&p.done[pa[r] >= 0 ? pa[r] : t]);
                fb[r] = aload(&p.done[pb[r] >= 0 ? pb[r] : t]);
            }
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {
                if (pa[r] >= 0 && !fa[r]) wait_for(p, &p.done[pa[r]], 1, false);
                if (pb[r] >= 0 && !fb[r]) wait_for(p, &p.done[pb[r]], 1, false);
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {
                va[r] = aload(&p.S[cell_at(p, i, sv[r])]);
                vb[r] = aload(&p.S[cell_at(p, j, sv[r])]);
            }
#pragma unroll
            for (int r = 0; r < FLOW_REGS; ++r) {
                const bool live = tid + r * FLOW_THREADS < L;
                va[r] = live ? va[r] : (T)0;
                vb[r] = live ? vb[r] : (T)0;
                x += va[r] - vb[r];
            }
        } else {
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            const int idx = tid + r * FLOW_THREADS;
            va[r] = (T)0;
            vb[r] = (T)0;
            if (idx < L) {
                if (SYM) {
                    if (pa[r] >= 0) wait_for(p, &p.done[pa[r]], 1, false);
                    if (pb[r] >= 0) wait_for(p, &p.done[pb[r]], 1, false);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST);
                }
                va[r] = aload(&p.S[cell_at(p, i, sv[r])]);
                vb[r] = aload(&p.S[cell_at(p, j, sv[r])]);
                x += va[r] - vb[r];
            }
        }
        }
        for (int idx = tid + FLOW_REGS * FLOW_THREADS; idx < L; idx += FLOW_THREADS) {    // profiles longer than 1024
            const int s = p.indices[rs + idx];
            if (SYM) {
                const int a = p.pred[cp + 2 * idx], b = p.pred[cp + 2 * idx + 1];
                if (a >= 0) wait_for(p, &p.done[a], 1, false);
                if (b >= 0) wait_for(p, &p.done[b], 1, false);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
            }
            x += aload(&p.S[cell_at(p, i, s)]) - aload(&p.S[cell_at(p, j, s)]);
        }
        x = wave_sum(x);
        if (lane == 0) s_part[wave] = x;
        __syncthreads();
        if (tid == 0) {
            T tot = (T)0;
#pragma unroll
            for (int w = 0; w < FLOW_THREADS / 64; ++w) tot += s_part[w];
            const T g = sigmoid_of_minus(tot);                         // .pyx:263
            T pw1 = (T)1, pw2 = (T)1;
            if (p.sgd_mode == MI355REC_ADAM) {
                const double tt = (double)(p.steps_before + t + 1);
                pw1 = (T)(1.0 - pow(p.beta_1_d, tt));
                pw2 = (T)(1.0 - pow(p.beta_2_d, tt));
            }
            s_g[0] = slim_adapt_cells(p, g, i, pw1, pw2, oc1_i, oc2_i);     // item i first, then j, as .pyx:267-268
            s_g[1] = j != i ? slim_adapt_cells(p, g, j, pw1, pw2, oc1_j, oc2_j) : slim_adapt(p, g, j, pw1, pw2);      // (a replayed stream may repeat the item)
            atomicAdd(&p.loss_slots[t & (LOSS_SLOTS - 1)], (double)tot * (double)tot);
        }
        __syncthreads();
        const T gi = s_g[0], gj = s_g[1];
        // the two rows move (.pyx:271-309); write-through stores
#pragma unroll
        for (int r = 0; r < FLOW_REGS; ++r) {
            const int idx = tid + r * FLOW_THREADS;
            if (idx < L) {
                const int s = sv[r];
                if (s != i) astore(&p.S[cell_at(p, i, s)], cell_plus(va[r], p.lr, gi, p.li_reg));
                if (s != j) astore(&p.S[cell_at(p, j, s)], cell_minus(vb[r], p.lr, gj, p.lj_reg));
            }
        }
        for (int idx = tid + FLOW_REGS * FLOW_THREADS; idx < L; idx += FLOW_THREADS) {
            const int s = p.indices[rs + idx];
            if (s != i) {
                T *c = &p.S[cell_at(p, i, s)];
                const T v = aload(c);
                astore(c, cell_plus(v, p.lr, gi, p.li_reg));
            }
            if (s != j) {
                T *c = &p.S[cell_at(p, j, s)];
                const T v = aload(c);
                astore(c, cell_minus(v, p.lr, gj, p.lj_reg));
            }
        }
        // publish: every storing wavefront drains its write-through stores, then ONE lane passes the tickets on
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (p.use_tickets) {
                astore(&p.ticket[i], p.seq[2 * t] + 1);
                astore(&p.ticket[j], p.seq[2 * t + 1] + 1);
            }
            if (SYM) astore(&p.done[t], 1);
        }
    }
}

// Fallback (symmetric store with more than 92 681 items: packed cell ids no longer fit the 32-bit sort key): one workgroup runs
// the steps one after the other.
template <class T>
__global__ __launch_bounds__(1024) void slim_ordered_kernel(const SlimParams<T> p) {
    __shared__ T s_part[16];
    __shared__ T s_g[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = 0; t < p.n_steps; ++t) {
        const int u = p.su[t], i = p.si[t], j = p.sj[t];
        const int rs = p.indptr[u], re = p.indptr[u + 1];
        T x = (T)0;
        for (int q = rs + tid; q < re; q += 1024) {
            const int s = p.indices[q];
            x += p.S[cell_at(p, i, s)] - p.S[cell_at(p, j, s)];
        }
        x = wave_sum(x);
        if (lane == 0) s_part[wave] = x;
        __syncthreads();
        if (tid == 0) {
            T tot = (T)0;
            for (int w = 0; w < 16; ++w) tot += s_part[w];
            const T g = sigmoid_of_minus(tot);
            T pw1 = (T)1, pw2 = (T)1;
            if (p.sgd_mode == MI355REC_ADAM) {
                const double tt = (double)(p.steps_before + t + 1);
                pw1 = (T)(1.0 - pow(p.beta_1_d, tt));
                pw2 = (T)(1.0 - pow(p.beta_2_d, tt));
            }
            s_g[0] = slim_adapt(p, g, i, pw1, pw2);
            s_g[1] = slim_adapt(p, g, j, pw1, pw2);
            p.loss_slots[t & (LOSS_SLOTS - 1)] += (double)tot * (double)tot;
        }
        __syncthreads();
        const T gi = s_g[0], gj = s_g[1];
        for (int q = rs + tid; q < re; q += 1024) {
            const int s = p.indices[q];
            if (s != i) {
                T *c = &p.S[cell_at(p, i, s)];
                const T v = *c;
                *c = cell_plus(v, p.lr, gi, p.li_reg);
            }
            if (s != j) {
                T *c = &p.S[cell_at(p, j, s)];
                const T v = *c;
                *c = cell_minus(v, p.lr, gj, p.lj_reg);
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
            const float v = c == r ? 0.f : (float)p.S[cell_at(p, r, c)];
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
        out[e] = r == c ? 0.f : (float)p.S[cell_at(p, r, c)];
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

struct mi355rec_slim {
    mi355rec_slim_config cfg{};
    int n_users = 0, n_items = 0;
    bool f64 = false;
    size_t nnz = 0;
    hipStream_t stream = nullptr;
    StreamTimer call_timer;
    DispatchTimers dispatch_timers;
    DeviceBuffer<int> indptr, indices, su, si, sj, seq, len2, ticket, done, queue, vals, vals_sorted, pred;
    DeviceBuffer<long long> cellptr;
    DeviceBuffer<unsigned long long> keys, keys_sorted;
    DeviceBuffer<unsigned char> S, c1, c2, cub_tmp;     // S, c1, c2: float or double by `f64`
    DeviceBuffer<double> loss_slots;
    size_t stream_capacity = 0, cell_capacity = 0;
    long long steps_done = 0, epochs_done = 0;
    std::vector<double> h_loss;
    mi355rec_stats stats{};

    ~mi355rec_slim() {
        if (stream) (void)hipStreamSynchronize(stream);
        call_timer.destroy();
        dispatch_timers.destroy();
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

template <class T>
void fill_params(mi355rec_slim *h, SlimParams<T> &p) {
    const auto &c = h->cfg;
    p.n_users = h->n_users; p.n_items = h->n_items; p.symmetric = c.symmetric; p.sgd_mode = c.sgd_mode;
    p.lr = (T)c.learning_rate; p.li_reg = (T)c.li_reg; p.lj_reg = (T)c.lj_reg;
    p.gamma = (T)c.gamma; p.beta_1 = (T)c.beta_1; p.beta_2 = (T)c.beta_2;
    p.one_m_gamma = (T)(1.0 - c.gamma); p.one_m_beta_1 = (T)(1.0 - c.beta_1); p.one_m_beta_2 = (T)(1.0 - c.beta_2);
    p.beta_1_d = c.beta_1; p.beta_2_d = c.beta_2;
    p.seed = c.random_seed;
    p.indptr = h->indptr.ptr; p.indices = h->indices.ptr;
    p.S = reinterpret_cast<T *>(h->S.ptr); p.c1 = reinterpret_cast<T *>(h->c1.ptr); p.c2 = reinterpret_cast<T *>(h->c2.ptr);
    p.su = h->su.ptr; p.si = h->si.ptr; p.sj = h->sj.ptr;
    p.seq = h->seq.ptr; p.cellptr = h->cellptr.ptr; p.pred = h->pred.ptr;
    p.ticket = h->ticket.ptr; p.done = h->done.ptr; p.queue = h->queue.ptr;
    p.loss_slots = h->loss_slots.ptr;
    p.epoch = h->epochs_done;
    p.steps_before = h->steps_done;
    p.n_steps = 0;
    p.use_tickets = 1;
}

bool flow_supported(const mi355rec_slim *h) {
    return !(h->cfg.symmetric && h->n_items > 92681) && !getenv("MI355REC_SLIM_ORDERED");      // (cell ids are 32-bit sort keys)
}

void ensure_capacity(mi355rec_slim *h, size_t n) {
    if (h->stream_capacity >= n) return;
    h->su.alloc(n); h->si.alloc(n); h->sj.alloc(n);
    h->seq.alloc(2 * n); h->len2.alloc(n + 1); h->done.alloc(n); h->cellptr.alloc(n + 1);
    h->stream_capacity = n;
}

void ensure_sort_capacity(mi355rec_slim *h, size_t n) {
    if (h->cell_capacity >= n) return;
    h->keys.alloc(n); h->keys_sorted.alloc(n); h->vals.alloc(n); h->vals_sorted.alloc(n); h->pred.alloc(n);
    size_t sort_bytes = 0, scan_bytes = 0;
    MI_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, h->keys.ptr, h->keys_sorted.ptr, h->vals.ptr, h->vals_sorted.ptr,
                                              (int)n, 0, 64, h->stream));
    MI_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, h->len2.ptr, h->cellptr.ptr, 0ll, h->stream_capacity + 1, rocprim::plus<long long>(), h->stream));
    h->cub_tmp.alloc(std::max(sort_bytes, scan_bytes) + 256);
    h->cell_capacity = n;
}

int bits_for(unsigned long long n_values) {
    int b = 1;
    while (b < 63 && (1ull << b) < n_values) ++b;
    return b;
}

// Runs n steps of su/si/sj, starting at step `first`, exactly in stream order.
template <class T>
void run_stream(mi355rec_slim *h, int n, double &sum_profile, int first = 0) {
    hipStream_t s = h->stream;
    SlimParams<T> p{};
    fill_params(h, p);
    p.su += first; p.si += first; p.sj += first;
    p.n_steps = n;
    const bool sym = h->cfg.symmetric != 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!flow_supported(h)) {
        h->dispatch_timers.next(e0, e1, 1 << 30);
        hipExtLaunchKernelGGL(slim_ordered_kernel<T>, dim3(1), dim3(1024), 0, s, e0, e1, 0, p);
        h->stats.n_launches += 1;
        MI_HIP(hipGetLastError());
        h->steps_done += n;
        return;
    }
    // ticket numbers: sort the 2n (item, step) pairs, position inside the item's run
    ensure_sort_capacity(h, std::max<size_t>(2 * (size_t)n, 1024));
    DepParams d{};
    d.n_steps = n; d.n_items = h->n_items;
    d.indptr = h->indptr.ptr; d.indices = h->indices.ptr; d.su = h->su.ptr + first; d.si = h->si.ptr + first; d.sj = h->sj.ptr + first;
    d.keys = h->keys.ptr; d.vals = h->vals.ptr; d.keys_sorted = h->keys_sorted.ptr; d.vals_sorted = h->vals_sorted.ptr;
    d.seq = h->seq.ptr; d.len2 = h->len2.ptr; d.cellptr = h->cellptr.ptr; d.pred = h->pred.ptr;
    hipLaunchKernelGGL(slim_item_keys_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, d);
    size_t bytes = h->cub_tmp.count;
    MI_HIP(rocprim::radix_sort_pairs(h->cub_tmp.ptr, bytes, h->keys.ptr, h->keys_sorted.ptr, h->vals.ptr, h->vals_sorted.ptr,
                                              2 * n, 0, 32 + bits_for((unsigned long long)h->n_items), s));
    hipLaunchKernelGGL(slim_seq_kernel, dim3(div_up(2 * n, 256)), dim3(256), 0, s, d);
    // profile lengths -> cell slots (also the algorithmic byte count of the call)
    MI_HIP(hipMemsetAsync(h->len2.ptr + n, 0, sizeof(int), s));
    bytes = h->cub_tmp.count;
    MI_HIP(rocprim::exclusive_scan(h->cub_tmp.ptr, bytes, h->len2.ptr, h->cellptr.ptr, 0ll, (size_t)(n + 1), rocprim::plus<long long>(), s));
    long long n_cells = 0;
    MI_HIP(hipMemcpyAsync(&n_cells, h->cellptr.ptr + n, sizeof(long long), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    sum_profile += 0.5 * (double)n_cells;
    if (sym) {
        MI_REQUIRE(n_cells < (1ll << 31), "stream too long for the cell sort (%lld cells)", n_cells);
        ensure_sort_capacity(h, (size_t)std::max<long long>(n_cells, 1024));
        d.keys = h->keys.ptr; d.vals = h->vals.ptr; d.keys_sorted = h->keys_sorted.ptr; d.vals_sorted = h->vals_sorted.ptr;
        d.pred = h->pred.ptr;
        d.n_cells = n_cells;
        hipLaunchKernelGGL(slim_cell_keys_kernel, dim3(div_up(n, 4)), dim3(256), 0, s, d);
        bytes = h->cub_tmp.count;
        MI_HIP(rocprim::radix_sort_pairs(h->cub_tmp.ptr, bytes, h->keys.ptr, h->keys_sorted.ptr, h->vals.ptr,
                                                  h->vals_sorted.ptr, (int)n_cells, 0, 64, s));
        hipLaunchKernelGGL(slim_pred_kernel, dim3(div_up(n_cells, 256)), dim3(256), 0, s, d);
        fill_params(h, p);          // (the sort buffers may have been re-allocated)
        p.su += first; p.si += first; p.sj += first;
        p.n_steps = n;
        MI_HIP(hipMemsetAsync(h->done.ptr, 0, sizeof(int) * (size_t)n, s));
    }
    // the symmetric store orders steps per cell; item tickets are then only needed for the per-item optimiser cells
    p.use_tickets = !sym || h->cfg.sgd_mode != MI355REC_SGD;
    MI_HIP(hipMemsetAsync(h->ticket.ptr, 0, sizeof(int) * (size_t)h->n_items, s));
    MI_HIP(hipMemsetAsync(h->queue.ptr, 0, sizeof(int) * 2, s));
    const int grid = std::min(n, multiprocessor_count() * 4);
    h->dispatch_timers.next(e0, e1, 1 << 30);
    if (sym && getenv("MI355REC_SLIM_BATCHED")) hipExtLaunchKernelGGL((slim_flow_kernel<T, true, true>), dim3(grid), dim3(FLOW_THREADS), 0, s, e0, e1, 0, p);
    else if (sym) hipExtLaunchKernelGGL((slim_flow_kernel<T, true>), dim3(grid), dim3(FLOW_THREADS), 0, s, e0, e1, 0, p);
    else hipExtLaunchKernelGGL((slim_flow_kernel<T, false>), dim3(grid), dim3(FLOW_THREADS), 0, s, e0, e1, 0, p);
    h->stats.n_launches += 1;
    MI_HIP(hipGetLastError());
    int flags[2] = {0, 0};
    MI_HIP(hipMemcpyAsync(flags, h->queue.ptr, sizeof(flags), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (flags[1]) fail(MI355REC_E_HIP, "SLIM-BPR: a hand-off between steps did not arrive (dataflow kernel aborted)");
    h->steps_done += n;
}

void prune_rows(mi355rec_slim *h, int with_diag) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    h->dispatch_timers.next(e0, e1, 1 << 30);
    hipExtLaunchKernelGGL(slim_prune_kernel, dim3(std::min(h->n_items, multiprocessor_count() * 8)), dim3(PRUNE_THREADS), 0, h->stream,
                          e0, e1, 0, reinterpret_cast<unsigned long long *>(h->S.ptr), h->n_items, h->cfg.topK, with_diag);
    h->stats.n_launches += 1;
    MI_HIP(hipGetLastError());
}

// One epoch of n steps on whichever store the handle has.  Sparse store: the stream is cut after every step whose index is a
// positive multiple of n / 5 -- `numCurrentBatch % (totalNumberOfBatch/5) == 0 and numCurrentBatch != 0` with C integer
// division (.pyx:320-324; the module sets cdivision) -- and the rows are pruned there.
template <class T>
void run_epoch_stream(mi355rec_slim *h, int n, double &sum_profile) {
    if (!h->cfg.train_with_sparse_weights || n < 5) {
        run_stream<T>(h, n, sum_profile);
        return;
    }
    const int every = n / 5;
    int first = 0;
    while (first < n) {
        // steps first .. cut (inclusive) run, then the rows are pruned if `cut` is a rebalance point
        const int cut = std::max(1, (first + every - 1) / every) * every;      // next multiple of `every` at or after `first`, never step 0
        const int last = std::min(cut, n - 1);
        run_stream<T>(h, last - first + 1, sum_profile, first);
        if (cut <= n - 1) prune_rows(h, 0);
        first = last + 1;
    }
}

void begin_call(mi355rec_slim *h) {
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
    ensure_capacity(h, (size_t)n);
    begin_call(h);
    double sum_profile = 0;
    for (int e = 0; e < n_epochs; ++e) {
        SlimParams<T> p{};
        fill_params(h, p);
        p.n_steps = n;
        hipLaunchKernelGGL(slim_sample_kernel<T>, dim3(div_up(n, 256)), dim3(256), 0, h->stream, p, h->su.ptr, h->si.ptr, h->sj.ptr);
        run_epoch_stream<T>(h, n, sum_profile);
        h->epochs_done += 1;
    }
    end_call(h, (long long)n * n_epochs, sum_profile);
}

template <class T>
void get_topk_typed(mi355rec_slim *h, int topK, int *nbr_idx, float *nbr_val) {
    const int n_pad = (h->n_items + 3) & ~3;
    const size_t lds = (size_t)n_pad * 4 + (size_t)AUX_WORDS * 4;
    if (lds + 2048 > 160 * 1024)
        fail(MI355REC_E_UNSUPPORTED, "n_items = %d: a row of S does not fit the 160 KiB LDS for the top-K selection", h->n_items);
    DeviceBuffer<int> d_idx;
    DeviceBuffer<float> d_val;
    const size_t n_out = (size_t)h->n_items * topK;
    d_idx.alloc(n_out);
    d_val.alloc(n_out);
    SlimParams<T> p{};
    fill_params(h, p);
    auto k = slim_topk_kernel<T, 1024>;
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / (lds + 2048))));
    hipLaunchKernelGGL(k, dim3(std::min(h->n_items, multiprocessor_count() * per_cu)), dim3(1024), lds, h->stream, p, topK,
                       n_pad, d_idx.ptr, d_val.ptr);
    MI_HIP(hipGetLastError());
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
    fill_params(h, p);
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
        h->f64 = cfg->precision == MI355REC_F64;
        h->nnz = (size_t)indptr[n_users];
        MI_REQUIRE(h->nnz > 0, "URM has no interactions");
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->call_timer.init();
        h->dispatch_timers.reserve(64);
        hipStream_t s = h->stream;
        const size_t ts = h->f64 ? sizeof(double) : sizeof(float);
        h->indptr.upload(indptr, (size_t)n_users + 1, s);
        h->indices.upload(indices, h->nnz, s);
        // .pyx:129 dense n x n; Triangular_Matrix :1237-1254 the packed lower triangle
        const size_t n_cells = h->cfg.symmetric ? (size_t)n_items * ((size_t)n_items + 1) / 2 : (size_t)n_items * n_items;
        h->S.alloc_zero(n_cells * ts, s);
        if (h->cfg.train_with_sparse_weights)                        // .pyx:124: an empty tree per row
            hipLaunchKernelGGL(slim_no_nodes_kernel, dim3(multiprocessor_count() * 8), dim3(256), 0, s,
                               reinterpret_cast<unsigned long long *>(h->S.ptr), (size_t)n_items * n_items);
        h->c1.alloc_zero((size_t)n_items * ts, s);
        h->c2.alloc_zero((size_t)n_items * ts, s);
        h->ticket.alloc_zero((size_t)n_items, s);
        h->queue.alloc_zero(2, s);
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
        if (h->f64) run_epochs_typed<double>(h, n_epochs); else run_epochs_typed<float>(h, n_epochs);
    });
}

extern "C" int mi355rec_slim_run_samples(mi355rec_slim_t h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n) {
    return guarded([&] {
        MI_REQUIRE(h && u && i && j, "NULL argument");
        MI_REQUIRE(n >= 0 && n < (1ll << 30), "n out of range");
        ensure_device();
        if (n == 0) return;
        ensure_capacity(h, (size_t)n);
        hipStream_t s = h->stream;
        for (int64_t t = 0; t < n; ++t)
            MI_REQUIRE(u[t] >= 0 && u[t] < h->n_users && i[t] >= 0 && i[t] < h->n_items && j[t] >= 0 && j[t] < h->n_items,
                       "sample %lld out of range", (long long)t);
        MI_HIP(hipMemcpyAsync(h->su.ptr, u, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(h->si.ptr, i, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(h->sj.ptr, j, sizeof(int) * n, hipMemcpyHostToDevice, s));
        begin_call(h);
        double sum_profile = 0;
        // (the n samples are ONE epoch of n steps: that is what the sparse store's rebalance rule counts against)
        if (h->f64) run_epoch_stream<double>(h, (int)n, sum_profile); else run_epoch_stream<float>(h, (int)n, sum_profile);
        end_call(h, n, sum_profile);
    });
}

extern "C" int mi355rec_slim_get_last_samples(mi355rec_slim_t h, int32_t *u, int32_t *i, int32_t *j, int64_t cap, int64_t *n) {
    return guarded([&] {
        MI_REQUIRE(h && n, "NULL argument");
        ensure_device();
        const int64_t have = h->epochs_done > 0 ? h->n_users + 1 : 0;
        *n = have;
        const size_t m = (size_t)std::min<int64_t>(cap, have);
        if (u) h->su.download(u, m, h->stream);
        if (i) h->si.download(i, m, h->stream);
        if (j) h->sj.download(j, m, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
    });
}

extern "C" int mi355rec_slim_get_S_topk(mi355rec_slim_t h, int32_t topK, int32_t *nbr_idx, float *nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && nbr_idx && nbr_val, "NULL argument");
        MI_REQUIRE(topK >= 1, "topK must be >= 1 (use mi355rec_slim_get_S_dense for the full matrix)");
        ensure_device();
        topK = std::min(topK, h->n_items);
        if (topK > MAX_TOPK) fail(MI355REC_E_UNSUPPORTED, "topK = %d exceeds the in-LDS selection limit of %d", topK, MAX_TOPK);
        if (h->f64) get_topk_typed<double>(h, topK, nbr_idx, nbr_val); else get_topk_typed<float>(h, topK, nbr_idx, nbr_val);
    });
}

extern "C" int mi355rec_slim_get_S_sparse(mi355rec_slim_t h, int32_t *nbr_idx, float *nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && nbr_idx && nbr_val, "NULL argument");
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

extern "C" void mi355rec_slim_destroy(mi355rec_slim_t h) { delete h; }
