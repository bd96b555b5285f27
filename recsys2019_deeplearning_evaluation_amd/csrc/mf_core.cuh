// mf_core.cuh -- what the kernels of mf.hip share: device state, task headers, kernel parameters, the optimiser step and
// the rounding-exact gradient helpers, the sampler (round 4: mf.hip split by stage; one translation unit, see mf.hip).
#pragma once

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

}  // namespace
}  // namespace mi355rec
