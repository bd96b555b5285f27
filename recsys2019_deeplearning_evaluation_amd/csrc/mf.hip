// mf.hip -- BPR-MF and FunkSVD mini-batch SGD epochs on MI355X (gfx950).
//
// Replaces MatrixFactorization/Cython/MatrixFactorization_Cython_Epoch.pyx (reference):
//   epochIteration_Cython_BPR_SGD :580-649, epochIteration_Cython_FUNK_SVD_SGD :286-361,
//   sampleBPR_Cython :940-985, sampleMSE_Cython :878-935, _add_*_sample_in_minibatch :737-766,
//   _apply_minibatch_updates_to_latent_factors :770-829, adaptive_gradient :835-873.
//
// Design.  The reference's mini-batch semantics make the B samples of a batch independent:
// every gradient is taken against start-of-batch factors, summed per row, and the sum is applied once.
// That maps to one sampling kernel per epoch and two kernels per mini-batch:
//   mf_sample_kernel one thread per sample of the epoch: counter-based RNG, user / positive draw, negative by
//                    rejection against the sorted CSR row (binary search); sampling does not depend on the
//                    factors, so it is taken off the per-batch critical path;
//   mf_grad_kernel   one 64-lane wavefront per sample: row gather, wavefront-shuffle dot product, gradient rows
//                    scattered into the fp32 accumulators with device-scope float atomics; the first toucher of
//                    a row (flag exchange, the reference's flag arrays) files it in the sample's own slot of the
//                    batch's touched list -- no shared counter, no same-address atomics;
//   mf_apply_kernel  one wavefront per list slot: mean over batch_size, optimiser step, += lr * step,
//                    accumulator and flag reset.
// A whole epoch (1 + 2*n_batches launches) is captured once into a hipGraph and replayed: all per-batch state
// (batch index -> RNG counter, Adam's beta^t) lives in device memory, so the launches carry no host arguments.
// There is no dense contraction here, hence no MFMA; the path is bound by row gather/scatter bandwidth and,
// at the reference's batch sizes (<= 1024), by the dependent-launch latency between mini-batches.
#include "common.h"
#include "sampling.cuh"

#include <memory>

namespace mi355rec {
namespace {

struct MfState {           // lives in device memory so that launches carry no per-batch host arguments
    long long grad_batch;  // number of gradient kernels run since create = 1-based index of the current mini-batch
    long long epoch;       // index (since create) of the epoch the next sampling kernel draws
    double beta_1_power, beta_2_power;   // AsySVD: Adam's running products (advanced once per step, .pyx:536-539)
    double asy_loss;
};

struct MfParams {
    int n_users, n_items, k, batch_size;
    int use_bias, sgd_mode, sample_negatives, algorithm_is_bpr;
    float lr, user_reg, item_reg, bias_reg, positive_reg, negative_reg, quota;
    float gamma, beta_1, beta_2, one_m_gamma, one_m_beta_1, one_m_beta_2;   // 1 - x formed in double on the host
    double beta_1_d, beta_2_d;
    unsigned long long seed;
    const int *indptr, *indices;
    const float *data;
    float *U, *V, *accU, *accV;
    float *bu, *bi, *mu, *acc_bu, *acc_bi;
    float *mu_slots;   // FunkSVD with bias: the global-bias gradient terms of a mini-batch, one partial sum per workgroup
    float *c1U, *c2U, *c1V, *c2V;        // optimiser state: c1 = cache / first moment, c2 = second moment
    float *c1_bu, *c2_bu, *c1_bi, *c2_bi, *c_mu;  // c_mu[0] = cache/m1, c_mu[1] = m2
    int *flag;                           // [n_users + n_items]
    int *list;                           // touched-row slots of the batch: 3 (BPR) / 2 (FunkSVD) per sample, -1 = not first toucher
    double *loss_slots;                  // [batch_size] per-sample-slot running loss (summed on the host)
    MfState *state;
    // sample stream: one epoch drawn by mf_sample_kernel (native) or the caller's stream (replay)
    int *su, *si, *sj;
    float *sr;
    long long samples_per_epoch;
    int n_in_batch;                      // samples in this launch's batch (<= batch_size; short only in replay)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// _add_*_sample_in_minibatch (.pyx:737-766): the first toucher of a row owns its apply
__device__ __forceinline__ void touch(const MfParams &p, int entry, int slot) {
    p.list[slot] = atomicExch(&p.flag[entry], 1) == 0 ? entry : -1;
}

// One thread per sample of one epoch (sampleBPR_Cython .pyx:940-985 / sampleMSE_Cython :878-935).
template <int ALGO>
__global__ __launch_bounds__(256) void mf_sample_kernel(const MfParams p) {
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
    // the grid is fully drained before the next kernel starts: a plain store by one thread is enough
    if (t == 0) p.state->epoch = epoch + 1;
}

// KI = ceil(k / 64) rows-in-registers specialisation (1..4); KI == 0: any k, rows are re-read for the scatter.
template <int ALGO, int KI>
__global__ __launch_bounds__(256) void mf_grad_kernel(const MfParams p, const int batch_local) {
    // batch_local = position of this mini-batch inside the stream buffer; it is a launch argument (baked into the
    // graph node), so the sample triplet is the first load of the kernel, not the second
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);   // sample slot inside the batch
    __shared__ float s_mu[4];
    float mu_term = 0.f;
    // count the mini-batches since create (Adam's beta^t): single writer here, read only by the apply kernel that follows
    if (blockIdx.x == 0 && threadIdx.x == 0) p.state->grad_batch += 1;
    double my_loss = 0.0;
    if (w < p.n_in_batch) {
        const long long slot = (long long)batch_local * p.batch_size + w;
        const int u = p.su[slot], i = p.si[slot];
        int j = -1;
        float rating = 0.f;
        if (ALGO == MI355REC_MF_BPR) j = p.sj[slot]; else rating = p.sr[slot];
        if (lane == 0) {
            constexpr int PER = ALGO == MI355REC_MF_BPR ? 3 : 2;
            touch(p, p.n_users + i, PER * w);
            if (ALGO == MI355REC_MF_BPR) touch(p, p.n_users + j, PER * w + 1);
            touch(p, u, PER * w + PER - 1);
        }
        const int k = p.k;
        const float *Wu = p.U + (size_t)u * k, *Hi = p.V + (size_t)i * k;
        float *aU = p.accU + (size_t)u * k, *aI = p.accV + (size_t)i * k;
        if (ALGO == MI355REC_MF_BPR) {
            const float *Hj = p.V + (size_t)j * k;
            float *aJ = p.accV + (size_t)j * k;
            float wu[KI ? KI : 1], hi[KI ? KI : 1], hj[KI ? KI : 1];
            float x = 0.f;
            if (KI) {
#pragma unroll
                for (int t = 0; t < KI; ++t) {
                    const int f = lane + 64 * t;
                    const bool ok = f < k;
                    wu[t] = ok ? Wu[f] : 0.f;
                    hi[t] = ok ? Hi[f] : 0.f;
                    hj[t] = ok ? Hj[f] : 0.f;
                    x += wu[t] * (hi[t] - hj[t]);
                }
            } else {
                for (int f = lane; f < k; f += 64) x += Wu[f] * (Hi[f] - Hj[f]);
            }
            x = wave_sum(x);
            const float s = 1.f / (1.f + __expf(x));          // gradient of log(sigm(-x_uij)), .pyx:619
            my_loss = (double)x * x;
            if (KI) {
#pragma unroll
                for (int t = 0; t < KI; ++t) {
                    const int f = lane + 64 * t;
                    if (f < k) {
                        atomicAdd(&aU[f], s * (hi[t] - hj[t]) - p.user_reg * wu[t]);
                        atomicAdd(&aI[f], s * wu[t] - p.positive_reg * hi[t]);
                        atomicAdd(&aJ[f], -s * wu[t] - p.negative_reg * hj[t]);
                    }
                }
            } else {
                for (int f = lane; f < k; f += 64) {
                    const float a = Wu[f], b = Hi[f], c = Hj[f];
                    atomicAdd(&aU[f], s * (b - c) - p.user_reg * a);
                    atomicAdd(&aI[f], s * a - p.positive_reg * b);
                    atomicAdd(&aJ[f], -s * a - p.negative_reg * c);
                }
            }
        } else {
            float wu[KI ? KI : 1], hi[KI ? KI : 1];
            float dot = 0.f;
            if (KI) {
#pragma unroll
                for (int t = 0; t < KI; ++t) {
                    const int f = lane + 64 * t;
                    const bool ok = f < k;
                    wu[t] = ok ? Wu[f] : 0.f;
                    hi[t] = ok ? Hi[f] : 0.f;
                    dot += wu[t] * hi[t];
                }
            } else {
                for (int f = lane; f < k; f += 64) dot += Wu[f] * Hi[f];
            }
            dot = wave_sum(dot);
            float pred = dot;
            if (p.use_bias) pred += p.mu[0] + p.bu[u] + p.bi[i];
            const float err = rating - pred;
            my_loss = (double)err * err;
            if (p.use_bias && lane == 0) {   // .pyx:329-336
                // the global bias has ONE accumulator: a thousand same-address atomics serialise at ~12 ns each (12 of
                // the 22 us of a mini-batch); the terms are summed per workgroup below and by the apply kernel instead
                mu_term = err - p.bias_reg * p.mu[0];
                atomicAdd(&p.acc_bi[i], err - p.bias_reg * p.bi[i]);
                atomicAdd(&p.acc_bu[u], err - p.bias_reg * p.bu[u]);
            }
            // NB the item gradient is regularised with positive_reg (sic, .pyx:346), never item_reg
            if (KI) {
#pragma unroll
                for (int t = 0; t < KI; ++t) {
                    const int f = lane + 64 * t;
                    if (f < k) {
                        atomicAdd(&aI[f], err * wu[t] - p.positive_reg * hi[t]);
                        atomicAdd(&aU[f], err * hi[t] - p.user_reg * wu[t]);
                    }
                }
            } else {
                for (int f = lane; f < k; f += 64) {
                    const float a = Wu[f], b = Hi[f];
                    atomicAdd(&aI[f], err * a - p.positive_reg * b);
                    atomicAdd(&aU[f], err * b - p.user_reg * a);
                }
            }
        }
    }
    if (lane == 0 && w < p.n_in_batch) p.loss_slots[w] += my_loss;   // slot w is private to this wavefront
    if (ALGO != MI355REC_MF_BPR && p.use_bias) {
        if (lane == 0) s_mu[threadIdx.x >> 6] = mu_term;
        __syncthreads();
        if (threadIdx.x == 0) p.mu_slots[blockIdx.x] = (s_mu[0] + s_mu[1]) + (s_mu[2] + s_mu[3]);
    }
}

// adaptive_gradient (.pyx:835-873) on one cell; pw1/pw2 = 1 - beta^t
__device__ __forceinline__ float adapt(const MfParams &p, float g, float *c1, float *c2, size_t at, float pw1, float pw2) {
    switch (p.sgd_mode) {
        case MI355REC_ADAGRAD: {
            float c = c1[at] + g * g;
            c1[at] = c;
            return g / (sqrtf(c) + 1e-8f);
        }
        case MI355REC_RMSPROP: {
            float c = c1[at] * p.gamma + p.one_m_gamma * (g * g);
            c1[at] = c;
            return g / (sqrtf(c) + 1e-8f);
        }
        case MI355REC_ADAM: {
            float m1 = c1[at] * p.beta_1 + p.one_m_beta_1 * g;
            float m2 = c2[at] * p.beta_2 + p.one_m_beta_2 * (g * g);
            c1[at] = m1;
            c2[at] = m2;
            return (m1 / pw1) / (sqrtf(m2 / pw2) + 1e-8f);
        }
        default:
            return g;
    }
}

// _apply_minibatch_updates_to_latent_factors (.pyx:770-829): one wavefront per touched row.
__global__ __launch_bounds__(256) void mf_apply_kernel(const MfParams p) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float invB = 1.f / (float)p.batch_size;      // mean over batch_size, NOT over the row's count (.pyx:802)
    float pw1 = 1.f, pw2 = 1.f;
    if (p.sgd_mode == MI355REC_ADAM) {                  // beta^(t) with t = batch + 1 (.pyx:217-218, :646-649)
        // the global mini-batch index lives in device memory (graph replays carry no host arguments); only Adam reads
        // it, so the other optimisers keep this dependent load off their critical path
        const long long t = p.state->grad_batch;        // advanced by the gradient kernel of this mini-batch: t = batch + 1
        pw1 = (float)(1.0 - pow(p.beta_1_d, (double)t));
        pw2 = (float)(1.0 - pow(p.beta_2_d, (double)t));
    }
    if (w == 0 && p.use_bias) {                         // global bias: sum of the gradient kernel's per-workgroup partials
        float acc_mu = 0.f;
        for (int q = lane; q < (p.n_in_batch + 3) / 4; q += 64) acc_mu += p.mu_slots[q];
        acc_mu = wave_sum(acc_mu);
        if (lane == 0) {
            float g = adapt(p, acc_mu * invB, p.c_mu, p.c_mu + 1, 0, pw1, pw2);
            p.mu[0] += p.lr * g;
        }
    }
    const int per = p.algorithm_is_bpr ? 3 : 2;
    if (w >= per * p.n_in_batch) return;
    const int entry = p.list[w];
    if (entry < 0) return;                              // this sample was not the first toucher of that row
    const bool is_item = entry >= p.n_users;
    const int row = is_item ? entry - p.n_users : entry;
    const int k = p.k;
    float *W = (is_item ? p.V : p.U) + (size_t)row * k;
    float *A = (is_item ? p.accV : p.accU) + (size_t)row * k;
    float *c1 = is_item ? p.c1V : p.c1U, *c2 = is_item ? p.c2V : p.c2U;
    for (int f = lane; f < k; f += 64) {
        float g = A[f] * invB;
        g = adapt(p, g, c1, c2, (size_t)row * k + f, pw1, pw2);
        W[f] += p.lr * g;
        A[f] = 0.f;
    }
    if (lane == 0) {
        if (p.use_bias) {
            float *b = is_item ? p.bi : p.bu, *ab = is_item ? p.acc_bi : p.acc_bu;
            float *b1 = is_item ? p.c1_bi : p.c1_bu, *b2 = is_item ? p.c2_bi : p.c2_bu;
            float g = adapt(p, ab[row] * invB, b1, b2, (size_t)row, pw1, pw2);
            b[row] += p.lr * g;
            ab[row] = 0.f;
        }
        p.flag[entry] = 0;
    }
}

// AsySVD (.pyx:393-541): batch_size is 1 and every step rewrites all the Y rows of the sampled user's profile, which
// nearly every other profile shares -- the steps are executed strictly in order by ONE 1024-thread workgroup (16
// wavefronts across the profile rows, lanes across the factors).  p.U is the n_items x k matrix Y ("USER_factors" in the
// reference), p.V the item factors X.
constexpr int ASY_KMAX = 256;
__global__ __launch_bounds__(1024) void mf_asy_kernel(const MfParams p, const long long first, const int count) {
    __shared__ float s_part[16][ASY_KMAX];
    __shared__ float s_acc[ASY_KMAX], s_xi[ASY_KMAX];
    __shared__ float s_err, s_pw1, s_pw2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = p.k;
    double b1p = 0.0, b2p = 0.0, loss = 0.0;
    if (tid == 0) {
        b1p = p.state->beta_1_power;
        b2p = p.state->beta_2_power;
        loss = p.state->asy_loss;
    }
    for (int s = 0; s < count; ++s) {
        const long long t = first + s;
        const int u = p.su[t], i = p.si[t];
        const float rating = p.sr[t];
        const int rs = p.indptr[u], re = p.indptr[u + 1];
        float *X = p.V + (size_t)i * k;
        for (int f = tid; f < k; f += 1024) s_xi[f] = X[f];
        // user vector: sum of the Y rows of the profile / sqrt(profile length)   (.pyx:424-441)
        float part[ASY_KMAX / 64];
#pragma unroll
        for (int c = 0; c < ASY_KMAX / 64; ++c) part[c] = 0.f;
        for (int q = rs + wave; q < re; q += 16) {
            const float *Y = p.U + (size_t)p.indices[q] * k;
#pragma unroll
            for (int c = 0; c < ASY_KMAX / 64; ++c) {
                const int f = lane + 64 * c;
                if (f < k) part[c] += Y[f];
            }
        }
#pragma unroll
        for (int c = 0; c < ASY_KMAX / 64; ++c) {
            const int f = lane + 64 * c;
            if (f < k) s_part[wave][f] = part[c];
        }
        __syncthreads();
        if (tid < k) {
            float a = 0.f;
            for (int w = 0; w < 16; ++w) a += s_part[w][tid];
            s_acc[tid] = a / sqrtf((float)(re - rs));
        }
        __syncthreads();
        if (wave == 0) {
            float dot = 0.f;
            for (int f = lane; f < k; f += 64) dot += s_acc[f] * s_xi[f];
            dot = wave_sum(dot);
            if (lane == 0) {
                float pred = dot;
                if (p.use_bias) pred += p.mu[0] + p.bu[u] + p.bi[i];
                const float err = rating - pred;
                loss += (double)err * err;
                const float pw1 = (float)(1.0 - b1p), pw2 = (float)(1.0 - b2p);
                if (p.use_bias) {       // global, item, user bias -- in that order (.pyx:458-490)
                    float g = adapt(p, err - p.bias_reg * p.mu[0], p.c_mu, p.c_mu + 1, 0, pw1, pw2);
                    p.mu[0] += p.lr * g;
                    g = adapt(p, err - p.bias_reg * p.bi[i], p.c1_bi, p.c2_bi, (size_t)i, pw1, pw2);
                    p.bi[i] += p.lr * g;
                    g = adapt(p, err - p.bias_reg * p.bu[u], p.c1_bu, p.c2_bu, (size_t)u, pw1, pw2);
                    p.bu[u] += p.lr * g;
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
        __syncthreads();
        const float err = s_err, pw1 = s_pw1, pw2 = s_pw2;
        // every Y row of the profile moves against the OLD X[i]   (.pyx:493-511)
        for (int q = rs + wave; q < re; q += 16) {
            const size_t row = (size_t)p.indices[q];
            float *Y = p.U + row * k;
            for (int f = lane; f < k; f += 64) {
                const float w = Y[f];
                const float g = adapt(p, err * s_xi[f] - p.user_reg * w, p.c1U, p.c2U, row * k + f, pw1, pw2);
                Y[f] = w + p.lr * g;
            }
        }
        // X[i] moves against the user vector formed BEFORE the Y update   (.pyx:514-531)
        if (tid < k) {
            const float h = s_xi[tid];
            const float g = adapt(p, err * s_acc[tid] - p.item_reg * h, p.c1V, p.c2V, (size_t)i * k + tid, pw1, pw2);
            X[tid] = h + p.lr * g;
        }
        __threadfence_block();
        __syncthreads();
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
    size_t nnz = 0;
    hipStream_t stream = nullptr;
    StreamTimer timer;
    DeviceBuffer<int> indptr, indices, flag, list, su, si, sj;
    DeviceBuffer<float> data, U, V, accU, accV, bu, bi, mu, acc_bu, acc_bi, mu_slots;
    DeviceBuffer<float> c1U, c2U, c1V, c2V, c1_bu, c2_bu, c1_bi, c2_bi, c_mu, sr;
    DeviceBuffer<double> loss_slots;
    DeviceBuffer<MfState> state;
    long long batches_done = 0;      // batches executed since create (device state mirrors this)
    long long last_call_samples = 0; // samples in the stream buffer after the last native call
    size_t stream_capacity = 0;
    mi355rec_stats stats{};
    DispatchTimers dispatch_timers;
    int max_timed = 0;
    hipGraphExec_t epoch_graph = nullptr;   // one native epoch: sample kernel + n_batches x (grad, apply)
    std::vector<double> host_loss;

    ~mi355rec_mf() {
        if (stream) (void)hipStreamSynchronize(stream);
        if (epoch_graph) (void)hipGraphExecDestroy(epoch_graph);
        timer.destroy();
        dispatch_timers.destroy();
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

long long batches_per_epoch(const mi355rec_mf *h) {
    // .pyx:583 (BPR: n_users / B + 1) and :289 (FunkSVD: nnz / B + 1)
    const long long B = h->cfg.batch_size;
    // ASY_SVD: nnz / 1 + 1 single-sample steps (.pyx:397)
    return (h->cfg.algorithm == MI355REC_MF_BPR ? (long long)h->n_users / B : (long long)h->nnz / B) + 1;
}

void fill_params(mi355rec_mf *h, MfParams &p) {
    const auto &c = h->cfg;
    p.n_users = h->n_users; p.n_items = h->n_items; p.k = h->k; p.batch_size = c.batch_size;
    p.use_bias = c.use_bias && c.algorithm != MI355REC_MF_BPR;
    p.sgd_mode = c.sgd_mode;
    p.algorithm_is_bpr = c.algorithm == MI355REC_MF_BPR;
    p.sample_negatives = c.negative_interactions_quota != 0.0;
    p.lr = (float)c.learning_rate; p.user_reg = (float)c.user_reg; p.item_reg = (float)c.item_reg; p.bias_reg = (float)c.bias_reg;
    p.positive_reg = (float)c.positive_reg; p.negative_reg = (float)c.negative_reg;
    p.quota = (float)c.negative_interactions_quota;
    p.gamma = (float)c.gamma; p.beta_1 = (float)c.beta_1; p.beta_2 = (float)c.beta_2;
    p.one_m_gamma = (float)(1.0 - c.gamma); p.one_m_beta_1 = (float)(1.0 - c.beta_1); p.one_m_beta_2 = (float)(1.0 - c.beta_2);
    p.beta_1_d = c.beta_1; p.beta_2_d = c.beta_2;
    p.seed = c.random_seed;
    p.indptr = h->indptr.ptr; p.indices = h->indices.ptr; p.data = h->data.ptr;
    p.U = h->U.ptr; p.V = h->V.ptr; p.accU = h->accU.ptr; p.accV = h->accV.ptr;
    p.bu = h->bu.ptr; p.bi = h->bi.ptr; p.mu = h->mu.ptr;
    p.acc_bu = h->acc_bu.ptr; p.acc_bi = h->acc_bi.ptr; p.mu_slots = h->mu_slots.ptr;
    p.c1U = h->c1U.ptr; p.c2U = h->c2U.ptr; p.c1V = h->c1V.ptr; p.c2V = h->c2V.ptr;
    p.c1_bu = h->c1_bu.ptr; p.c2_bu = h->c2_bu.ptr; p.c1_bi = h->c1_bi.ptr; p.c2_bi = h->c2_bi.ptr; p.c_mu = h->c_mu.ptr;
    p.flag = h->flag.ptr; p.list = h->list.ptr; p.loss_slots = h->loss_slots.ptr; p.state = h->state.ptr;
    p.su = h->su.ptr; p.si = h->si.ptr; p.sj = h->sj.ptr; p.sr = h->sr.ptr;
    p.samples_per_epoch = batches_per_epoch(h) * (long long)c.batch_size;
    p.n_in_batch = c.batch_size;
}

template <int ALGO, int KI>
void launch_grad(mi355rec_mf *h, const MfParams &p, int grid, int batch_local, hipEvent_t e0, hipEvent_t e1) {
    if (e0) hipExtLaunchKernelGGL((mf_grad_kernel<ALGO, KI>), dim3(grid), dim3(256), 0, h->stream, e0, e1, 0, p, batch_local);
    else hipLaunchKernelGGL((mf_grad_kernel<ALGO, KI>), dim3(grid), dim3(256), 0, h->stream, p, batch_local);   // capturable
}

template <int ALGO>
void launch_grad_ki(mi355rec_mf *h, const MfParams &p, int grid, int batch_local, bool timed) {
    const int ki = h->k <= 256 ? (h->k + 63) / 64 : 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) h->dispatch_timers.next(e0, e1, h->max_timed);
    switch (ki) {
        case 1: launch_grad<ALGO, 1>(h, p, grid, batch_local, e0, e1); break;
        case 2: launch_grad<ALGO, 2>(h, p, grid, batch_local, e0, e1); break;
        case 3: launch_grad<ALGO, 3>(h, p, grid, batch_local, e0, e1); break;
        case 4: launch_grad<ALGO, 4>(h, p, grid, batch_local, e0, e1); break;
        default: launch_grad<ALGO, 0>(h, p, grid, batch_local, e0, e1); break;
    }
}

void launch_batch(mi355rec_mf *h, const MfParams &p, int batch_local, bool timed) {
    const int grad_grid = div_up(p.n_in_batch, 4);
    const bool bpr = h->cfg.algorithm == MI355REC_MF_BPR;
    if (bpr) launch_grad_ki<MI355REC_MF_BPR>(h, p, grad_grid, batch_local, timed);
    else launch_grad_ki<MI355REC_MF_FUNK_SVD>(h, p, grad_grid, batch_local, timed);
    const int slots = (bpr ? 3 : 2) * p.n_in_batch;
    hipLaunchKernelGGL(mf_apply_kernel, dim3(div_up(slots, 4)), dim3(256), 0, h->stream, p);
}

void launch_sampler(mi355rec_mf *h, const MfParams &p) {
    const int grid = div_up(p.samples_per_epoch, 256);
    if (h->cfg.algorithm == MI355REC_MF_BPR) hipLaunchKernelGGL(mf_sample_kernel<MI355REC_MF_BPR>, dim3(grid), dim3(256), 0, h->stream, p);
    else hipLaunchKernelGGL(mf_sample_kernel<MI355REC_MF_FUNK_SVD>, dim3(grid), dim3(256), 0, h->stream, p);
}

// One native epoch as plain launches (the first `timed` gradient launches carry per-dispatch events).
constexpr int ASY_CHUNK = 1 << 16;   // steps per launch of the ordered AsySVD kernel (keeps single launches short)

void enqueue_asy_steps(mi355rec_mf *h, const MfParams &p, long long n_steps, bool timed) {
    for (long long first = 0; first < n_steps; first += ASY_CHUNK) {
        const int count = (int)std::min<long long>(ASY_CHUNK, n_steps - first);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) h->dispatch_timers.next(e0, e1, h->max_timed);
        if (e0) hipExtLaunchKernelGGL(mf_asy_kernel, dim3(1), dim3(1024), 0, h->stream, e0, e1, 0, p, first, count);
        else hipLaunchKernelGGL(mf_asy_kernel, dim3(1), dim3(1024), 0, h->stream, p, first, count);
    }
}

void enqueue_epoch(mi355rec_mf *h, const MfParams &p, bool timed) {
    launch_sampler(h, p);
    if (h->cfg.algorithm == MI355REC_MF_ASY_SVD) {
        enqueue_asy_steps(h, p, p.samples_per_epoch, timed);
        return;
    }
    const long long nb = batches_per_epoch(h);
    for (long long b = 0; b < nb; ++b) launch_batch(h, p, (int)b, timed);
}

// Capture one epoch into a graph (once per handle; re-captured only if the stream buffers are re-allocated).
constexpr long long MAX_GRAPH_BATCHES = 4096;
void ensure_epoch_graph(mi355rec_mf *h, const MfParams &p) {
    if (h->epoch_graph) return;
    if (h->epoch_graph) {
        (void)hipGraphExecDestroy(h->epoch_graph);
        h->epoch_graph = nullptr;
    }
    hipGraph_t g = nullptr;
    MI_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    enqueue_epoch(h, p, false);
    MI_HIP(hipStreamEndCapture(h->stream, &g));
    MI_HIP(hipGraphInstantiate(&h->epoch_graph, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
}

void ensure_stream_capacity(mi355rec_mf *h, size_t n) {
    if (h->stream_capacity >= n) return;
    if (h->epoch_graph) {   // the graph holds the old buffer addresses
        (void)hipGraphExecDestroy(h->epoch_graph);
        h->epoch_graph = nullptr;
    }
    h->su.alloc(n);
    h->si.alloc(n);
    h->sj.alloc(n);
    h->sr.alloc(n);
    h->stream_capacity = n;
}

double bytes_per_sample(const mi355rec_mf *h) {
    // ALGORITHMIC lower bound of DESIGN.md section 4: every row a sample touches is read once and written once
    // (3 rows for BPR, 2 for FunkSVD), fp32.
    const double rows = h->cfg.algorithm == MI355REC_MF_BPR ? 3.0 : 2.0;   // (AsySVD: its profile-sized term is added per call)
    return rows * 2.0 * 4.0 * (double)h->k;
}

void begin_call(mi355rec_mf *h) {
    MI_HIP(hipMemsetAsync(h->loss_slots.ptr, 0, sizeof(double) * h->cfg.batch_size, h->stream));
    MI_HIP(hipMemsetAsync(&h->state.ptr->asy_loss, 0, sizeof(double), h->stream));
    h->dispatch_timers.reset();
}

void finish_call(mi355rec_mf *h, long long n_samples, long long n_batches) {
    MI_HIP(hipGetLastError());
    h->host_loss.resize(h->cfg.batch_size);
    h->loss_slots.download(h->host_loss.data(), h->cfg.batch_size, h->stream);
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
    h->stats.n_launches = n_batches;            // launches of the dominant (gradient) kernel
    h->stats.n_units = n_samples;
    h->stats.algorithmic_bytes = bytes_per_sample(h) * (double)n_samples;
    h->stats.algorithmic_flops = 0;
    h->stats.loss = loss;
}

}  // namespace

extern "C" int mi355rec_mf_create(mi355rec_mf_t *out, const mi355rec_mf_config *cfg, int32_t n_users, int32_t n_items,
                                  const int32_t *indptr, const int32_t *indices, const float *data, const float *U0,
                                  const float *V0) {
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
        MI_REQUIRE(cfg->batch_size >= 1, "batch_size must be >= 1");
        ensure_device();
        std::unique_ptr<mi355rec_mf> h(new mi355rec_mf());
        h->cfg = *cfg;
        h->n_users = n_users;
        h->n_items = n_items;
        h->k = cfg->n_factors;
        h->n_u_rows = asy ? n_items : n_users;
        h->nnz = (size_t)indptr[n_users];
        MI_REQUIRE(h->nnz > 0, "URM has no interactions");
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->timer.init();
        hipStream_t s = h->stream;
        const size_t nu = (size_t)h->n_u_rows * h->k, ni = (size_t)n_items * h->k;
        h->indptr.upload(indptr, (size_t)n_users + 1, s);
        h->indices.upload(indices, h->nnz, s);
        h->data.upload(data, h->nnz, s);
        h->U.upload(U0, nu, s);
        h->V.upload(V0, ni, s);
        h->accU.alloc_zero(nu, s);
        h->accV.alloc_zero(ni, s);
        h->bu.alloc_zero(n_users, s);
        h->bi.alloc_zero(n_items, s);
        h->mu.alloc_zero(1, s);
        h->acc_bu.alloc_zero(n_users, s);
        h->acc_bi.alloc_zero(n_items, s);
        h->mu_slots.alloc_zero((size_t)(cfg->batch_size + 3) / 4 + 1, s);
        if (cfg->sgd_mode != MI355REC_SGD) {
            h->c1U.alloc_zero(nu, s);
            h->c1V.alloc_zero(ni, s);
            h->c1_bu.alloc_zero(n_users, s);
            h->c1_bi.alloc_zero(n_items, s);
            h->c_mu.alloc_zero(2, s);
            if (cfg->sgd_mode == MI355REC_ADAM) {
                h->c2U.alloc_zero(nu, s);
                h->c2V.alloc_zero(ni, s);
                h->c2_bu.alloc_zero(n_users, s);
                h->c2_bi.alloc_zero(n_items, s);
            }
        }
        h->flag.alloc_zero((size_t)n_users + n_items, s);
        h->list.alloc((size_t)cfg->batch_size * 3);
        h->loss_slots.alloc_zero((size_t)cfg->batch_size, s);
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
        const long long B = h->cfg.batch_size;
        const long long per_epoch = batches_per_epoch(h);
        const long long n_batches = per_epoch * n_epochs;
        ensure_stream_capacity(h, (size_t)(per_epoch * B));
        MfParams p{};
        fill_params(h, p);
        begin_call(h);
        // epochs whose gradient launches carry timing events run as plain launches, the rest replays the graph
        const long long timed_epochs = h->max_timed > 0 ? std::min<long long>(n_epochs, (h->max_timed + per_epoch - 1) / per_epoch) : 0;
        // MI355REC_NO_GRAPH=1: plain launches only (rocprofv3 on ROCm 7.2 crashes while tracing graph replays)
        const bool use_graph = per_epoch <= MAX_GRAPH_BATCHES && n_epochs - timed_epochs > 0 && !getenv("MI355REC_NO_GRAPH") &&
                               h->cfg.algorithm != MI355REC_MF_ASY_SVD;
        if (use_graph) ensure_epoch_graph(h, p);
        h->timer.start(h->stream);
        for (long long e = 0; e < n_epochs; ++e) {
            if (e < timed_epochs || !use_graph) enqueue_epoch(h, p, e < timed_epochs);
            else MI_HIP(hipGraphLaunch(h->epoch_graph, h->stream));
        }
        h->timer.stop(h->stream);
        h->batches_done += n_batches;
        h->last_call_samples = n_epochs > 0 ? per_epoch * B : 0;
        finish_call(h, n_batches * B, n_batches);
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
        ensure_stream_capacity(h, (size_t)std::max<long long>(n, batches_per_epoch(h) * B));
        hipStream_t s = h->stream;
        MI_HIP(hipMemcpyAsync(h->su.ptr, u, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(h->si.ptr, i, sizeof(int) * n, hipMemcpyHostToDevice, s));
        if (bpr) MI_HIP(hipMemcpyAsync(h->sj.ptr, j, sizeof(int) * n, hipMemcpyHostToDevice, s));
        else MI_HIP(hipMemcpyAsync(h->sr.ptr, rating, sizeof(float) * n, hipMemcpyHostToDevice, s));
        h->last_call_samples = 0;
        MfParams p{};
        fill_params(h, p);
        const long long n_batches = (n + B - 1) / B;
        begin_call(h);
        h->timer.start(s);
        if (h->cfg.algorithm == MI355REC_MF_ASY_SVD) {
            enqueue_asy_steps(h, p, n, true);
            h->timer.stop(s);
            h->batches_done += n;
            finish_call(h, n, (n + ASY_CHUNK - 1) / ASY_CHUNK);
            return;
        }
        for (long long b = 0; b < n_batches; ++b) {
            p.n_in_batch = (int)std::min<long long>(B, n - b * B);
            launch_batch(h, p, (int)b, true);
        }
        h->timer.stop(s);
        h->batches_done += n_batches;
        finish_call(h, n, n_batches);
    });
}

extern "C" int mi355rec_mf_get_factors(mi355rec_mf_t h, float *U, float *V, float *bu, float *bi, float *mu) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        hipStream_t s = h->stream;
        if (U) h->U.download(U, (size_t)h->n_u_rows * h->k, s);
        if (V) h->V.download(V, (size_t)h->n_items * h->k, s);
        if (bu) h->bu.download(bu, h->n_users, s);
        if (bi) h->bi.download(bi, h->n_items, s);
        if (mu) h->mu.download(mu, 1, s);
        MI_HIP(hipStreamSynchronize(s));
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

extern "C" int mi355rec_mf_get_stats(mi355rec_mf_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_mf_destroy(mi355rec_mf_t h) { delete h; }
