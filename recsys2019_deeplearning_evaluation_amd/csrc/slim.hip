// slim.hip -- SLIM-BPR epoch on MI355X (gfx950).
//
// Replaces SLIM_BPR/Cython/SLIM_BPR_Cython_Epoch.pyx (reference): epochIteration_Cython :212-317 with the dense
// store (:129) and the symmetric triangular store (Triangular_Matrix :1226-1330), sampleBPR_Cython :439-483,
// adaptive_gradient :398-436, get_S :343-391.  The wrapper hard-codes batch_size = 1
// (SLIM_BPR/Cython/SLIM_BPR_Cython.py:140), so an epoch is n_users + 1 STRICTLY ORDERED SGD steps.
//
// Design.  S is a dense fp32 n_items x n_items matrix in HBM (2.86 GB at ML-20M shape; the
// symmetric store uses the lower triangle of the same array).  The sample stream of an epoch does not depend on
// S and is drawn up front by one kernel.  Exact sequential semantics are kept in two ways:
//   dense (asymmetric) store  step t touches only rows i_t and j_t of S and the two per-item optimiser cells, so
//                             steps with disjoint {i, j} commute.  The host level-schedules the epoch (level(t) =
//                             1 + max level of the previous step on row i_t / j_t) and every level runs as one
//                             launch, one 64-lane wavefront per step;
//   symmetric store           cell (r, c) aliases (c, r): conflicts are per cell, and almost every step touches a
//                             popular row through its profile.  This round the epoch runs as ONE persistent
//                             workgroup executing the steps in order, 1024 lanes across the profile.
// Both are gather/scatter of 4-byte cells (2 L_u reads + 2 L_u writes per step): HBM/L2-latency bound, no MFMA.
#include "common.h"
#include "sampling.cuh"
#include "topk.cuh"

#include <algorithm>
#include <memory>
#include <numeric>

namespace mi355rec {
namespace {

struct SlimParams {
    int n_users, n_items, symmetric, sgd_mode;
    float lr, li_reg, lj_reg, gamma, beta_1, beta_2, one_m_gamma, one_m_beta_1, one_m_beta_2;
    unsigned long long seed;
    const int *indptr, *indices;
    float *S;
    float *c1, *c2;                 // per-ITEM optimiser scalars (.pyx:177-181): cache / first moment, second moment
    const int *su, *si, *sj;        // sample stream of the call
    const float *pw1, *pw2;         // 1 - beta^t for every step of the stream (running product formed on the host in double)
    const int *order;               // level schedule: step ids grouped by level
    double *loss_slots;             // [LOSS_SLOTS]
    long long epoch;                // RNG counter base
    int n_steps;
};
constexpr int LOSS_SLOTS = 1024;

__device__ __forceinline__ size_t cell_at(const SlimParams &p, int r, int c) {
    // Triangular_Matrix.get_value/add_value (.pyx:1290-1330): in symmetric mode (r, c) with c > r lives at (c, r)
    if (p.symmetric && c > r) { const int t = r; r = c; c = t; }
    return (size_t)r * p.n_items + c;
}

// per-ITEM adaptive step (.pyx:398-436); pw1 / pw2 = 1 - beta^t of this step
__device__ __forceinline__ float slim_adapt(const SlimParams &p, float g, int item, float pw1, float pw2) {
    switch (p.sgd_mode) {
        case MI355REC_ADAGRAD: {
            const float c = p.c1[item] + g * g;
            p.c1[item] = c;
            return g / (sqrtf(c) + 1e-8f);
        }
        case MI355REC_RMSPROP: {
            const float c = p.c1[item] * p.gamma + p.one_m_gamma * (g * g);
            p.c1[item] = c;
            return g / (sqrtf(c) + 1e-8f);
        }
        case MI355REC_ADAM: {
            const float m1 = p.c1[item] * p.beta_1 + p.one_m_beta_1 * g;
            const float m2 = p.c2[item] * p.beta_2 + p.one_m_beta_2 * (g * g);
            p.c1[item] = m1;
            p.c2[item] = m2;
            return (m1 / pw1) / (sqrtf(m2 / pw2) + 1e-8f);
        }
        default:
            return g;
    }
}

__global__ __launch_bounds__(256) void slim_sample_kernel(SlimParams p, int *su, int *si, int *sj) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= p.n_steps) return;
    int u, i, j;
    sample_bpr(p.seed, (unsigned long long)(p.epoch * (long long)p.n_steps + t), p.n_users, p.n_items, p.indptr, p.indices, u, i, j);
    su[t] = u;
    si[t] = i;
    sj[t] = j;
}

// One SGD step (.pyx:243-317) by LANES cooperating lanes; `reduce` sums x over them.
template <int LANES, class Reduce>
__device__ __forceinline__ void slim_step(const SlimParams &p, int t, int lane, Reduce reduce) {
    const int u = p.su[t], i = p.si[t], j = p.sj[t];
    const int rs = p.indptr[u], re = p.indptr[u + 1];
    float x = 0.f;
    for (int q = rs + lane; q < re; q += LANES) {
        const int s = p.indices[q];
        x += p.S[cell_at(p, i, s)] - p.S[cell_at(p, j, s)];
    }
    float gi = 0.f, gj = 0.f;
    x = reduce(x, i, j, gi, gj, t);          // also turns x into the two per-item steps (one lane updates the caches)
    for (int q = rs + lane; q < re; q += LANES) {
        const int s = p.indices[q];
        if (s != i) {
            float *c = &p.S[cell_at(p, i, s)];
            const float v = *c;
            *c = v + p.lr * (gi - p.li_reg * v);
        }
        if (s != j) {
            float *c = &p.S[cell_at(p, j, s)];
            const float v = *c;
            *c = v - p.lr * (gj - p.lj_reg * v);
        }
    }
}

// Level-parallel path (dense store): one wavefront per step of the level.
__global__ __launch_bounds__(256) void slim_level_kernel(const SlimParams p, int first, int count) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= count) return;
    const int t = p.order[first + w];
    auto reduce = [&](float x, int i, int j, float &gi, float &gj, int step) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        const float g = 1.f / (1.f + __expf(x));
        float a = 0.f, b = 0.f;
        if (lane == 0) {
            a = slim_adapt(p, g, i, p.pw1[step], p.pw2[step]);     // item i first, then j, as .pyx:267-268
            b = slim_adapt(p, g, j, p.pw1[step], p.pw2[step]);
            atomicAdd(&p.loss_slots[step & (LOSS_SLOTS - 1)], (double)x * x);   // steps of one level may share a slot
        }
        gi = __shfl(a, 0);
        gj = __shfl(b, 0);
        return x;
    };
    slim_step<64>(p, t, lane, reduce);
}

// Level-parallel path, one WORKGROUP per step: a level lasts as long as its longest profile, and a single wavefront
// walks a 2000-item profile in 2 x 32 dependent gather rounds (~1 us each); THREADS lanes do it in a few.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void slim_level_wg_kernel(const SlimParams p, int first, int count) {
    __shared__ float s_part[THREADS / 64];
    __shared__ float s_g[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= count) return;
    const int t = p.order[first + blockIdx.x];
    auto reduce = [&](float x, int i, int j, float &gi, float &gj, int step) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) s_part[wave] = x;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < THREADS / 64; ++w) tot += s_part[w];
            const float g = 1.f / (1.f + __expf(tot));
            s_g[0] = slim_adapt(p, g, i, p.pw1[step], p.pw2[step]);     // item i first, then j, as .pyx:267-268
            s_g[1] = slim_adapt(p, g, j, p.pw1[step], p.pw2[step]);
            atomicAdd(&p.loss_slots[step & (LOSS_SLOTS - 1)], (double)tot * tot);   // steps of one level may share a slot
        }
        __syncthreads();
        gi = s_g[0];
        gj = s_g[1];
        return x;
    };
    slim_step<THREADS>(p, t, tid, reduce);
}

// Ordered path (any store): one workgroup runs steps [0, n_steps) one after the other.
__global__ __launch_bounds__(1024) void slim_ordered_kernel(const SlimParams p) {
    __shared__ float s_part[16];
    __shared__ float s_g[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = 0; t < p.n_steps; ++t) {
        auto reduce = [&](float x, int i, int j, float &gi, float &gj, int step) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0) s_part[wave] = x;
            __syncthreads();
            if (tid == 0) {
                float tot = 0.f;
                for (int w = 0; w < 16; ++w) tot += s_part[w];
                const float g = 1.f / (1.f + __expf(tot));
                s_g[0] = slim_adapt(p, g, i, p.pw1[step], p.pw2[step]);
                s_g[1] = slim_adapt(p, g, j, p.pw1[step], p.pw2[step]);
                p.loss_slots[step & (LOSS_SLOTS - 1)] += (double)tot * tot;
            }
            __syncthreads();
            gi = s_g[0];
            gj = s_g[1];
            return x;
        };
        slim_step<1024>(p, t, tid, reduce);
        __threadfence_block();       // the next step of this workgroup must read what this one wrote
        __syncthreads();
    }
}

// get_S (.pyx:343-391): row r of S with the diagonal zeroed (symmetric store mirrored), then the per-row top-K.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void slim_topk_kernel(const SlimParams p, int topK, int n_pad, int *out_idx,
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
            const float v = c == r ? 0.f : p.S[cell_at(p, r, c)];
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

__global__ void slim_dense_kernel(const SlimParams p, float *out) {
    const size_t n = (size_t)p.n_items;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n * n; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / n), c = (int)(e % n);
        out[e] = r == c ? 0.f : p.S[cell_at(p, r, c)];
    }
}

}  // namespace
}  // namespace mi355rec

using namespace mi355rec;

struct mi355rec_slim {
    mi355rec_slim_config cfg{};
    int n_users = 0, n_items = 0;
    size_t nnz = 0;
    hipStream_t stream = nullptr;
    StreamTimer call_timer;
    DispatchTimers dispatch_timers;
    DeviceBuffer<int> indptr, indices, su, si, sj, order;
    DeviceBuffer<float> S, c1, c2, pw1, pw2;
    DeviceBuffer<double> loss_slots;
    size_t stream_capacity = 0;
    long long steps_done = 0, epochs_done = 0;
    double beta_1_power = 0, beta_2_power = 0;       // running products, like the reference's beta_*_power_t
    std::vector<int> h_u, h_i, h_j, h_order, h_level_ptr, last_level;
    std::vector<int> indptr_host;
    std::vector<float> h_pw1, h_pw2;
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

void fill_params(mi355rec_slim *h, SlimParams &p) {
    const auto &c = h->cfg;
    p.n_users = h->n_users; p.n_items = h->n_items; p.symmetric = c.symmetric; p.sgd_mode = c.sgd_mode;
    p.lr = (float)c.learning_rate; p.li_reg = (float)c.li_reg; p.lj_reg = (float)c.lj_reg;
    p.gamma = (float)c.gamma; p.beta_1 = (float)c.beta_1; p.beta_2 = (float)c.beta_2;
    p.one_m_gamma = (float)(1.0 - c.gamma); p.one_m_beta_1 = (float)(1.0 - c.beta_1); p.one_m_beta_2 = (float)(1.0 - c.beta_2);
    p.seed = c.random_seed;
    p.indptr = h->indptr.ptr; p.indices = h->indices.ptr;
    p.S = h->S.ptr; p.c1 = h->c1.ptr; p.c2 = h->c2.ptr;
    p.su = h->su.ptr; p.si = h->si.ptr; p.sj = h->sj.ptr;
    p.pw1 = h->pw1.ptr; p.pw2 = h->pw2.ptr;
    p.order = h->order.ptr;
    p.loss_slots = h->loss_slots.ptr;
    p.epoch = h->epochs_done;
    p.n_steps = 0;
}

void ensure_capacity(mi355rec_slim *h, size_t n) {
    if (h->stream_capacity >= n) return;
    h->su.alloc(n); h->si.alloc(n); h->sj.alloc(n); h->order.alloc(n);
    h->pw1.alloc(n); h->pw2.alloc(n);
    h->stream_capacity = n;
}

// Runs the steps currently in su/si/sj (n of them) exactly in order; host copies of the stream are in h_u/h_i/h_j.
void run_stream(mi355rec_slim *h, int n, double &sum_profile) {
    hipStream_t s = h->stream;
    // Adam's bias corrections: the reference multiplies beta_power by beta after every sample (.pyx:313-317)
    h->h_pw1.resize(n); h->h_pw2.resize(n);
    for (int t = 0; t < n; ++t) {
        h->h_pw1[t] = (float)(1.0 - h->beta_1_power);
        h->h_pw2[t] = (float)(1.0 - h->beta_2_power);
        if (h->cfg.sgd_mode == MI355REC_ADAM) {
            h->beta_1_power *= h->cfg.beta_1;
            h->beta_2_power *= h->cfg.beta_2;
        }
    }
    MI_HIP(hipMemcpyAsync(h->pw1.ptr, h->h_pw1.data(), sizeof(float) * n, hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(h->pw2.ptr, h->h_pw2.data(), sizeof(float) * n, hipMemcpyHostToDevice, s));
    SlimParams p{};
    fill_params(h, p);
    p.n_steps = n;
    for (int t = 0; t < n; ++t) sum_profile += h->indptr_host[h->h_u[t] + 1] - h->indptr_host[h->h_u[t]];
    if (h->cfg.symmetric) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        h->dispatch_timers.next(e0, e1, 1 << 30);
        hipExtLaunchKernelGGL(slim_ordered_kernel, dim3(1), dim3(1024), 0, s, e0, e1, 0, p);
        h->stats.n_launches += 1;
    } else {
        // level schedule: a step may run once the previous steps on its two rows are done
        h->last_level.assign(h->n_items, 0);
        std::vector<int> level(n);
        int n_levels = 0;
        for (int t = 0; t < n; ++t) {
            const int l = 1 + std::max(h->last_level[h->h_i[t]], h->last_level[h->h_j[t]]);
            level[t] = l;
            h->last_level[h->h_i[t]] = l;
            h->last_level[h->h_j[t]] = l;
            n_levels = std::max(n_levels, l);
        }
        h->h_level_ptr.assign(n_levels + 2, 0);
        for (int t = 0; t < n; ++t) h->h_level_ptr[level[t] + 1]++;
        for (int l = 1; l <= n_levels + 1; ++l) h->h_level_ptr[l] += h->h_level_ptr[l - 1];
        h->h_order.resize(n);
        std::vector<int> cursor(h->h_level_ptr.begin(), h->h_level_ptr.end());
        for (int t = 0; t < n; ++t) h->h_order[cursor[level[t]]++] = t;     // stable: stream order inside a level
        MI_HIP(hipMemcpyAsync(h->order.ptr, h->h_order.data(), sizeof(int) * n, hipMemcpyHostToDevice, s));
        const bool wave_levels = getenv("MI355REC_SLIM_WAVE_LEVELS") != nullptr;
        for (int l = 1; l <= n_levels; ++l) {
            const int first = h->h_level_ptr[l], count = h->h_level_ptr[l + 1] - first;
            if (count == 0) continue;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            const bool timed = h->dispatch_timers.next(e0, e1, 512);
            if (wave_levels) {       // MI355REC_SLIM_WAVE_LEVELS=1: the one-wavefront-per-step kernel (comparison / diagnostics)
                if (timed) hipExtLaunchKernelGGL(slim_level_kernel, dim3(div_up(count, 4)), dim3(256), 0, s, e0, e1, 0, p, first, count);
                else hipLaunchKernelGGL(slim_level_kernel, dim3(div_up(count, 4)), dim3(256), 0, s, p, first, count);
            } else {
                if (timed) hipExtLaunchKernelGGL(slim_level_wg_kernel<512>, dim3(count), dim3(512), 0, s, e0, e1, 0, p, first, count);
                else hipLaunchKernelGGL(slim_level_wg_kernel<512>, dim3(count), dim3(512), 0, s, p, first, count);
            }
            h->stats.n_launches += 1;
        }
    }
    MI_HIP(hipGetLastError());
    h->steps_done += n;
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

}  // namespace

extern "C" int mi355rec_slim_create(mi355rec_slim_t *out, const mi355rec_slim_config *cfg, int32_t n_users, int32_t n_items,
                                    const int32_t *indptr, const int32_t *indices) {
    return guarded([&] {
        MI_REQUIRE(out && cfg && indptr && indices, "NULL argument");
        MI_REQUIRE(n_users > 0 && n_items > 0, "empty URM");
        MI_REQUIRE(cfg->sgd_mode >= MI355REC_SGD && cfg->sgd_mode <= MI355REC_ADAM, "Value for 'sgd_mode' not recognized (%d)",
                   cfg->sgd_mode);
        ensure_device();
        std::unique_ptr<mi355rec_slim> h(new mi355rec_slim());
        h->cfg = *cfg;
        h->n_users = n_users;
        h->n_items = n_items;
        h->nnz = (size_t)indptr[n_users];
        MI_REQUIRE(h->nnz > 0, "URM has no interactions");
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->call_timer.init();
        h->dispatch_timers.reserve(512);
        hipStream_t s = h->stream;
        h->indptr.upload(indptr, (size_t)n_users + 1, s);
        h->indices.upload(indices, h->nnz, s);
        h->indptr_host.assign(indptr, indptr + n_users + 1);
        h->S.alloc_zero((size_t)n_items * n_items, s);          // .pyx:129 / :1237-1254
        h->c1.alloc_zero((size_t)n_items, s);
        h->c2.alloc_zero((size_t)n_items, s);
        h->loss_slots.alloc_zero(LOSS_SLOTS, s);
        h->beta_1_power = cfg->beta_1;                           // starts at beta^1 (.pyx:163-164)
        h->beta_2_power = cfg->beta_2;
        MI_HIP(hipStreamSynchronize(s));
        *out = h.release();
    });
}

extern "C" int mi355rec_slim_run_epochs(mi355rec_slim_t h, int32_t n_epochs) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        MI_REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        ensure_device();
        const int n = h->n_users + 1;                            // totalNumberOfBatch with batch_size 1 (.pyx:215)
        ensure_capacity(h, (size_t)n);
        begin_call(h);
        double sum_profile = 0;
        for (int e = 0; e < n_epochs; ++e) {
            SlimParams p{};
            fill_params(h, p);
            p.n_steps = n;
            hipLaunchKernelGGL(slim_sample_kernel, dim3(div_up(n, 256)), dim3(256), 0, h->stream, p, h->su.ptr, h->si.ptr, h->sj.ptr);
            h->h_u.resize(n); h->h_i.resize(n); h->h_j.resize(n);
            h->su.download(h->h_u.data(), n, h->stream);
            h->si.download(h->h_i.data(), n, h->stream);
            h->sj.download(h->h_j.data(), n, h->stream);
            MI_HIP(hipStreamSynchronize(h->stream));
            run_stream(h, n, sum_profile);
            h->epochs_done += 1;
        }
        end_call(h, (long long)n * n_epochs, sum_profile);
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
        MI_HIP(hipMemcpyAsync(h->su.ptr, u, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(h->si.ptr, i, sizeof(int) * n, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(h->sj.ptr, j, sizeof(int) * n, hipMemcpyHostToDevice, s));
        h->h_u.assign(u, u + n); h->h_i.assign(i, i + n); h->h_j.assign(j, j + n);
        for (int64_t t = 0; t < n; ++t)
            MI_REQUIRE(u[t] >= 0 && u[t] < h->n_users && i[t] >= 0 && i[t] < h->n_items && j[t] >= 0 && j[t] < h->n_items,
                       "sample %lld out of range", (long long)t);
        begin_call(h);
        double sum_profile = 0;
        run_stream(h, (int)n, sum_profile);
        end_call(h, n, sum_profile);
    });
}

extern "C" int mi355rec_slim_get_S_topk(mi355rec_slim_t h, int32_t topK, int32_t *nbr_idx, float *nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && nbr_idx && nbr_val, "NULL argument");
        MI_REQUIRE(topK >= 1, "topK must be >= 1 (use mi355rec_slim_get_S_dense for the full matrix)");
        ensure_device();
        topK = std::min(topK, h->n_items);
        if (topK > MAX_TOPK) fail(MI355REC_E_UNSUPPORTED, "topK = %d exceeds the in-LDS selection limit of %d", topK, MAX_TOPK);
        const int n_pad = (h->n_items + 3) & ~3;
        const size_t lds = (size_t)n_pad * 4 + (size_t)AUX_WORDS * 4;
        if (lds + 2048 > 160 * 1024)
            fail(MI355REC_E_UNSUPPORTED, "n_items = %d: a row of S does not fit the 160 KiB LDS for the top-K selection", h->n_items);
        DeviceBuffer<int> d_idx;
        DeviceBuffer<float> d_val;
        const size_t n_out = (size_t)h->n_items * topK;
        d_idx.alloc(n_out);
        d_val.alloc(n_out);
        SlimParams p{};
        fill_params(h, p);
        auto k = slim_topk_kernel<1024>;
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int per_cu = std::max(1, std::min(2, (int)((160 * 1024) / (lds + 2048))));
        hipLaunchKernelGGL(k, dim3(std::min(h->n_items, multiprocessor_count() * per_cu)), dim3(1024), lds, h->stream, p, topK,
                           n_pad, d_idx.ptr, d_val.ptr);
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
        const size_t n2 = (size_t)h->n_items * h->n_items;
        DeviceBuffer<float> out;
        out.alloc(n2);
        SlimParams p{};
        fill_params(h, p);
        hipLaunchKernelGGL(slim_dense_kernel, dim3((unsigned)std::min<size_t>((n2 + 255) / 256, 8192)), dim3(256), 0, h->stream, p, out.ptr);
        MI_HIP(hipGetLastError());
        out.download(S, n2, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
    });
}

extern "C" int mi355rec_slim_get_stats(mi355rec_slim_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_slim_destroy(mi355rec_slim_t h) { delete h; }
