// ials.hip -- IALS solve step on MI355X (gfx950).
//
// Replaces MatrixFactorization/IALSRecommender.py (reference, pure NumPy): _run_epoch :137-166 (user pass against
// V and V^T V, then item pass against the UPDATED U and U^T U) and _update_row :170-201
//      x = inv(YtY + Y_I^T (C_I - 1) Y_I + reg I) . (Y_I^T c_I).
//
// Design.  This is the one dense contraction on the hot path; it is done in float64 like the
// reference (the systems are ill-conditioned for small `reg`, fp32 cannot meet the 1e-5 parity bar), on the FP64
// vector pipe -- on CDNA4 the FP64 matrix rate equals the FP64 vector rate, so MFMA buys nothing here.
//   gram_kernel      G = Y^T Y, register tiles + double atomics (tiny: n * k^2).
//   ials_row_kernel  one 1024-thread workgroup per row, rows pulled longest-profile-first from a queue.  The k x k
//                    system lives entirely in registers, lower-triangular tiles only: thread (ty, tx) of the 32 x 32
//                    grid owns B[ty + 32a][tx + 32b], a >= b.  (1) B = G + reg I; (2) rank-1 updates (c-1) y y^T with
//                    the profile's factor rows staged through LDS 16 at a time; (3) in-place right-looking Cholesky
//                    (SPD: no pivoting), pivot column and rhs broadcast through double-buffered LDS, one barrier per
//                    step, forward substitution fused; (4) column-oriented back substitution, one barrier per step.
#include "common.h"

#include <algorithm>
#include <memory>
#include <numeric>

namespace mi355rec {
namespace {

constexpr int CHUNK = 16;   // profile rows staged per LDS round

struct IalsParams {
    int k;
    double reg;
    const int *ptr, *idx;      // sparse rows being solved (users: C as CSR; items: C as CSC)
    const float *conf;
    const double *Y;           // fixed side factors (n_other x k)
    const double *G;           // Y^T Y (k x k)
    double *X;                 // factors being solved (n x k)
    const int *order;          // rows of this call, longest profile first
    int n_local;
    unsigned *queue;
};

// G += Y[rows]^T Y[rows]; 256 threads as a 16 x 16 grid, thread owns G[ty + 16a][tx + 16b].
template <int KT16>
__global__ __launch_bounds__(256) void gram_kernel(const double *Y, int n, int k, double *G) {
    __shared__ double ys[8][256];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double acc[KT16][KT16];
#pragma unroll
    for (int a = 0; a < KT16; ++a)
#pragma unroll
        for (int b = 0; b < KT16; ++b) acc[a][b] = 0.0;
    const int rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    for (int base = r0; base < r1; base += 8) {
        const int nr = min(8, r1 - base);
        __syncthreads();
        for (int e = tid; e < 8 * 256; e += 256) {
            const int r = e >> 8, f = e & 255;
            ys[r][f] = (r < nr && f < k) ? Y[(size_t)(base + r) * k + f] : 0.0;
        }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            double ya[KT16], yb[KT16];
#pragma unroll
            for (int a = 0; a < KT16; ++a) {
                ya[a] = ys[r][ty + 16 * a];
                yb[a] = ys[r][tx + 16 * a];
            }
#pragma unroll
            for (int a = 0; a < KT16; ++a)
#pragma unroll
                for (int b = 0; b < KT16; ++b) acc[a][b] += ya[a] * yb[b];
        }
    }
#pragma unroll
    for (int a = 0; a < KT16; ++a)
#pragma unroll
        for (int b = 0; b < KT16; ++b) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            if (r < k && c < k) atomicAdd(&G[(size_t)r * k + c], acc[a][b]);
        }
}

// 1024 threads as a 32 x 32 grid.  Thread (ty, tx) owns the cells B[ty + 32a][tx + 32b] of the LOWER-triangular
// tiles a >= b only (the system is symmetric): KT(KT+1)/2 doubles per thread, 28 at k = 200 -- no spills at the
// 128-register budget of a 16-wave workgroup.  KT = ceil(k / 32).
template <int KT>
__global__ __launch_bounds__(1024) void ials_row_kernel(const IalsParams p) {
    constexpr int KPAD = KT * 32;
    __shared__ double ys[CHUNK][KPAD];        // staged factor rows of the profile
    __shared__ double wts[CHUNK];             // c - 1
    __shared__ double cfs[CHUNK];             // c
    __shared__ double pub[2][KPAD];           // pivot column (factorisation) / pivot row (back substitution), double-buffered
    __shared__ double rhsb[2];
    __shared__ int s_row;

    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int k = p.k;

    for (;;) {
        if (tid == 0) s_row = (int)atomicAdd(p.queue, 1u);
        __syncthreads();
        const int slot = s_row;
        if (slot >= p.n_local) break;
        const int row = p.order[slot];
        const int beg = p.ptr[row], end = p.ptr[row + 1];

        // (1) B = YtY + reg I                                    (IALSRecommender.py:199), lower tiles only
        double B[KT][KT];
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const int r = ty + 32 * a, c = tx + 32 * b;
                double v = 0.0;
                if (r < k && c < k) v = p.G[(size_t)r * k + c] + (r == c ? p.reg : 0.0);
                else if (r == c) v = 1.0;                        // identity padding keeps the factorisation well defined
                B[a][b] = v;
            }
        double rhs = 0.0;                                         // thread t < k carries (Y_I^T c)[t]  (:201)

        // (2) A = Y_I^T ((c - 1) o Y_I), rhs = Y_I^T c           (:197, :201)
        for (int base = beg; base < end; base += CHUNK) {
            const int nr = min(CHUNK, end - base);
            __syncthreads();
            for (int e = tid; e < CHUNK * KPAD; e += 1024) {
                const int r = e / KPAD, f = e % KPAD;
                double v = 0.0;
                if (r < nr && f < k) v = p.Y[(size_t)p.idx[base + r] * k + f];
                ys[r][f] = v;
            }
            if (tid < CHUNK) {
                const double c = tid < nr ? (double)p.conf[base + tid] : 1.0;
                cfs[tid] = tid < nr ? c : 0.0;
                wts[tid] = c - 1.0;
            }
            __syncthreads();
            for (int r = 0; r < nr; ++r) {
                const double w = wts[r];
                double yb[KT];
#pragma unroll
                for (int b = 0; b < KT; ++b) yb[b] = ys[r][tx + 32 * b];
#pragma unroll
                for (int a = 0; a < KT; ++a) {
                    const double ya = ys[r][ty + 32 * a] * w;
#pragma unroll
                    for (int b = 0; b <= a; ++b) B[a][b] += ya * yb[b];
                }
            }
            if (tid < k) {
                double s = 0.0;
                for (int r = 0; r < nr; ++r) s += ys[r][tid] * cfs[r];
                rhs += s;
            }
        }

        // (3) B = L L^T in place (right-looking Cholesky, one published column and one barrier per step), with the
        // forward substitution L z = rhs fused in.  The tile loop is unrolled at compile time so that every register
        // index is static.  (The reference forms inv(B) . rhs, :201; same solution.)
        int step = 0;
#pragma unroll
        for (int JB = 0; JB < KT; ++JB) {
#pragma unroll 1
            for (int jl = 0; jl < 32; ++jl) {
                const int j = JB * 32 + jl;
                if (j >= k) break;
                const int buf = (step++) & 1;
                if (tx == jl) {      // owners of column j publish B[:, j] (rows of tiles a >= JB)
#pragma unroll
                    for (int a = JB; a < KT; ++a) pub[buf][ty + 32 * a] = B[a][JB];
                }
                if (tid == j) rhsb[buf] = rhs;
                __syncthreads();
                const double d = pub[buf][j];
                const double inv = 1.0 / sqrt(d);               // 1 / L_jj
                const double zj = rhsb[buf] * inv;
                double lc[KT];
#pragma unroll
                for (int b = JB; b < KT; ++b) {
                    const int c = tx + 32 * b;
                    lc[b] = c > j ? pub[buf][c] * inv : 0.0;     // L[c][j]; columns <= j are not updated
                }
#pragma unroll
                for (int a = JB; a < KT; ++a) {
                    const int r = ty + 32 * a;
                    const double lr = r > j ? pub[buf][r] * inv : 0.0;   // L[r][j]
#pragma unroll
                    for (int b = JB; b <= a; ++b) {
                        double v = B[a][b] - lr * lc[b];
                        if (b == JB && tx == jl) v = r > j ? lr : (r == j ? d * inv : B[a][b]);   // column j becomes L[:, j]
                        B[a][b] = v;
                    }
                }
                if (tid < k) rhs = tid == j ? zj : (tid > j ? rhs - pub[buf][tid] * inv * zj : rhs);
            }
        }
        // (4) back substitution L^T x = z, column oriented: x_j = z_j / L_jj, then z_c -= L[j][c] x_j for c < j.
        // Row j of L is held by the threads with ty == j % 32 in the tiles (JA, b <= JA).
#pragma unroll
        for (int JA = KT - 1; JA >= 0; --JA) {
#pragma unroll 1
            for (int jl = 31; jl >= 0; --jl) {
                const int j = JA * 32 + jl;
                if (j >= k) continue;
                const int buf = (step++) & 1;
                if (ty == jl) {
#pragma unroll
                    for (int b = 0; b <= JA; ++b) pub[buf][tx + 32 * b] = B[JA][b];
                }
                if (tid == j) rhsb[buf] = rhs;
                __syncthreads();
                const double xj = rhsb[buf] / pub[buf][j];
                if (tid < k) rhs = tid == j ? xj : (tid < j ? rhs - pub[buf][tid] * xj : rhs);
            }
        }
        if (tid < k) p.X[(size_t)row * k + tid] = rhs;
        __syncthreads();
    }
}

// CSR -> CSC of the confidence matrix on the device (row ids inside a column end up unordered: irrelevant here,
// the Gramian is a sum).
__global__ void ials_count_kernel(const int *idx, size_t nnz, int *cnt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x)
        atomicAdd(&cnt[idx[i]], 1);
}
__global__ __launch_bounds__(1024) void ials_scan_kernel(const int *cnt, int *out, int *cursor, int n) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int b = tid * chunk, e = min(n, b + chunk);
    long long s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long long t = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    long long run = tid ? part[tid - 1] : 0;
    for (int i = b; i < e; ++i) {
        out[i] = (int)run;
        cursor[i] = (int)run;
        run += cnt[i];
    }
    if (tid == 1023) out[n] = (int)part[1023];
}
__global__ void ials_scatter_kernel(const int *ptr, const int *idx, const float *val, int n_rows, int *cursor, int *t_idx,
                                    float *t_val) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_rows) return;
    for (int q = ptr[wave] + lane; q < ptr[wave + 1]; q += 64) {
        const int pos = atomicAdd(&cursor[idx[q]], 1);
        t_idx[pos] = wave;
        t_val[pos] = val[q];
    }
}

}  // namespace
}  // namespace mi355rec

using namespace mi355rec;

struct mi355rec_ials {
    int n_users = 0, n_items = 0, k = 0;
    double reg = 0;
    size_t nnz = 0;
    hipStream_t stream = nullptr;
    StreamTimer call_timer;
    DispatchTimers dispatch_timers;
    DeviceBuffer<int> u_ptr, u_idx, i_ptr, i_idx, order;
    DeviceBuffer<float> u_conf, i_conf;
    DeviceBuffer<double> U, V, G;
    DeviceBuffer<unsigned> queue;
    std::vector<int> user_order, item_order;     // longest profile first
    std::vector<int> u_ptr_host, i_ptr_host;
    std::vector<int> staging;
    mi355rec_stats stats{};
    double flops_acc = 0, bytes_acc = 0;
    long long rows_acc = 0, launches_acc = 0;

    ~mi355rec_ials() {
        if (stream) (void)hipStreamSynchronize(stream);
        call_timer.destroy();
        dispatch_timers.destroy();
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

template <int KT16>
void launch_gram_t(mi355rec_ials *h, const double *Y, int n, int grid) {
    hipLaunchKernelGGL(gram_kernel<KT16>, dim3(grid), dim3(256), 0, h->stream, Y, n, h->k, h->G.ptr);
}

void launch_gram(mi355rec_ials *h, const double *Y, int n) {
    MI_HIP(hipMemsetAsync(h->G.ptr, 0, sizeof(double) * h->k * h->k, h->stream));
    const int grid = std::max(1, std::min(2 * multiprocessor_count(), (n + 63) / 64));
    switch ((h->k + 15) / 16) {
        case 1: launch_gram_t<1>(h, Y, n, grid); break;
        case 2: launch_gram_t<2>(h, Y, n, grid); break;
        case 3: launch_gram_t<3>(h, Y, n, grid); break;
        case 4: launch_gram_t<4>(h, Y, n, grid); break;
        case 5: launch_gram_t<5>(h, Y, n, grid); break;
        case 6: launch_gram_t<6>(h, Y, n, grid); break;
        case 7: launch_gram_t<7>(h, Y, n, grid); break;
        case 8: launch_gram_t<8>(h, Y, n, grid); break;
        case 9: launch_gram_t<9>(h, Y, n, grid); break;
        case 10: launch_gram_t<10>(h, Y, n, grid); break;
        case 11: launch_gram_t<11>(h, Y, n, grid); break;
        case 12: launch_gram_t<12>(h, Y, n, grid); break;
        case 13: launch_gram_t<13>(h, Y, n, grid); break;
        default: launch_gram_t<14>(h, Y, n, grid); break;
    }
}

template <int KT>
void launch_rows_t(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1) {
    hipExtLaunchKernelGGL(ials_row_kernel<KT>, dim3(grid), dim3(1024), 0, h->stream, e0, e1, 0, p);
}

void launch_rows(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1) {
    switch ((h->k + 31) / 32) {
        case 1: launch_rows_t<1>(h, p, grid, e0, e1); break;
        case 2: launch_rows_t<2>(h, p, grid, e0, e1); break;
        case 3: launch_rows_t<3>(h, p, grid, e0, e1); break;
        case 4: launch_rows_t<4>(h, p, grid, e0, e1); break;
        case 5: launch_rows_t<5>(h, p, grid, e0, e1); break;
        case 6: launch_rows_t<6>(h, p, grid, e0, e1); break;
        default: launch_rows_t<7>(h, p, grid, e0, e1); break;
    }
}

// Solve the rows [r0, r1) of one side.
void half_step(mi355rec_ials *h, bool users, int r0, int r1) {
    const int n_side = users ? h->n_users : h->n_items;
    MI_REQUIRE(r0 >= 0 && r1 <= n_side && r0 <= r1, "row range [%d,%d) outside [0,%d)", r0, r1, n_side);
    const std::vector<int> &full = users ? h->user_order : h->item_order;
    const std::vector<int> &ptr = users ? h->u_ptr_host : h->i_ptr_host;
    h->staging.clear();
    double nnz_rows = 0;
    for (int r : full)
        if (r >= r0 && r < r1 && ptr[r + 1] > ptr[r]) {       // warm rows only (IALSRecommender.py:78-82)
            h->staging.push_back(r);
            nnz_rows += ptr[r + 1] - ptr[r];
        }
    const int n_local = (int)h->staging.size();
    launch_gram(h, users ? h->V.ptr : h->U.ptr, users ? h->n_items : h->n_users);
    if (n_local == 0) return;
    MI_HIP(hipMemcpyAsync(h->order.ptr, h->staging.data(), sizeof(int) * n_local, hipMemcpyHostToDevice, h->stream));
    MI_HIP(hipMemsetAsync(h->queue.ptr, 0, sizeof(unsigned), h->stream));
    IalsParams p{};
    p.k = h->k;
    p.reg = h->reg;
    p.ptr = users ? h->u_ptr.ptr : h->i_ptr.ptr;
    p.idx = users ? h->u_idx.ptr : h->i_idx.ptr;
    p.conf = users ? h->u_conf.ptr : h->i_conf.ptr;
    p.Y = users ? h->V.ptr : h->U.ptr;
    p.G = h->G.ptr;
    p.X = users ? h->U.ptr : h->V.ptr;
    p.order = h->order.ptr;
    p.n_local = n_local;
    p.queue = h->queue.ptr;
    const int grid = std::min(n_local, multiprocessor_count());
    hipEvent_t e0 = nullptr, e1 = nullptr;
    h->dispatch_timers.next(e0, e1, 1 << 30);
    launch_rows(h, p, grid, e0, e1);
    MI_HIP(hipGetLastError());
    // ALGORITHMIC work, SURVEY.md section 8(d): Gramian 2 * nnz * k^2 flop per pass (+ the k x k base Gramian 2 n k^2),
    // solve k^3/3 + 2 k^2 per row; bytes: gathered factor rows + confidences + the solved rows.
    const double k = h->k;
    h->flops_acc += 2.0 * nnz_rows * k * k + (double)n_local * (k * k * k / 3.0 + 2.0 * k * k) +
                    2.0 * (double)(users ? h->n_items : h->n_users) * k * k;
    h->bytes_acc += nnz_rows * (8.0 * k + 8.0) + (double)n_local * 8.0 * k;
    h->rows_acc += n_local;
    h->launches_acc += 1;
}

void begin_call(mi355rec_ials *h) {
    h->dispatch_timers.reset();
    h->flops_acc = h->bytes_acc = 0;
    h->rows_acc = h->launches_acc = 0;
    h->call_timer.start(h->stream);
}

void end_call(mi355rec_ials *h, bool sync) {
    h->call_timer.stop(h->stream);
    if (!sync) return;
    MI_HIP(hipStreamSynchronize(h->stream));
    h->stats.call_ms = h->call_timer.elapsed_ms();
    h->stats.kernel_ms = h->dispatch_timers.total_ms();
    h->stats.n_timed = h->dispatch_timers.used;
    h->stats.n_launches = h->launches_acc;
    h->stats.n_units = h->rows_acc;
    h->stats.algorithmic_bytes = h->bytes_acc;
    h->stats.algorithmic_flops = h->flops_acc;
    h->stats.loss = 0;
}

}  // namespace

extern "C" int mi355rec_ials_create(mi355rec_ials_t *out, int32_t n_users, int32_t n_items, int32_t n_factors, double reg,
                                    const int32_t *indptr, const int32_t *indices, const float *confidence, const double *U0,
                                    const double *V0) {
    return guarded([&] {
        MI_REQUIRE(out && indptr && indices && confidence && V0, "NULL argument");
        MI_REQUIRE(n_users > 0 && n_items > 0, "empty URM");
        MI_REQUIRE(n_factors >= 1, "num_factors must be >= 1");
        if (n_factors > 224)   // 7 x 7 tiles of 32: 28 lower-triangular doubles per thread; the search space stops at 200
            fail(MI355REC_E_UNSUPPORTED, "num_factors = %d: the register-resident solver covers num_factors <= 224", n_factors);
        ensure_device();
        std::unique_ptr<mi355rec_ials> h(new mi355rec_ials());
        h->n_users = n_users;
        h->n_items = n_items;
        h->k = n_factors;
        h->reg = reg;
        h->nnz = (size_t)indptr[n_users];
        MI_REQUIRE(h->nnz > 0, "URM has no interactions");
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->call_timer.init();
        h->dispatch_timers.reserve(8);
        hipStream_t s = h->stream;
        const size_t nnz = h->nnz;
        h->u_ptr.upload(indptr, (size_t)n_users + 1, s);
        h->u_idx.upload(indices, nnz, s);
        h->u_conf.upload(confidence, nnz, s);
        // item-major view (the reference keeps C_csc, IALSRecommender.py:106), built on the device
        DeviceBuffer<int> cnt, cursor;
        cnt.alloc_zero((size_t)n_items, s);
        cursor.alloc((size_t)n_items);
        h->i_ptr.alloc((size_t)n_items + 1);
        h->i_idx.alloc(nnz);
        h->i_conf.alloc(nnz);
        const int eb = 256, eg = (int)std::min<size_t>((nnz + eb - 1) / eb, 4096);
        hipLaunchKernelGGL(ials_count_kernel, dim3(eg), dim3(eb), 0, s, h->u_idx.ptr, nnz, cnt.ptr);
        hipLaunchKernelGGL(ials_scan_kernel, dim3(1), dim3(1024), 0, s, cnt.ptr, h->i_ptr.ptr, cursor.ptr, n_items);
        hipLaunchKernelGGL(ials_scatter_kernel, dim3(div_up((int64_t)n_users * 64, 256)), dim3(256), 0, s, h->u_ptr.ptr,
                           h->u_idx.ptr, h->u_conf.ptr, n_users, cursor.ptr, h->i_idx.ptr, h->i_conf.ptr);
        MI_HIP(hipGetLastError());
        const size_t nu = (size_t)n_users * h->k, ni = (size_t)n_items * h->k;
        if (U0) h->U.upload(U0, nu, s); else h->U.alloc_zero(nu, s);
        h->V.upload(V0, ni, s);
        h->G.alloc((size_t)h->k * h->k);
        h->order.alloc((size_t)std::max(n_users, n_items));
        h->queue.alloc(1);
        h->u_ptr_host.assign(indptr, indptr + n_users + 1);
        h->i_ptr_host.resize((size_t)n_items + 1);
        h->i_ptr.download(h->i_ptr_host.data(), (size_t)n_items + 1, s);
        MI_HIP(hipStreamSynchronize(s));
        auto by_length = [](const std::vector<int> &ptr, int n) {
            std::vector<int> o(n);
            std::iota(o.begin(), o.end(), 0);
            std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return ptr[a + 1] - ptr[a] > ptr[b + 1] - ptr[b]; });
            return o;
        };
        h->user_order = by_length(h->u_ptr_host, n_users);
        h->item_order = by_length(h->i_ptr_host, n_items);
        *out = h.release();
    });
}

extern "C" int mi355rec_ials_run_epochs(mi355rec_ials_t h, int32_t n_epochs) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        MI_REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        ensure_device();
        h->dispatch_timers.reserve(2 * std::max(1, n_epochs));
        begin_call(h);
        for (int e = 0; e < n_epochs; ++e) {
            half_step(h, true, 0, h->n_users);      // fit user factors against V     (IALSRecommender.py:141-152)
            half_step(h, false, 0, h->n_items);     // then item factors against the UPDATED U (:156-166)
        }
        end_call(h, true);
    });
}

extern "C" int mi355rec_ials_user_half(mi355rec_ials_t h, int32_t u0, int32_t u1) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        h->dispatch_timers.reserve(2);
        begin_call(h);
        half_step(h, true, u0, u1);
        end_call(h, false);
    });
}

extern "C" int mi355rec_ials_item_half(mi355rec_ials_t h, int32_t i0, int32_t i1) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        h->dispatch_timers.reserve(2);
        begin_call(h);
        half_step(h, false, i0, i1);
        end_call(h, false);
    });
}

extern "C" int mi355rec_ials_device_factors(mi355rec_ials_t h, double **d_U, double **d_V) {
    return guarded([&] {
        MI_REQUIRE(h && d_U && d_V, "NULL argument");
        *d_U = h->U.ptr;
        *d_V = h->V.ptr;
    });
}

extern "C" int mi355rec_ials_sync(mi355rec_ials_t h) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        MI_HIP(hipStreamSynchronize(h->stream));
        h->stats.call_ms = h->call_timer.elapsed_ms();
        h->stats.kernel_ms = h->dispatch_timers.total_ms();
        h->stats.n_timed = h->dispatch_timers.used;
        h->stats.n_launches = h->launches_acc;
        h->stats.n_units = h->rows_acc;
        h->stats.algorithmic_bytes = h->bytes_acc;
        h->stats.algorithmic_flops = h->flops_acc;
    });
}

extern "C" int mi355rec_ials_get_factors(mi355rec_ials_t h, double *U, double *V) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        if (U) h->U.download(U, (size_t)h->n_users * h->k, h->stream);
        if (V) h->V.download(V, (size_t)h->n_items * h->k, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
    });
}

extern "C" int mi355rec_ials_get_stats(mi355rec_ials_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_ials_destroy(mi355rec_ials_t h) { delete h; }
