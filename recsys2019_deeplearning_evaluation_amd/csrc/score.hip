// score.hip -- matrix-factorisation scoring and top-cutoff ranking on MI355X (gfx950)  [SURVEY.md section 8(f) rank 1].
//
// Replaces, for factor models, BaseMatrixFactorizationRecommender._compute_item_score
// (Base/BaseMatrixFactorizationRecommender.py:38-70: scores = USER_factors[users] . ITEM_factors^T (+ biases)) and the
// filtering + ranking half of BaseRecommender.recommend (Base/BaseRecommender.py:131-222: seen / excluded items -> -inf,
// the `cutoff` best items per user in descending score order, -inf items dropped) -- the step the reference's
// EvaluatorHoldout runs on every validation (Base/Evaluation/Evaluator.py:436).
//
//   score_gemm_kernel  the one GEMM-shaped op next to the hot path: a 128-user x 128-item tile per workgroup, 2 x 2 wavefronts
//                      with four 32 x 32 accumulators each on v_mfma_f32_32x32x2_f32 (exact f32: bitwise an fmaf chain, at the
//                      f32 vector rate), K in LDS-staged chunks of 32 with the next chunk's 16-byte loads in flight; biases
//                      added in the epilogue.
//   score_rank_kernel  one workgroup per user: the score row is pulled into LDS, seen / excluded items are set to -inf,
//                      the same in-LDS radix-select + bitonic-sort top-K as the similarity build emits the ranking.
//   wide ranking       catalogues whose score row does not fit LDS (> ~32 k items) and cutoffs above the in-LDS selection limit
//                      (the reference's default cutoff=None ranks ALL items): rows stay in HBM, a filter kernel writes -inf,
//                      one segmented radix sort (rocPRIM, stable: ties keep the lower item id, like the in-LDS path) orders every
//                      row, the first `cutoff` finite entries are the ranking.
#include "common.h"
#include "topk.cuh"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <memory>

namespace mi355rec {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, TN = 128;   // users x items of a workgroup's tile: 2 x 2 wavefronts, 64 x 64 (four 32 x 32 accumulators) each
constexpr int TK = 32;              // K chunk staged in LDS
constexpr int LDT = 129;            // leading dimension of the [k][row] tiles: consecutive rows -> consecutive banks for the MFMA operand
                                    // reads, and 4 * LDT = 4 (mod 32) spreads the 8 k-quads x 4 rows a half-wave stores over all 32 banks

struct ScoreParams {
    int n_users, n_items, k, use_bias;
    const float *U, *V, *bu, *bi;
    float mu;
    const int *users;               // user id of every row of the batch
    int n_batch;
    float *scores;                  // [n_batch][n_items]
};

// four consecutive factors of one row: one 16-byte load when the row stride keeps it aligned and the quad lies inside the row
__device__ __forceinline__ float4 load_quad(const float *row, int kk, int k, bool aligned) {
    if (row == nullptr || kk >= k) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (aligned && kk + 3 < k) return *reinterpret_cast<const float4 *>(row + kk);
    float4 r;
    r.x = row[kk];
    r.y = kk + 1 < k ? row[kk + 1] : 0.f;
    r.z = kk + 2 < k ? row[kk + 2] : 0.f;
    r.w = kk + 3 < k ? row[kk + 3] : 0.f;
    return r;
}

// scores[b][item] = U[users[b]] . V[item] (+ biases).  Round 1's kernel (32 x 128 tile, one accumulator per wavefront, the K
// chunk staged with 40 scalar loads + 40 4-byte LDS stores per thread and nothing in flight during the MFMAs) spent its time
// staging: 0.14 of the f32 matrix peak.  Here a 128 x 128 tile, four 32 x 32 accumulators per wavefront (every operand read from
// LDS feeds two MFMAs), 16-byte global loads of the NEXT K chunk in flight while the current one is multiplied.
__global__ __launch_bounds__(256) void score_gemm_kernel(const ScoreParams p) {
    __shared__ float As[TK][LDT];          // [k][user row of the tile]
    __shared__ float Bs[TK][LDT];          // [k][item of the tile]
    __shared__ int s_user[TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TN;
    if (tid < TM) s_user[tid] = row0 + tid < p.n_batch ? p.users[row0 + tid] : -1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    __syncthreads();
    // staging: thread t moves quads e = t + 256 i (i < 4) of both tiles: row m = e / 8, k-quad kq = e % 8
    const bool aligned = (p.k & 3) == 0;
    const float *arow[4], *brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = (tid + 256 * i) >> 3;
        const int u = s_user[m];
        arow[i] = u >= 0 ? p.U + (size_t)u * p.k : nullptr;
        brow[i] = col0 + m < p.n_items ? p.V + (size_t)(col0 + m) * p.k : nullptr;
    }
    const int kq4 = (tid & 7) * 4;
    float4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = load_quad(arow[i], kq4, p.k, aligned);
        rb[i] = load_quad(brow[i], kq4, p.k, aligned);
    }
    for (int k0 = 0; k0 < p.k; k0 += TK) {
        __syncthreads();                   // everybody has finished multiplying the previous chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = (tid + 256 * i) >> 3;
            As[kq4][m] = ra[i].x; As[kq4 + 1][m] = ra[i].y; As[kq4 + 2][m] = ra[i].z; As[kq4 + 3][m] = ra[i].w;
            Bs[kq4][m] = rb[i].x; Bs[kq4 + 1][m] = rb[i].y; Bs[kq4 + 2][m] = rb[i].z; Bs[kq4 + 3][m] = rb[i].w;
        }
        __syncthreads();
        if (k0 + TK < p.k) {               // the next chunk's loads fly during this chunk's MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = load_quad(arow[i], k0 + TK + kq4, p.k, aligned);
                rb[i] = load_quad(brow[i], k0 + TK + kq4, p.k, aligned);
            }
        }
        // A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]   (32x32x2 f32 operand maps); factors beyond k are zero
        const float *a_col = &As[lane >> 5][wm * 64 + (lane & 31)];
        const float *b_col = &Bs[lane >> 5][wn * 64 + (lane & 31)];
#pragma unroll 4
        for (int s = 0; s < TK; s += 2) {
            const float a0 = a_col[s * LDT], a1 = a_col[s * LDT + 32];
            const float b0 = b_col[s * LDT], b1 = b_col[s * LDT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int item = col0 + wn * 64 + b * 32 + (lane & 31);
        if (item >= p.n_items) continue;
        const float item_term = p.use_bias ? p.bi[item] + p.mu : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = wm * 64 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int u = s_user[r];
                if (u < 0) continue;
                float v = acc[a][b][reg] + item_term;
                if (p.use_bias) v += p.bu[u];
                p.scores[(size_t)(row0 + r) * p.n_items + item] = v;
            }
        }
    }
}

// THRESHOLD-FIRST ranking of a score row in LDS (round 6; the similarity build's selection, csrc/sim.hip): the cutoff largest scores are
// all >= the cutoff-th largest of the THREADS thread maxima (the cutoff largest maxima are cutoff different cells), so one 16-bit radix
// select over THREADS keys gives a bound, one pass over the row collects the cells at or above it -- a few times `cutoff` of them, ties
// included -- and those are ranked exactly (value descending, ties towards the lower item, as block_topk_emit does).  Returns false --
// nothing emitted, *ncand back at 0 -- where the full radix select over the row has to run instead: fewer finite scores than the cutoff,
// a cutoff that is not small next to the number of maxima, a candidate list that overflows (a sparse model's row of zeros).
// aux[0..255] must be zero on entry, `tmax` is the calling thread's largest score, *ncand 0.  Ends behind a barrier either way.
template <int THREADS>
__device__ __forceinline__ bool rank_threshold_first(const float *acc, int n, int cutoff, uint32_t nfinite, float tmax, uint32_t *aux, SelectScratch &sc,
                                                     uint32_t *ncand_shared, int *out) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (!(nfinite >= (uint32_t)cutoff && 4 * cutoff <= THREADS)) return false;
    const uint32_t prefix = block_kth_largest_prefix16<THREADS>(float_key(tmax), (uint32_t)cutoff, aux, sc);
    const uint32_t T = prefix << 16;
    if (tid == 0) sc.out_count = 0;
    uint64_t *cand = reinterpret_cast<uint64_t *>(aux);
    constexpr int CAND_MAX = AUX_WORDS / 2;
    for (int j0 = 0; j0 < n; j0 += THREADS) {
        const int j = j0 + tid;
        const float v = j < n ? acc[j] : -INFINITY;
        const uint32_t key = float_key(v);
        const bool take = v > -INFINITY && key >= T;
        const unsigned long long m = __ballot(take);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(ncand_shared, (uint32_t)__popcll(m));
            base = __shfl(base, leader);
            if (take) {
                const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (slot < (uint32_t)CAND_MAX) cand[slot] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)j);
            }
        }
    }
    __syncthreads();
    const uint32_t ncand = *ncand_shared;
    if (ncand <= (uint32_t)CAND_MAX) {
        block_rank_emit<THREADS>(cand, (int)ncand, cutoff, (uint32_t)cutoff, 0u, sc, out, nullptr);
        return true;
    }
    __syncthreads();
    if (tid == 0) *ncand_shared = 0;
    __syncthreads();
    return false;
}

struct RankParams {
    int n_items, n_pad, cutoff, remove_seen;
    const int *users, *seen_ptr, *seen_idx;
    const unsigned char *allowed;   // nullable: 0 marks an excluded item (items_to_compute / top-pop / custom filters)
    float *scores;
    int *ranked;
    int write_back;                 // store the filtered row (return_scores=True)
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void score_rank_kernel(const RankParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *acc = smem;
    uint32_t *aux = reinterpret_cast<uint32_t *>(smem + p.n_pad);
    __shared__ SelectScratch sc;
    __shared__ uint32_t s_ncand, s_nfinite;
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.x;
    float *row = p.scores + (size_t)b * p.n_items;
    if (tid == 0) { s_ncand = 0; s_nfinite = 0; }
    for (int j = tid; j < p.n_items; j += THREADS) {
        float v = row[j];
        if (p.allowed && !p.allowed[j]) v = -INFINITY;
        acc[j] = v;
    }
    __syncthreads();
    if (p.remove_seen) {            // _remove_seen_on_scores (BaseRecommender.py:115-123)
        const int u = p.users[b];
        for (int q = p.seen_ptr[u] + tid; q < p.seen_ptr[u + 1]; q += THREADS) acc[p.seen_idx[q]] = -INFINITY;
        __syncthreads();
    }
    uint32_t nfin = 0;
    float tmax = -INFINITY;         // this thread's largest score (threshold-first selection below)
    for (int j = tid; j < p.n_items; j += THREADS) {
        const float v = acc[j];
        nfin += v > -INFINITY;
        tmax = fmaxf(tmax, v);
        if (p.write_back) row[j] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nfin += __shfl_down(nfin, off);
    if (lane == 0 && nfin) atomicAdd(&s_nfinite, nfin);
    for (int w = tid; w < 256; w += THREADS) aux[w] = 0;           // (the bins of block_kth_largest_prefix16)
    __syncthreads();
    int *out = p.ranked + (size_t)b * p.cutoff;
    if (rank_threshold_first<THREADS>(acc, p.n_items, p.cutoff, s_nfinite, tmax, aux, sc, &s_ncand, out)) return;
    block_topk_emit<THREADS>(acc, p.n_items, p.cutoff, s_nfinite, 0u, TOPK_FINITE, aux, sc, &s_ncand, out, nullptr);
}

// ---- wide ranking: rows in HBM -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_filter_kernel(float *scores, int *ids, int n_items, const int *users, const int *seen_ptr,
                                                          const int *seen_idx, const unsigned char *allowed, int remove_seen) {
    const int b = blockIdx.y;
    float *row = scores + (size_t)b * n_items;
    int *id_row = ids + (size_t)b * n_items;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_items; j += gridDim.x * 256) {
        if (allowed && !allowed[j]) row[j] = -INFINITY;
        id_row[j] = j;
    }
    if (remove_seen && blockIdx.x == 0) {      // a -inf store racing with the `allowed` -inf store of another block is the same value
        const int u = users[b];
        for (int q = seen_ptr[u] + threadIdx.x; q < seen_ptr[u + 1]; q += 256) row[seen_idx[q]] = -INFINITY;
    }
}

__global__ __launch_bounds__(256) void wide_emit_kernel(const float *sorted_scores, const int *sorted_ids, int n_items, int cutoff,
                                                        int *ranked) {
    const int b = blockIdx.y;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < cutoff; r += gridDim.x * 256) {
        const size_t at = (size_t)b * n_items + r;
        ranked[(size_t)b * cutoff + r] = sorted_scores[at] > -INFINITY ? sorted_ids[at] : -1;
    }
}

struct WideRanker {
    DeviceBuffer<int> ids_in, ids_out, offsets;
    DeviceBuffer<float> keys_out;
    DeviceBuffer<unsigned char> tmp;
    size_t capacity = 0;
    int offsets_n = 0, offsets_items = 0;

    // scores: [n][n_items] in HBM (unfiltered); ranked: [n][cutoff] on the device
    void rank(hipStream_t s, float *scores, int n, int n_items, int cutoff, const int *users, const int *seen_ptr, const int *seen_idx,
              const unsigned char *allowed, int remove_seen, int *ranked) {
        const size_t total = (size_t)n * n_items;
        if (total >= (1ull << 31)) fail(MI355REC_E_UNSUPPORTED, "user block of %d x %d scores exceeds the segmented sort (2^31 cells): use smaller blocks", n, n_items);
        if (capacity < total) {
            ids_in.alloc(total); ids_out.alloc(total); keys_out.alloc(total);
            capacity = total;
        }
        if (offsets_n < n || offsets_items != n_items) {
            std::vector<int> host((size_t)n + 1);
            for (int b = 0; b <= n; ++b) host[b] = (int)((size_t)b * n_items);
            offsets.alloc((size_t)n + 1);
            MI_HIP(hipMemcpyAsync(offsets.ptr, host.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, s));
            MI_HIP(hipStreamSynchronize(s));
            offsets_n = n; offsets_items = n_items;
        }
        hipLaunchKernelGGL(wide_filter_kernel, dim3(std::min(div_up(n_items, 256), 64), n), dim3(256), 0, s, scores, ids_in.ptr, n_items,
                           users, seen_ptr, seen_idx, allowed, remove_seen);
        size_t bytes = 0;
        MI_HIP(rocprim::segmented_radix_sort_pairs_desc(nullptr, bytes, scores, keys_out.ptr, ids_in.ptr, ids_out.ptr, (int)total,
                                                                     n, offsets.ptr, offsets.ptr + 1, 0, 32, s));
        if (tmp.count < bytes) tmp.alloc(bytes + 256);
        bytes = tmp.count;
        MI_HIP(rocprim::segmented_radix_sort_pairs_desc(tmp.ptr, bytes, scores, keys_out.ptr, ids_in.ptr, ids_out.ptr, (int)total,
                                                                     n, offsets.ptr, offsets.ptr + 1, 0, 32, s));
        hipLaunchKernelGGL(wide_emit_kernel, dim3(std::min(div_up(cutoff, 256), 64), n), dim3(256), 0, s, keys_out.ptr, ids_out.ptr, n_items,
                           cutoff, ranked);
    }
};

bool fits_lds_rank(int n_items, int cutoff) {
    const size_t lds = ((size_t)((n_items + 3) & ~3)) * 4 + (size_t)AUX_WORDS * 4 + 2048;
    return lds <= 160 * 1024 && cutoff <= MAX_TOPK;
}

}  // namespace
}  // namespace mi355rec

using namespace mi355rec;

struct mi355rec_scorer {
    WideRanker wide;
    int n_users = 0, n_items = 0, k = 0, use_bias = 0;
    float mu = 0.f;
    hipStream_t stream = nullptr;
    StreamTimer gemm_timer, call_timer;
    DeviceBuffer<float> U, V, bu, bi, scores;
    DeviceBuffer<int> seen_ptr, seen_idx, users, ranked;
    DeviceBuffer<unsigned char> allowed;
    mi355rec_stats stats{};

    ~mi355rec_scorer() {
        if (stream) (void)hipStreamSynchronize(stream);
        gemm_timer.destroy();
        call_timer.destroy();
        ReleaseScope::forget(stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {
void upload_model(mi355rec_scorer *h, const float *U, const float *V, const float *bu, const float *bi, float mu) {
    hipStream_t s = h->stream;
    MI_HIP(hipMemcpyAsync(h->U.ptr, U, sizeof(float) * (size_t)h->n_users * h->k, hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(h->V.ptr, V, sizeof(float) * (size_t)h->n_items * h->k, hipMemcpyHostToDevice, s));
    if (h->use_bias) {
        MI_REQUIRE(bu && bi, "use_bias is set but the bias vectors are NULL");
        MI_HIP(hipMemcpyAsync(h->bu.ptr, bu, sizeof(float) * h->n_users, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(h->bi.ptr, bi, sizeof(float) * h->n_items, hipMemcpyHostToDevice, s));
        h->mu = mu;
    }
    MI_HIP(hipStreamSynchronize(s));
}
}  // namespace

extern "C" int mi355rec_scorer_create(mi355rec_scorer_t *out, int32_t n_users, int32_t n_items, int32_t n_factors,
                                      const float *U, const float *V, int32_t use_bias, const float *user_bias,
                                      const float *item_bias, float global_bias, const int32_t *seen_indptr,
                                      const int32_t *seen_indices) {
    return guarded([&] {
        MI_REQUIRE(out && U && V && seen_indptr && seen_indices, "NULL argument");
        MI_REQUIRE(n_users > 0 && n_items > 0 && n_factors > 0, "empty model");
        ensure_device();
        std::unique_ptr<mi355rec_scorer> h(new mi355rec_scorer());
        h->n_users = n_users; h->n_items = n_items; h->k = n_factors; h->use_bias = use_bias != 0;
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->gemm_timer.init();
        h->call_timer.init();
        h->U.alloc((size_t)n_users * n_factors);
        h->V.alloc((size_t)n_items * n_factors);
        if (h->use_bias) { h->bu.alloc(n_users); h->bi.alloc(n_items); }
        h->seen_ptr.upload(seen_indptr, (size_t)n_users + 1, h->stream);
        h->seen_idx.upload(seen_indices, (size_t)seen_indptr[n_users], h->stream);
        h->allowed.alloc(n_items);
        upload_model(h.get(), U, V, user_bias, item_bias, global_bias);
        *out = h.release();
    });
}

extern "C" int mi355rec_scorer_update(mi355rec_scorer_t h, const float *U, const float *V, const float *user_bias,
                                      const float *item_bias, float global_bias) {
    return guarded([&] {
        MI_REQUIRE(h && U && V, "NULL argument");
        ensure_device();
        upload_model(h, U, V, user_bias, item_bias, global_bias);
    });
}

extern "C" int mi355rec_scorer_recommend(mi355rec_scorer_t h, const int32_t *user_ids, int32_t n, int32_t cutoff,
                                         int32_t remove_seen, const uint8_t *item_allowed, int32_t *ranked, float *scores) {
    return guarded([&] {
        MI_REQUIRE(h && user_ids && ranked, "NULL argument");
        MI_REQUIRE(n > 0, "empty user batch");
        MI_REQUIRE(cutoff >= 1 && cutoff <= h->n_items, "cutoff must be in [1, n_items]");
        for (int i = 0; i < n; ++i)
            MI_REQUIRE(user_ids[i] >= 0 && user_ids[i] < h->n_users, "Cold users not allowed. Users in trained model are %d, "
                       "requested prediction for user %d", h->n_users, user_ids[i]);
        ensure_device();
        hipStream_t s = h->stream;
        if (h->users.count < (size_t)n) h->users.alloc(n);
        if (h->ranked.count < (size_t)n * cutoff) h->ranked.alloc((size_t)n * cutoff);
        if (h->scores.count < (size_t)n * h->n_items) h->scores.alloc((size_t)n * h->n_items);
        MI_HIP(hipMemcpyAsync(h->users.ptr, user_ids, sizeof(int) * n, hipMemcpyHostToDevice, s));
        if (item_allowed) MI_HIP(hipMemcpyAsync(h->allowed.ptr, item_allowed, h->n_items, hipMemcpyHostToDevice, s));
        h->call_timer.start(s);
        ScoreParams sp{};
        sp.n_users = h->n_users; sp.n_items = h->n_items; sp.k = h->k; sp.use_bias = h->use_bias;
        sp.U = h->U.ptr; sp.V = h->V.ptr; sp.bu = h->bu.ptr; sp.bi = h->bi.ptr; sp.mu = h->mu;
        sp.users = h->users.ptr; sp.n_batch = n; sp.scores = h->scores.ptr;
        hipExtLaunchKernelGGL(score_gemm_kernel, dim3(div_up(h->n_items, TN), div_up(n, TM)), dim3(256), 0, s, h->gemm_timer.t0,
                              h->gemm_timer.t1, 0, sp);
        if (fits_lds_rank(h->n_items, cutoff)) {
            RankParams rp{};
            rp.n_items = h->n_items; rp.n_pad = (h->n_items + 3) & ~3; rp.cutoff = cutoff;
            rp.remove_seen = remove_seen;
            rp.users = h->users.ptr; rp.seen_ptr = h->seen_ptr.ptr; rp.seen_idx = h->seen_idx.ptr;
            rp.allowed = item_allowed ? h->allowed.ptr : nullptr;
            rp.scores = h->scores.ptr; rp.ranked = h->ranked.ptr; rp.write_back = scores != nullptr;
            const size_t lds = (size_t)rp.n_pad * 4 + (size_t)AUX_WORDS * 4;
            auto k = score_rank_kernel<1024>;
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k, dim3(n), dim3(1024), lds, s, rp);
        } else {
            h->wide.rank(s, h->scores.ptr, n, h->n_items, cutoff, h->users.ptr, h->seen_ptr.ptr, h->seen_idx.ptr,
                         item_allowed ? h->allowed.ptr : nullptr, remove_seen, h->ranked.ptr);
        }
        MI_HIP(hipGetLastError());
        h->call_timer.stop(s);
        h->ranked.download(ranked, (size_t)n * cutoff, s);
        if (scores) h->scores.download(scores, (size_t)n * h->n_items, s);
        MI_HIP(hipStreamSynchronize(s));
        h->stats = mi355rec_stats{};
        h->stats.call_ms = h->call_timer.elapsed_ms();
        h->stats.kernel_ms = h->gemm_timer.elapsed_ms();
        h->stats.n_launches = 1;
        h->stats.n_timed = 1;
        h->stats.n_units = n;
        h->stats.algorithmic_flops = 2.0 * (double)n * h->n_items * h->k;
        h->stats.algorithmic_bytes = 4.0 * ((double)n * h->k + (double)h->n_items * h->k + (double)n * h->n_items);
    });
}

extern "C" int mi355rec_scorer_get_stats(mi355rec_scorer_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_scorer_destroy(mi355rec_scorer_t h) {
    if (!h) return;
    ReleaseScope scope(h->stream);
    delete h;
}

// ------------------------------------------------------------------------------------------------------
// Similarity-model scoring: scores[u] = A[u, :] . B with A and B sparse (CSR).
//   ItemKNN / SLIM  (BaseItemSimilarityMatrixRecommender._compute_item_score, Base/BaseSimilarityMatrixRecommender.py:73-92):
//                   A = URM_train, B = W_sparse                         -> user_profile . W
//   UserKNN         (BaseUserSimilarityMatrixRecommender._compute_item_score, :101-116):  A = W_sparse, B = URM_train
// One workgroup per user: the dense score row lives in LDS (it IS dense in the reference: .toarray()), every stored cell
// of A[u] streams one row of B into it with LDS float atomics, then the same filter + top-cutoff as the factor models.
// ------------------------------------------------------------------------------------------------------
namespace mi355rec {
namespace {

struct SpScoreParams {
    int n_out, n_pad, cutoff, remove_seen, write_back;
    const int *a_ptr, *a_idx, *b_ptr, *b_idx;
    const float *a_val, *b_val;
    const int *users, *seen_ptr, *seen_idx;
    const unsigned char *allowed;
    float *scores;                  // [n_batch][n_out] when write_back
    int *ranked;
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void spscore_kernel(const SpScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *acc = smem;
    uint32_t *aux = reinterpret_cast<uint32_t *>(smem + p.n_pad);
    __shared__ SelectScratch sc;
    __shared__ uint32_t s_ncand, s_nfinite;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WAVES = THREADS / 64;
    const int b = blockIdx.x, u = p.users[b];
    if (tid == 0) { s_ncand = 0; s_nfinite = 0; }
    for (int j = tid; j < p.n_pad; j += THREADS) acc[j] = 0.f;
    __syncthreads();
    // one wavefront per stored cell of A[u]: its row of B is added, scaled, to the score row
    for (int q = p.a_ptr[u] + wave; q < p.a_ptr[u + 1]; q += WAVES) {
        const int m = p.a_idx[q];
        const float w = p.a_val[q];
        for (int t = p.b_ptr[m] + lane; t < p.b_ptr[m + 1]; t += 64) atomicAdd(&acc[p.b_idx[t]], w * p.b_val[t]);
    }
    __syncthreads();
    if (p.allowed)
        for (int j = tid; j < p.n_out; j += THREADS)
            if (!p.allowed[j]) acc[j] = -INFINITY;
    if (p.remove_seen)
        for (int q = p.seen_ptr[u] + tid; q < p.seen_ptr[u + 1]; q += THREADS) acc[p.seen_idx[q]] = -INFINITY;
    __syncthreads();
    uint32_t nfin = 0;
    float tmax = -INFINITY;
    float *row = p.scores + (size_t)b * p.n_out;
    for (int j = tid; j < p.n_out; j += THREADS) {
        const float v = acc[j];
        nfin += v > -INFINITY;
        tmax = fmaxf(tmax, v);
        if (p.write_back) row[j] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nfin += __shfl_down(nfin, off);
    if (lane == 0 && nfin) atomicAdd(&s_nfinite, nfin);
    for (int w = tid; w < 256; w += THREADS) aux[w] = 0;           // (the bins of block_kth_largest_prefix16)
    __syncthreads();
    int *out = p.ranked + (size_t)b * p.cutoff;
    // (a user with few non-zero scores has a zero among the largest thread maxima: every cell of the row is then "at or above" the
    // bound, the candidate list overflows and the full select below runs as before)
    if (rank_threshold_first<THREADS>(acc, p.n_out, p.cutoff, s_nfinite, tmax, aux, sc, &s_ncand, out)) return;
    block_topk_emit<THREADS>(acc, p.n_out, p.cutoff, s_nfinite, 0u, TOPK_FINITE, aux, sc, &s_ncand, out, nullptr);
}

// the same accumulation with the score row in HBM (rows that do not fit LDS): global float atomics
__global__ __launch_bounds__(1024) void spscore_wide_kernel(const SpScoreParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, u = p.users[b];
    float *row = p.scores + (size_t)b * p.n_out;
    for (int q = p.a_ptr[u] + wave; q < p.a_ptr[u + 1]; q += 16) {
        const int m = p.a_idx[q];
        const float w = p.a_val[q];
        for (int t = p.b_ptr[m] + lane; t < p.b_ptr[m + 1]; t += 64) atomicAdd(&row[p.b_idx[t]], w * p.b_val[t]);
    }
}

}  // namespace
}  // namespace mi355rec

struct mi355rec_spscorer {
    WideRanker wide;
    int n_users = 0, n_mid = 0, n_out = 0;
    hipStream_t stream = nullptr;
    StreamTimer timer;
    DeviceBuffer<int> a_ptr, a_idx, b_ptr, b_idx, seen_ptr, seen_idx, users, ranked;
    DeviceBuffer<float> a_val, b_val, scores;
    DeviceBuffer<unsigned char> allowed;
    mi355rec_stats stats{};
    double nnz_a = 0, nnz_b = 0;

    ~mi355rec_spscorer() {
        if (stream) (void)hipStreamSynchronize(stream);
        timer.destroy();
        ReleaseScope::forget(stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" int mi355rec_spscorer_create(mi355rec_spscorer_t *out, int32_t n_users, int32_t n_mid, int32_t n_out,
                                        const int32_t *a_indptr, const int32_t *a_indices, const float *a_data,
                                        const int32_t *b_indptr, const int32_t *b_indices, const float *b_data,
                                        const int32_t *seen_indptr, const int32_t *seen_indices) {
    return guarded([&] {
        MI_REQUIRE(out && a_indptr && a_indices && a_data && b_indptr && b_indices && b_data && seen_indptr && seen_indices, "NULL argument");
        MI_REQUIRE(n_users > 0 && n_mid > 0 && n_out > 0, "empty model");
        ensure_device();
        std::unique_ptr<mi355rec_spscorer> h(new mi355rec_spscorer());
        h->n_users = n_users; h->n_mid = n_mid; h->n_out = n_out;
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->timer.init();
        hipStream_t s = h->stream;
        const size_t na = (size_t)a_indptr[n_users], nb = (size_t)b_indptr[n_mid], ns = (size_t)seen_indptr[n_users];
        h->a_ptr.upload(a_indptr, (size_t)n_users + 1, s);
        h->a_idx.upload(a_indices, na, s);
        h->a_val.upload(a_data, na, s);
        h->b_ptr.upload(b_indptr, (size_t)n_mid + 1, s);
        h->b_idx.upload(b_indices, nb, s);
        h->b_val.upload(b_data, nb, s);
        h->seen_ptr.upload(seen_indptr, (size_t)n_users + 1, s);
        h->seen_idx.upload(seen_indices, ns, s);
        h->allowed.alloc(n_out);
        h->nnz_a = (double)na; h->nnz_b = (double)nb;
        MI_HIP(hipStreamSynchronize(s));
        *out = h.release();
    });
}

extern "C" int mi355rec_spscorer_recommend(mi355rec_spscorer_t h, const int32_t *user_ids, int32_t n, int32_t cutoff,
                                           int32_t remove_seen, const uint8_t *item_allowed, int32_t *ranked, float *scores) {
    return guarded([&] {
        MI_REQUIRE(h && user_ids && ranked, "NULL argument");
        MI_REQUIRE(n > 0, "empty user batch");
        MI_REQUIRE(cutoff >= 1 && cutoff <= h->n_out, "cutoff must be in [1, n_items]");
        for (int i = 0; i < n; ++i) MI_REQUIRE(user_ids[i] >= 0 && user_ids[i] < h->n_users, "user id %d out of range", user_ids[i]);
        ensure_device();
        hipStream_t s = h->stream;
        if (h->users.count < (size_t)n) h->users.alloc(n);
        if (h->ranked.count < (size_t)n * cutoff) h->ranked.alloc((size_t)n * cutoff);
        const bool in_lds = fits_lds_rank(h->n_out, cutoff);
        if ((scores || !in_lds) && h->scores.count < (size_t)n * h->n_out) h->scores.alloc((size_t)n * h->n_out);
        MI_HIP(hipMemcpyAsync(h->users.ptr, user_ids, sizeof(int) * n, hipMemcpyHostToDevice, s));
        if (item_allowed) MI_HIP(hipMemcpyAsync(h->allowed.ptr, item_allowed, h->n_out, hipMemcpyHostToDevice, s));
        SpScoreParams p{};
        p.n_out = h->n_out; p.n_pad = (h->n_out + 3) & ~3; p.cutoff = cutoff;
        p.remove_seen = remove_seen; p.write_back = scores != nullptr;
        p.a_ptr = h->a_ptr.ptr; p.a_idx = h->a_idx.ptr; p.a_val = h->a_val.ptr;
        p.b_ptr = h->b_ptr.ptr; p.b_idx = h->b_idx.ptr; p.b_val = h->b_val.ptr;
        p.users = h->users.ptr; p.seen_ptr = h->seen_ptr.ptr; p.seen_idx = h->seen_idx.ptr;
        p.allowed = item_allowed ? h->allowed.ptr : nullptr;
        p.scores = h->scores.ptr; p.ranked = h->ranked.ptr;
        if (in_lds) {
            const size_t lds = (size_t)p.n_pad * 4 + (size_t)AUX_WORDS * 4;
            auto k = spscore_kernel<1024>;
            MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipExtLaunchKernelGGL(k, dim3(n), dim3(1024), (unsigned)lds, s, h->timer.t0, h->timer.t1, 0, p);
        } else {
            MI_HIP(hipMemsetAsync(h->scores.ptr, 0, sizeof(float) * (size_t)n * h->n_out, s));
            hipExtLaunchKernelGGL(spscore_wide_kernel, dim3(n), dim3(1024), 0, s, h->timer.t0, h->timer.t1, 0, p);
            h->wide.rank(s, h->scores.ptr, n, h->n_out, cutoff, h->users.ptr, h->seen_ptr.ptr, h->seen_idx.ptr, p.allowed, remove_seen,
                         h->ranked.ptr);
        }
        MI_HIP(hipGetLastError());
        h->ranked.download(ranked, (size_t)n * cutoff, s);
        if (scores) h->scores.download(scores, (size_t)n * h->n_out, s);
        MI_HIP(hipStreamSynchronize(s));
        h->stats = mi355rec_stats{};
        h->stats.kernel_ms = h->stats.call_ms = h->timer.elapsed_ms();
        h->stats.n_launches = h->stats.n_timed = 1;
        h->stats.n_units = n;
    });
}

extern "C" int mi355rec_spscorer_get_stats(mi355rec_spscorer_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_spscorer_destroy(mi355rec_spscorer_t h) {
    if (!h) return;
    ReleaseScope scope(h->stream);
    delete h;
}
