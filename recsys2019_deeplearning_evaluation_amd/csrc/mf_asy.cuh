// mf_asy.cuh -- AsySVD: strictly ordered steps on one 1024-thread workgroup.  Included by mf.hip after mf_batch.cuh.
#pragma once

namespace mi355rec {
namespace {

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
