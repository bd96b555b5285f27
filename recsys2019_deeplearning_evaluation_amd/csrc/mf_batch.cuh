// mf_batch.cuh -- the mini-batch kernels: one wavefront per task (mf_batch_body), the single-model and the replica-batched launch,
// the any-k kernel, the row exchange of the exact multi-GPU mode, the getters' gather kernels.  Included by mf.hip after mf_schedule.cuh.
#pragma once

namespace mi355rec {
namespace {

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

}  // namespace
}  // namespace mi355rec
