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

// One translation unit, split by stage (round 4): the device code lives in the four headers below (parameters + optimiser step +
// sampler | schedules | mini-batch kernels | AsySVD), this file holds the handles, kernel selection, graphs and the C ABI.
#include "mf_core.cuh"
#include "mf_schedule.cuh"
#include "mf_batch.cuh"
#include "mf_asy.cuh"


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
        ReleaseScope::forget(stream);
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
    f.fill_all = 1;
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
    hipLaunchKernelGGL(mf_sched_emit_kernel, dim3(div_up(f.batch_size, 256), (unsigned)n_batches), dim3(256), 0, s, f);
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
        ReleaseScope scope(h->stream);
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
    // Look-ahead schedule (all_fast): the sampler, sort, emit and finish launches of epoch e + 1 -- 38 % of a group epoch, and they read
    // nothing the mini-batch kernels write -- run on `side` while the mini-batch launches of epoch e run on `stream`.  What the
    // mini-batch kernels READ of a schedule (task headers, records, pair records, slots in use) exists twice per member: set A is the
    // member's own, set B belongs to the group; the tables below are the A tables with the four pointers swapped.
    struct SetB {
        DeviceBuffer<TaskHeader> tasks;
        DeviceBuffer<int4> recs, slot_recs;
        DeviceBuffer<int> used;
    };
    std::vector<std::unique_ptr<SetB>> set_b;
    DeviceBuffer<unsigned char> table_b;
    DeviceBuffer<FastSchedParams> sched_table_b;
    hipStream_t side = nullptr;
    hipEvent_t ahead_fork = nullptr, ahead_join = nullptr;
    hipEvent_t fork = nullptr;
    std::vector<hipEvent_t> join;
    hipGraphExec_t graph = nullptr;
    bool graph_failed = false;
    mi355rec_stats stats{};

    ~mi355rec_mf_group() {
        if (stream) (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);
        if (ahead_fork) (void)hipEventDestroy(ahead_fork);
        if (ahead_join) (void)hipEventDestroy(ahead_join);
        ReleaseScope::forget(side);
        if (side) (void)hipStreamDestroy(side);
        if (graph) (void)hipGraphExecDestroy(graph);
        timer.destroy();
        dispatch_timers.destroy();
        if (fork) (void)hipEventDestroy(fork);
        for (auto e : join) (void)hipEventDestroy(e);
        ReleaseScope::forget(stream);
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
    hipLaunchKernelGGL(mf_group_sched_emit_kernel, dim3(div_up(f0.batch_size, 256), (unsigned)nb, R), dim3(256), 0, s, sched_table);
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
        if (all_fast) {
            sched[m] = fast_sched_params(h, nb * (long long)h->cfg.batch_size);
            // (the group's mini-batch kernel reads `used` and never looks at a slot past the last workgroup in use: two thirds of the
            // header slots of a fused BPR mini-batch are not zero-filled -- 150 MB of 16-byte stores per 32-model epoch)
            sched[m].fill_all = 0;
        }
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
    const bool ahead = g->all_fast && n_epochs > 1 && !getenv("MI355REC_MF_GROUP_NO_LOOKAHEAD");
    bool use_graph = !ahead && nb <= GRAPH_SEGMENT && n_epochs - timed_epochs > 0 && !getenv("MI355REC_NO_GRAPH");
    if (use_graph) {
        group_ensure_graph<T>(g);
        use_graph = g->graph != nullptr;
    }
    if (ahead) {
        // set B and its tables (once per group, again when a member's buffers moved)
        bool fresh = g->set_b.size() != (size_t)R;
        for (int m = 0; m < R && !fresh; ++m) {
            const mi355rec_mf *h = g->members[m];
            const mi355rec_mf_group::SetB &b = *g->set_b[m];
            fresh = b.tasks.count != h->tasks.count || b.recs.count != h->recs.count || b.slot_recs.count != h->slot_recs.count || b.used.count != h->used.count;
        }
        if (fresh) {
            MI_HIP(hipStreamSynchronize(g->stream));
            if (!g->side) MI_HIP(hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking));
            if (!g->ahead_fork) MI_HIP(hipEventCreateWithFlags(&g->ahead_fork, hipEventDisableTiming));
            if (!g->ahead_join) MI_HIP(hipEventCreateWithFlags(&g->ahead_join, hipEventDisableTiming));
            g->set_b.clear();
            for (int m = 0; m < R; ++m) {
                mi355rec_mf *h = g->members[m];
                std::unique_ptr<mi355rec_mf_group::SetB> b(new mi355rec_mf_group::SetB());
                b->tasks.alloc_zero(h->tasks.count, g->stream);
                b->recs.alloc(h->recs.count);
                b->slot_recs.alloc_zero(h->slot_recs.count, g->stream);
                b->used.alloc(h->used.count);
                g->set_b.push_back(std::move(b));
            }
        }
        std::vector<unsigned char> table_b = table;
        std::vector<FastSchedParams> sched_b = sched;
        for (int m = 0; m < R; ++m) {
            MfParams<T> pb;
            memcpy(&pb, table_b.data() + sizeof(MfParams<T>) * (size_t)m, sizeof(pb));
            const mi355rec_mf_group::SetB &b = *g->set_b[m];
            pb.tasks = b.tasks.ptr; pb.recs = b.recs.ptr; pb.slot_recs = b.slot_recs.ptr; pb.used = b.used.ptr;
            memcpy(table_b.data() + sizeof(MfParams<T>) * (size_t)m, &pb, sizeof(pb));
            sched_b[m].tasks = b.tasks.ptr; sched_b[m].recs = b.recs.ptr; sched_b[m].slot_recs = b.slot_recs.ptr; sched_b[m].used = b.used.ptr;
        }
        if (g->table_b.count < table_b.size()) g->table_b.alloc(table_b.size());
        if (g->sched_table_b.count < sched_b.size()) g->sched_table_b.alloc(sched_b.size());
        MI_HIP(hipMemcpyAsync(g->table_b.ptr, table_b.data(), table_b.size(), hipMemcpyHostToDevice, g->stream));
        MI_HIP(hipMemcpyAsync(g->sched_table_b.ptr, sched_b.data(), sizeof(FastSchedParams) * sched_b.size(), hipMemcpyHostToDevice, g->stream));
        MI_HIP(hipStreamSynchronize(g->stream));           // (the two vectors are locals)
    }
    g->timer.start(g->stream);
    if (ahead) {
        const MfParams<T> *tab[2] = {reinterpret_cast<const MfParams<T> *>(g->table.ptr), reinterpret_cast<const MfParams<T> *>(g->table_b.ptr)};
        const FastSchedParams *sch[2] = {g->sched_table.ptr, g->sched_table_b.ptr};
        group_enqueue_schedule<T>(g, tab[0], sch[0], g->stream);                   // the call's first epoch: nothing to hide it behind
        for (long long e = 0; e < n_epochs; ++e) {
            const int cur = (int)(e & 1);
            if (e + 1 < n_epochs) {
                MI_HIP(hipEventRecord(g->ahead_fork, g->stream));                   // (behind epoch e's schedule and epoch e - 1's mini-batches)
                MI_HIP(hipStreamWaitEvent(g->side, g->ahead_fork, 0));
                group_enqueue_schedule<T>(g, tab[cur ^ 1], sch[cur ^ 1], g->side);
                MI_HIP(hipEventRecord(g->ahead_join, g->side));
            }
            group_enqueue_batches<T>(g, tab[cur], e < timed_epochs);
            if (e + 1 < n_epochs) MI_HIP(hipStreamWaitEvent(g->stream, g->ahead_join, 0));
        }
    } else {
        for (long long e = 0; e < n_epochs; ++e) {
            if (e < timed_epochs || !use_graph) group_enqueue_epoch<T>(g, e < timed_epochs);
            else MI_HIP(hipGraphLaunch(g->graph, g->stream));
        }
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

extern "C" void mi355rec_mf_group_destroy(mi355rec_mf_group_t g) {
    if (!g) return;
    ReleaseScope scope(g->stream);
    delete g;
}

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

extern "C" void mi355rec_mf_destroy(mi355rec_mf_t h) {
    if (!h) return;
    ReleaseScope scope(h->stream);       // (a group's launches on the group's stream have been waited for by the call that made them)
    delete h;
}
