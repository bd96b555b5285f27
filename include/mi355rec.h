/*
 * mi355rec.h -- C ABI of libmi355rec.so, the MI355X (gfx950) implementation of the baseline training
 * kernels of MaurizioFD/RecSys2019_DeepLearning_Evaluation.
 *
 * The reference has no C/FFI boundary on this path: its "operator API" is the Python-level interface of
 * three Cython `cdef class`es and one NumPy loop.  Each group of entry points below replaces one of them;
 * the reference interface it stands in for is cited as file:line (relative to the reference root).  The
 * ctypes stub a maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MI355REC_E_* code otherwise;
 *     mi355rec_last_error() returns a thread-local human-readable message for the last failure;
 *   - host buffers are owned by the caller and only read/written for the duration of the call;
 *     *_create copies its inputs to HBM; handles own device memory and one HIP stream;
 *   - pointers named d_* are DEVICE pointers (e.g. torch tensor .data_ptr()), everything else is host;
 *   - all calls are blocking unless stated; one handle must not be used from two threads at once;
 *   - the HIP context is created lazily by the first *_create in the calling process (fork-safe import);
 *   - indices are int32, values float32; CSR rows must have sorted column indices.
 */
#ifndef MI355REC_H
#define MI355REC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355REC_OK              0
#define MI355REC_E_INVALID      -1   /* bad argument (maps to ValueError on the Python side) */
#define MI355REC_E_HIP          -2   /* HIP runtime failure */
#define MI355REC_E_NO_DEVICE    -3   /* no gfx950 device visible */
#define MI355REC_E_UNSUPPORTED  -4   /* valid in the reference, not (yet) covered by the device path */
#define MI355REC_E_NUMERIC      -5   /* non-finite value / not positive definite */

const char *mi355rec_last_error(void);

/* Raw device buffers of the calling process's device, for callers that keep results on the GPU and exchange them between
 * GPUs themselves (sharding.py: RCCL all-gather of similarity slabs through ctypes, no PyTorch).  to_device: 1 = host -> device,
 * 0 = device -> host, 2 = device -> device; the copies are blocking. */
int mi355rec_device_malloc(void **out, uint64_t bytes);
int mi355rec_device_free(void *p);
int mi355rec_device_memcpy(void *dst, const void *src, uint64_t bytes, int to_device);
int mi355rec_device_synchronize(void);
/* Device blocks that handles have released stay in a per-process cache (hipFree drains the device and costs 0.2 - 0.5 ms per block: a
 * dozen temporaries made up a third of a similarity constructor); at most MI355REC_POOL_BYTES of them (default 8 GiB of the device's 288, 0: no cache).
 * mi355rec_device_trim gives every cached block back to the driver -- for a process that shares the device with another allocator
 * (PyTorch, RCCL) and is done with a phase of fits; *freed_bytes (nullable) receives what was returned. */
int mi355rec_device_trim(uint64_t *freed_bytes);
/* Number of visible HIP devices (0 and MI355REC_OK when none). */
int mi355rec_device_count(int *count);
/* Select the device used by handles created afterwards in this process (one process per GPU: LOCAL_RANK). */
int mi355rec_set_device(int device);
/* "gfx950:..." architecture string of the current device. */
int mi355rec_device_name(char *buf, int buf_len);

/* Timing of the last run/compute call on a handle, measured with HIP events on the handle's own stream
 * (bench.py's throughput and roofline figures come from here).
 *   call_ms    events recorded on the stream immediately before the first and after the last kernel of the call;
 *   kernel_ms  SUM of the durations of the timed launches of the DOMINANT kernel (mf: gradient kernel, sim: column
 *              kernel, slim: step kernel, ials: row-solve kernel), each measured by the start/stop events the
 *              dispatch itself carries (hipExtLaunchKernelGGL) -- the same begin/end timestamps rocprofv3
 *              --kernel-trace reports; n_timed of the n_launches launches were timed (see *_set_profiling). */
typedef struct {
    double  call_ms;
    double  kernel_ms;
    int64_t n_launches;     /* launches of the dominant kernel in the call */
    int64_t n_timed;        /* how many of them carry timing events (kernel_ms / n_timed = average launch duration) */
    int64_t n_units;        /* units processed: samples (mf, slim), columns (sim), row solves (ials) */
    double  algorithmic_bytes; /* ALGORITHMIC bytes of the WHOLE call (definition: DESIGN.md section 4) */
    double  algorithmic_flops; /* ALGORITHMIC flops of the whole call (ials), 0 elsewhere */
    double  loss;           /* cumulative x_uij^2 / squared error of the call where defined, else 0 */
} mi355rec_stats;

/* ------------------------------------------------------------------------------------------------------
 * Compute_Similarity  (Base/Similarity/Cython/Compute_Similarity_Cython.pyx:51  cdef class
 * Compute_Similarity_Cython; __init__ :72-213; compute_similarity(start_col, end_col) :411-607)
 * ---------------------------------------------------------------------------------------------------- */

enum {                       /* `similarity=` strings of Compute_Similarity_Cython.__init__ (.pyx:118-143) */
    MI355REC_SIM_COSINE     = 0,
    MI355REC_SIM_ADJUSTED   = 1,
    MI355REC_SIM_ASYMMETRIC = 2,
    MI355REC_SIM_PEARSON    = 3,
    MI355REC_SIM_JACCARD    = 4,   /* == "tanimoto" */
    MI355REC_SIM_DICE       = 5,
    MI355REC_SIM_TVERSKY    = 6,
    MI355REC_SIM_EUCLIDEAN  = 7    /* Base/Similarity/Compute_Similarity_Euclidean.py:16 (reached through the dispatcher,
                                    * Compute_Similarity.py:59-62): 1 / (f(distance) + shrink + 1e-9), float32 arithmetic */
};

enum {                       /* `similarity_from_distance_mode=` of Compute_Similarity_Euclidean (:46-57, :189-196) */
    MI355REC_EUCLID_LIN = 0, /* f(d) = d         */
    MI355REC_EUCLID_LOG = 1, /* f(d) = log(d+1)  */
    MI355REC_EUCLID_EXP = 2  /* f(d) = exp(d)    */
};

typedef struct {
    int32_t topK;            /* 0 = dense output only (.pyx:507-510); clamped to n_cols like .pyx:146 */
    int32_t shrink;          /* already truncated to int, as `cdef int shrink` does (.pyx:64) */
    int32_t normalize;
    int32_t similarity;      /* MI355REC_SIM_* */
    float   asymmetric_alpha, tversky_alpha, tversky_beta;   /* `cdef float` in the reference (.pyx:65) */
    int32_t unit_column_side;   /* 1: the value of (row u, column c) on the COLUMN side of the product is taken as 1, i.e.
                                 * out(c, j) = sum over the rows u storing c of data[u, j] -- the boolean-transpose product of
                                 * the graph recommenders (GraphBased/P3alphaRecommender.py:69-104); 0: the plain self-product */
    int32_t normalize_avg_row;  /* MI355REC_SIM_EUCLIDEAN only: squared distance divided by n_rows (Euclidean.py:181-182);
                                 * `normalize` then means squared distance / (|a| |b|) (:178-179) */
    int32_t euclidean_mode;     /* MI355REC_EUCLID_* */
    /* Optional pre-pass: the BM25 / TF-IDF re-weighting the KNN recommenders apply to the matrix before the build
     * (Base/IR_feature_weighting.py:13 okapi_BM_25, :55 TF_IDF; KNN/ItemKNNCFRecommender.py:40-48, UserKNNCFRecommender.py:40-48),
     * on the stored values already in HBM.  "Documents" are the rows of the matrix the reference hands to those functions:
     * for ItemKNN (URM.T weighted, URM built) the COLUMNS of dataMatrix, for UserKNN (URM.T weighted, URM.T built) its ROWS. */
    int32_t feature_weighting;      /* MI355REC_WEIGHT_* */
    int32_t weighting_documents;    /* 0: documents = columns of dataMatrix, 1: documents = rows */
    float   bm25_k1, bm25_b;        /* okapi_BM_25(K1 = 1.2, B = 0.75) */
    /* How the reference sums the squares behind the column norms, `dataMatrix.power(2).sum(axis=0)` (.pyx:169,
     * Compute_Similarity_Euclidean.py:112): in the matrix's own dtype -- float32 for a URM -- and in an ORDER that depends on the
     * sparse format SciPy finds: a CSR matrix is summed as ones @ X, i.e. every column's squares are added one after the other in
     * row order into a float32 (heavy columns of jittered ratings lose 2e-4 of their norm that way), a CSC matrix by
     * np.add.reduceat, i.e. first square + NumPy's pairwise float32 sum of the rest.  0: the CSR order (also adjusted cosine,
     * whose pre-pass returns CSR, .pyx:275); 1: the CSC order (also pearson, whose pre-pass returns CSC, .pyx:234). */
    int32_t norm_sum_order;
    int32_t reserved;
} mi355rec_sim_config;

enum { MI355REC_WEIGHT_NONE = 0, MI355REC_WEIGHT_BM25 = 1, MI355REC_WEIGHT_TFIDF = 2 };

typedef struct mi355rec_sim *mi355rec_sim_t;

/* dataMatrix (n_rows x n_cols) as CSR; similarities are computed between its COLUMNS.  row_weights is
 * nullable (length n_rows; with MI355REC_SIM_EUCLIDEAN it needs n_rows == n_cols, Compute_Similarity_Euclidean.py:174-175, else
 * MI355REC_E_INVALID).  Pre-processing (mean-centring / binarisation / column norms, .pyx:153-192),
 * the CSR->CSC transposition (.pyx:198-207) and the cost-ordered column schedule are done on the device. */
int mi355rec_sim_create(mi355rec_sim_t *out, const mi355rec_sim_config *cfg, int32_t n_rows, int32_t n_cols,
                        const int32_t *csr_indptr, const int32_t *csr_indices, const float *csr_data,
                        const float *row_weights);
/* The same constructor for a dataMatrix that is ALREADY in device memory (d_* are device pointers of the calling process's GPU;
 * row_weights stays a host array): the handle copies the three arrays at HBM speed instead of uploading 8 B per stored value over
 * PCIe -- 2.9 of the 6.7 ms the constructor takes at ML-20M shape.  The reference's hyper-parameter search fits hundreds of models
 * on one URM_train (ParameterTuning/SearchAbstractClass.py:253-262, one `fit` per configuration): the matrix is uploaded once
 * (`ResidentURM` in the Python front-end) and every ItemKNN fit starts from the resident copy (P3alpha / RP3beta build from a
 * row-normalised, re-weighted matrix of their own and upload that).  The caller keeps
 * ownership of the arrays; they are not modified and may be freed as soon as the call returns. */
int mi355rec_sim_create_resident(mi355rec_sim_t *out, const mi355rec_sim_config *cfg, int32_t n_rows, int32_t n_cols,
                                 const int32_t *d_csr_indptr, const int32_t *d_csr_indices, const float *d_csr_data,
                                 const float *row_weights);
/* Multi-GPU partition with equal column COUNTS and equal COST per part: the columns in cost order are dealt to n_parts parts in
 * serpentine order; part `part` computes its columns and writes column mi355rec_sim_part_columns()[q] to row q of the
 * DEVICE slabs (ceil(n_cols / n_parts) x topK each).  Same kernel as mi355rec_sim_compute_device; the contiguous
 * [start_col, end_col) entry points keep the reference's own seam (Compute_Similarity_Cython.pyx:411, 447-451). */
int mi355rec_sim_compute_part_device(mi355rec_sim_t h, int32_t part, int32_t n_parts, int32_t *d_nbr_idx, float *d_nbr_val);
/* The same for rows [slot_first, slot_first + slot_count) of the part only, written to rows 0 .. slot_count - 1 of the slabs handed
 * in: a sharded build computes its part in chunks so that the all-gather of a finished chunk travels while the next one is built. */
int mi355rec_sim_compute_part_chunk_device(mi355rec_sim_t h, int32_t part, int32_t n_parts, int32_t slot_first, int32_t slot_count,
                                           int32_t *d_nbr_idx, float *d_nbr_val);
/* The exchange format of the sharded build when n_cols <= 65 535 (6 bytes per neighbour instead of 8): n_cells float32 values followed by
 * n_cells 16-bit neighbour ids (0xFFFF = the -1 of an empty slot), padded to a multiple of 4 bytes -- n_cells + (n_cells + 1) / 2 words.
 * pack: from the (idx, val) slabs a compute_*_device call filled; unpack: back into such slabs.  Asynchronous on the handle's stream.
 * What travels is what Compute_Similarity_Cython.pyx:593-607 would assemble into W_sparse on one process. */
int mi355rec_sim_pack_slab_device(mi355rec_sim_t h, const int32_t *d_nbr_idx, const float *d_nbr_val, int64_t n_cells, void *d_packed);
int mi355rec_sim_unpack_slab_device(mi355rec_sim_t h, const void *d_packed, int64_t n_cells, int32_t *d_nbr_idx, float *d_nbr_val);
/* The columns of a part in output-row order (columns may be NULL: only the count). */
int mi355rec_sim_part_columns(mi355rec_sim_t h, int32_t part, int32_t n_parts, int32_t *columns, int32_t *n_columns);
/* The re-weighted stored values (feature_weighting != NONE), in the order of the csr_data passed to mi355rec_sim_create: what the
 * reference recommender keeps as its URM_train afterwards (ItemKNNCFRecommender.py:42-43). */
int mi355rec_sim_get_weighted_values(mi355rec_sim_t h, float *csr_data);
/* Columns [start_col, end_col): for local column c the topK (neighbour, value) pairs in descending value
 * order at nbr_idx/nbr_val[(c - start_col) * topK ...], padded with (-1, 0).  Zero similarities are never
 * emitted (.pyx:555). */
int mi355rec_sim_compute(mi355rec_sim_t h, int32_t start_col, int32_t end_col, int32_t *nbr_idx, float *nbr_val);
/* Same, results left in device memory (for the RCCL gather of the column-sharded build); asynchronous on
 * the handle's stream -- call mi355rec_sim_sync before another stream/library reads the buffers. */
int mi355rec_sim_compute_device(mi355rec_sim_t h, int32_t start_col, int32_t end_col, int32_t *d_nbr_idx, float *d_nbr_val);
/* Same columns, assembled on the device into the CSR matrix the reference returns (.pyx:603-605): row = neighbour,
 * column = source item, column indices ascending inside a row.  indptr: n_cols + 1; indices / data: capacity
 * (end_col - start_col) * topK, the first *nnz entries are written.  (The host-side COO -> CSR conversion of 2.7 M cells
 * takes ~200 ms in SciPy -- more than ten times the whole device build at ML-20M shape.) */
int mi355rec_sim_compute_csr(mi355rec_sim_t h, int32_t start_col, int32_t end_col, int32_t *indptr, int32_t *indices,
                             float *data, int64_t *nnz);
/* topK == 0 variant (.pyx:507-510): W[j * ld + (c - start_col)] = similarity(j, c); W is host, row-major, ld >= end_col-start_col. */
int mi355rec_sim_compute_dense(mi355rec_sim_t h, int32_t start_col, int32_t end_col, float *W, int64_t ld);
/* cost(c) = sum over users of column c of their profile length: the work of one column; used to cut
 * cost-balanced column ranges for multi-GPU sharding. */
int mi355rec_sim_column_costs(mi355rec_sim_t h, int64_t *cost /* n_cols */);
/* Schedule of the last compute call: work items queued, columns that were split over several workgroups, and the
 * number of parts they were split into (diagnostics; a heavy column is accumulated by several workgroups). */
int mi355rec_sim_schedule_info(mi355rec_sim_t h, int32_t *n_items, int32_t *n_split_columns, int32_t *n_parts);
/* Type of the in-LDS column accumulator this handle's builds use: 0 = uint32 co-occurrence counts (all stored values 1.0, no
 * row_weights), 3 = exact int32 sums (every stored value times 2^s, s <= 3, is a small integer -- star ratings, half stars -- and no
 * row_weights / mean-centring: the products scaled by *fixed_scale = 4^s are integers; MI355REC_SIM_NO_INT32=1 disables it),
 * 1 = int64 fixed point (real-valued data whose products, scaled by the power of two *fixed_scale, keep every sum
 * inside 62 bits with a worst-case rounding below 1e-6 of the smallest normalised result), 2 = float64 sums like the reference's
 * double array (Compute_Similarity_Cython.pyx:363; chosen when no such scale exists, or with MI355REC_SIM_F64_SUMS=1 in the
 * environment at create time).  Diagnostics for the parity tests. */
int mi355rec_sim_accumulator_info(mi355rec_sim_t h, int32_t *kind, double *fixed_scale);
/* Selection path of the last compute call (diagnostics for the parity tests): columns whose top-K was found threshold-first
 * (per-thread maxima -> K-th largest maximum -> exact division of the survivors only; csrc/sim.hip), the survivors those columns
 * ranked in total, and columns that went back to the full normalise + radix select after their scan (more survivors than the
 * candidate buffer holds).  Columns with fewer than topK positive cells, accumulator tiles, 8-byte cells, topK == 0 and the
 * Euclidean cell map always take the full path and are counted nowhere.  MI355REC_SIM_FAST_TOPK=0 switches the first path off. */
int mi355rec_sim_selection_info(mi355rec_sim_t h, int64_t *threshold_first_columns, int64_t *candidates, int64_t *fallbacks);
int mi355rec_sim_sync(mi355rec_sim_t h);
/* The rate the column kernel's accumulation is priced against (bench.py's `roofline.peak` of the ItemKNN build), measured on the device at
 * hand: ds_add_u32 lane-adds per second of the whole device on uniformly random cells of a 128 KiB LDS array, one 1024-thread workgroup
 * per CU (the loop of scripts/micro/lds_atomics.hip).  ~10 ms. */
int mi355rec_lds_atomic_rate(double *lane_adds_per_second);
int mi355rec_sim_get_stats(mi355rec_sim_t h, mi355rec_stats *stats);
void mi355rec_sim_destroy(mi355rec_sim_t h);

/* ------------------------------------------------------------------------------------------------------
 * Matrix-factorisation SGD epochs  (MatrixFactorization/Cython/MatrixFactorization_Cython_Epoch.pyx:50
 * cdef class MatrixFactorization_Cython_Epoch; __init__ :95-148; epochIteration_Cython :273;
 * get_USER_factors ... get_GLOBAL_bias :685-702)
 * ---------------------------------------------------------------------------------------------------- */

enum { MI355REC_MF_BPR = 0, MI355REC_MF_FUNK_SVD = 1, MI355REC_MF_ASY_SVD = 2 };   /* algorithm_name (.pyx:93) */
enum { MI355REC_SGD = 0, MI355REC_ADAGRAD = 1, MI355REC_RMSPROP = 2, MI355REC_ADAM = 3 };  /* sgd_mode (.pyx:92) */
enum { MI355REC_F32 = 0, MI355REC_F64 = 1 };

typedef struct {
    int32_t algorithm;       /* MI355REC_MF_* */
    int32_t n_factors;
    int32_t batch_size;
    int32_t use_bias;
    int32_t sgd_mode;        /* MI355REC_SGD ... */
    /* hyper-parameters are double like the reference's `cdef double` attributes (.pyx:55-56): derived constants
     * such as (1 - gamma) = 0.005 must not be formed from float-rounded inputs */
    double  learning_rate;
    double  user_reg, item_reg, bias_reg, positive_reg, negative_reg;
    double  negative_interactions_quota;
    double  gamma, beta_1, beta_2;
    uint64_t random_seed;    /* seeds the on-device counter-based sampler */
    int32_t precision;       /* MI355REC_F32 / MI355REC_F64: storage AND arithmetic type of factors, biases and optimiser
                              * moments on the device, and the element type of U0 / V0.  The reference computes in double
                              * (.pyx:55-66); float32 holds the 1e-5 bar for plain sgd, the adaptive optimisers divide
                              * every gradient component by sqrt(running g^2) + 1e-8 and need float64 state for it. */
    int32_t reserved;
} mi355rec_mf_config;

typedef struct mi355rec_mf *mi355rec_mf_t;

/* URM (n_users x n_items) as CSR with sorted indices.  U0 (n_users x k; n_ITEMS x k for ASY_SVD, whose "user" matrix is
 * the second item-sized matrix Y, .pyx:163-166) and V0 (n_items x k) are the initial factors, row-major, float32 or float64
 * as cfg->precision says (the host draws them exactly like .pyx:174-175 does).  ASY_SVD requires batch_size == 1 (.pyx:395) and runs its nnz + 1 steps per
 * epoch strictly in order. */
int mi355rec_mf_create(mi355rec_mf_t *out, const mi355rec_mf_config *cfg, int32_t n_users, int32_t n_items,
                       const int32_t *indptr, const int32_t *indices, const float *data,
                       const void *U0, const void *V0);
/* n_epochs x epochIteration_Cython(): each epoch is n_users/B+1 (BPR, .pyx:583) or nnz/B+1 (FunkSVD, :289)
 * mini-batches of B samples drawn ON THE DEVICE. */
int mi355rec_mf_run_epochs(mi355rec_mf_t h, int32_t n_epochs);
/* Parity mode: the same arithmetic on a caller-provided sample stream, cut into consecutive mini-batches of
 * batch_size (the last one may be short but is still averaged over batch_size, like .pyx:802).
 * BPR: (u, i, j); FunkSVD: (u, i, rating) with j == NULL. */
int mi355rec_mf_run_samples(mi355rec_mf_t h, const int32_t *u, const int32_t *i, const int32_t *j,
                            const float *rating, int64_t n);
/* Exact multi-GPU mini-batches (MF_BPR with sgd): one epoch with the tasks of every mini-batch split over
 * `world` identical replicas -- same URM, same initial factors, same random_seed, hence the same on-device sample stream and
 * schedule.  Rank `rank` computes the new version of the rows owned by its share of a mini-batch's tasks
 * (mi355rec_mf_shard_batch: kernel + packing of those rows into *d_send, blocking); the caller all-gathers *d_send
 * (bytes_per_rank) into *d_recv (world x bytes_per_rank: ncclAllGather / all_gather_into_tensor); mi355rec_mf_shard_merge copies
 * the other ranks' rows in (blocking).  After every merge all replicas hold bit-identical factors, equal to a single-GPU
 * epoch's.  One exchange per mini-batch: only pays for large batch_size.  */
int mi355rec_mf_shard_begin_epoch(mi355rec_mf_t h, int32_t rank, int32_t world, void **d_send, void **d_recv,
                                  uint64_t *bytes_per_rank, int32_t *n_batches);
int mi355rec_mf_shard_batch(mi355rec_mf_t h, int32_t batch);
int mi355rec_mf_shard_merge(mi355rec_mf_t h, int32_t batch);
int mi355rec_mf_shard_end_epoch(mi355rec_mf_t h);
/* Any output pointer may be NULL.  bu/bi/mu are only written when use_bias. */
int mi355rec_mf_get_factors(mi355rec_mf_t h, float *U, float *V, float *bu, float *bi, float *mu);
/* The same as float64, the type the reference's getters return (.pyx:685-702): the device state itself when it is float64
 * (adagrad / rmsprop / adam, AsySVD), the float32 state widened otherwise. */
int mi355rec_mf_get_factors_f64(mi355rec_mf_t h, double *U, double *V, double *bu, double *bi, double *mu);
/* Copy the (u, i, j|rating) stream drawn by the LAST mi355rec_mf_run_epochs call (at most cap entries);
 * returns the number of samples of that call in *n. */
int mi355rec_mf_get_last_samples(mi355rec_mf_t h, int32_t *u, int32_t *i, int32_t *j, float *rating, int64_t cap, int64_t *n);
/* Time (at most) the first max_timed_launches mini-batch kernel launches of every subsequent call with per-dispatch
 * events; 0 (default) disables it. */
int mi355rec_mf_set_profiling(mi355rec_mf_t h, int32_t max_timed_launches);
/* Diagnostics (handles created with MI355REC_MF_TICKS=1 in the environment): shader-clock stamps of every wavefront of the
 * LAST mini-batch run, 8 words per task slot {entry, header arrived, first rows arrived, list done, exit, list length,
 * own row written, taken-over item rows written}.  *n receives the number of words available (0 when the handle was created without the variable). */
int mi355rec_mf_get_phase_ticks(mi355rec_mf_t h, uint64_t *out, int64_t cap, int64_t *n);
int mi355rec_mf_get_stats(mi355rec_mf_t h, mi355rec_stats *stats);
void mi355rec_mf_destroy(mi355rec_mf_t h);

/* Replica-batched epochs: R independent models trained side by side, the way the reference's hyper-parameter search runs this
 * path (ParameterTuning/run_parameter_search.py:498-503: a multiprocessing Pool, one model per worker; each worker's fit() is a
 * loop over MatrixFactorization_Cython_Epoch.epochIteration_Cython, MatrixFactorization_Cython.py:126).  One epoch of ONE model
 * is a chain of dependent mini-batches of ~3 MB each, which fills a tenth of an MI355X; a group launches mini-batch b of ALL
 * members as one grid, so the chain keeps its length and every link moves R mini-batches.  Members are ordinary handles
 * (not owned by the group, usable on their own between group calls, each with its own factors, hyper-parameters, optimiser,
 * seed and sample stream); they must share what a launch shares: algorithm (MF_BPR or FUNK_SVD), precision, batch_size, the
 * number of mini-batches per epoch and the kernel instance n_factors selects (float32: n_factors % 4 == 0 and <= 64 / <= 128 /
 * <= 256 / <= 512; float64: % 2 and half those bounds).  Every member ends bit-identical to running its epochs alone. */
typedef struct mi355rec_mf_group *mi355rec_mf_group_t;
int mi355rec_mf_group_create(mi355rec_mf_group_t *out, const mi355rec_mf_t *members, int32_t n_members);
/* n_epochs x epochIteration_Cython() of every member (blocking).  Afterwards each member's get_factors / get_last_samples /
 * get_stats (n_units, loss of ITS samples) answer as after its own run_epochs call. */
int mi355rec_mf_group_run_epochs(mi355rec_mf_group_t g, int32_t n_epochs);
int mi355rec_mf_group_set_profiling(mi355rec_mf_group_t g, int32_t max_timed_launches);
/* call_ms / kernel_ms of the last call; n_units, algorithmic_bytes and loss summed over the members. */
int mi355rec_mf_group_get_stats(mi355rec_mf_group_t g, mi355rec_stats *stats);
void mi355rec_mf_group_destroy(mi355rec_mf_group_t g);     /* the members stay alive */

/* ------------------------------------------------------------------------------------------------------
 * SLIM-BPR epoch  (SLIM_BPR/Cython/SLIM_BPR_Cython_Epoch.pyx:59 cdef class SLIM_BPR_Cython_Epoch;
 * __init__ :87-135; epochIteration_Cython :212; get_S :343-391; _dealloc :195)
 * ---------------------------------------------------------------------------------------------------- */

typedef struct {
    int32_t symmetric;       /* 1: S[i,s] aliases S[s,i] (Triangular_Matrix, .pyx:1290-1330) */
    int32_t sgd_mode;
    double  learning_rate, li_reg, lj_reg;
    double  gamma, beta_1, beta_2;
    uint64_t random_seed;
    int32_t precision;       /* MI355REC_F32 / MI355REC_F64: type of the dense store's cells and per-item optimiser cells on the device
                              * (the reference is double throughout).  Rows an owning workgroup keeps in LDS for a launch are float32
                              * there; the symmetric store holds float32 values in 8-byte {value, tag} cells, computes in float64
                              * and keeps the optimiser cells as float64 (two float32 halves) whatever this field says */
    int32_t train_with_sparse_weights;   /* 1: the semantics of the Sparse_Matrix_Tree_CSR store (.pyx:582-1030) on the dense
                              * device array: a cell "has a node" once it has been written; rebalance_tree(TopK) (.pyx:320-324,
                              * 785-805) keeps the TopK largest nodes per row after every (n_steps / 5)-th step of an epoch, get_S
                              * selects once more and keeps the selection (.pyx:381-382, 740-780).  Forces symmetric = 0
                              * (.pyx:112-113) and needs MI355REC_F64 */
    int32_t topK;            /* sparse store only: the TopK of the selections; 0 = the reference's `topK = False` (no selection) */
    int32_t reserved;
} mi355rec_slim_config;

typedef struct mi355rec_slim *mi355rec_slim_t;

int mi355rec_slim_create(mi355rec_slim_t *out, const mi355rec_slim_config *cfg, int32_t n_users, int32_t n_items,
                         const int32_t *indptr, const int32_t *indices);
/* n_epochs x epochIteration_Cython() with the wrapper's batch_size = 1 (SLIM_BPR_Cython.py:140): n_users+1
 * strictly ordered samples per epoch, drawn on the device and executed in stream order. */
int mi355rec_slim_run_epochs(mi355rec_slim_t h, int32_t n_epochs);
int mi355rec_slim_run_samples(mi355rec_slim_t h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n);
/* The (u, i, j) stream of the LAST epoch drawn on the device (at most cap entries; *n = its length, 0 before the first epoch). */
int mi355rec_slim_get_last_samples(mi355rec_slim_t h, int32_t *u, int32_t *i, int32_t *j, int64_t cap, int64_t *n);
/* get_S: diagonal zeroed, per-ROW top-K (.pyx:343-391): nbr_idx/nbr_val[(row) * topK ...], descending,
 * (-1, 0) padded, zeros never emitted. */
int mi355rec_slim_get_S_topk(mi355rec_slim_t h, int32_t topK, int32_t *nbr_idx, float *nbr_val);
/* W_sparse = similarityMatrixTopK(get_S(), k = topK) (SLIM_BPR_Cython.py:186-197 at every validation; Base/Recommender_utils.py:55-122): of
 * the per-row selection above, every COLUMN keeps its topK largest non-zero cells (of equal values the highest rows, as the reference's
 * stable ascending sort leaves them), returned as canonical CSR (indptr [n_items + 1], indices / data with room for n_items * topK
 * entries, *nnz = how many were written).  nbr_idx / nbr_val (optional, may be NULL): the row slabs of mi355rec_slim_get_S_topk from
 * the same pass.  n_items <= 65 535. */
int mi355rec_slim_get_W_csr(mi355rec_slim_t h, int32_t topK, int32_t *nbr_idx, float *nbr_val, int32_t *indptr, int32_t *indices,
                            float *data, int64_t *nnz);
/* get_S of the sparse store with topK > 0 (.pyx:343-352, 381-382): the diagonal becomes a node holding zero, every row keeps
 * its TopK largest nodes (the model changes, as in the reference), and the surviving NON-ZERO nodes of row r are listed in
 * column order in nbr_idx/nbr_val[r * topK ...], (-1, 0) padded (from_linked_list_to_python_list :862-875). */
int mi355rec_slim_get_S_sparse(mi355rec_slim_t h, int32_t *nbr_idx, float *nbr_val);
/* Dense S (n_items x n_items, row-major, diagonal zeroed, symmetric mode mirrored). */
int mi355rec_slim_get_S_dense(mi355rec_slim_t h, float *S);
int mi355rec_slim_get_stats(mi355rec_slim_t h, mi355rec_stats *stats);
/* Diagnostics of the last dense-store launch: rows kept in the LDS of an owning workgroup (the busiest items of the stream; 0 on
 * the symmetric / sparse store, for catalogues whose row does not fit the LDS, or when no compute units could be leased) and the
 * number of steps that ran on rows in HBM. */
int mi355rec_slim_schedule_info(mi355rec_slim_t h, int32_t *n_owned_rows, int32_t *n_cold_steps);
void mi355rec_slim_destroy(mi355rec_slim_t h);

/* ------------------------------------------------------------------------------------------------------
 * IALS solve step  (MatrixFactorization/IALSRecommender.py:137 _run_epoch, :170 _update_row)
 * ---------------------------------------------------------------------------------------------------- */

typedef struct mi355rec_ials *mi355rec_ials_t;

/* C (confidence, n_users x n_items) as CSR with float32 data = 1 + alpha*r etc. (IALSRecommender.py:111-123,
 * computed by the host exactly as the reference does).  V0: initial item factors (n_items x k); U0: initial
 * user factors (only cold users keep them; nullable = zeros). */
int mi355rec_ials_create(mi355rec_ials_t *out, int32_t n_users, int32_t n_items, int32_t n_factors, double reg,
                         const int32_t *indptr, const int32_t *indices, const float *confidence,
                         const double *U0, const double *V0);
/* Restrict the row solves of subsequent epochs to users [u0,u1) and items [i0,i1) (multi-GPU sharding);
 * the caller all-gathers the factor shards between the two half-steps through the *_half entry points. */
int mi355rec_ials_run_epochs(mi355rec_ials_t h, int32_t n_epochs);
int mi355rec_ials_user_half(mi355rec_ials_t h, int32_t u0, int32_t u1);
int mi355rec_ials_item_half(mi355rec_ials_t h, int32_t i0, int32_t i1);
/* Device pointers to the float64 factor matrices (n_users x k, n_items x k), for the RCCL all-gather. */
int mi355rec_ials_device_factors(mi355rec_ials_t h, double **d_U, double **d_V);
int mi355rec_ials_sync(mi355rec_ials_t h);
int mi355rec_ials_get_factors(mi355rec_ials_t h, double *U, double *V);
/* Schedule of the last half-step: rows whose Gramian was split over several workgroups (profiles longer than 8192 entries: parts of
 * 4096, the last arriver adds the parts up in part order and solves) and the number of parts (diagnostics). */
int mi355rec_ials_schedule_info(mi355rec_ials_t h, int32_t *n_split_rows, int32_t *n_parts);
int mi355rec_ials_get_stats(mi355rec_ials_t h, mi355rec_stats *stats);
void mi355rec_ials_destroy(mi355rec_ials_t h);

/* ------------------------------------------------------------------------------------------------------
 * Scoring + ranking of factor models  (Base/BaseMatrixFactorizationRecommender.py:38
 * _compute_item_score and the filter/rank half of Base/BaseRecommender.py:131 recommend)
 * ---------------------------------------------------------------------------------------------------- */

typedef struct mi355rec_scorer *mi355rec_scorer_t;

/* U (n_users x k), V (n_items x k) float32 row-major; biases only read when use_bias; the "seen" CSR is URM_train
 * (sorted or not), used by remove_seen. */
int mi355rec_scorer_create(mi355rec_scorer_t *out, int32_t n_users, int32_t n_items, int32_t n_factors,
                           const float *U, const float *V, int32_t use_bias, const float *user_bias, const float *item_bias,
                           float global_bias, const int32_t *seen_indptr, const int32_t *seen_indices);
/* New factor values of the same shapes (after more training epochs). */
int mi355rec_scorer_update(mi355rec_scorer_t h, const float *U, const float *V, const float *user_bias, const float *item_bias,
                           float global_bias);
/* For every user of the batch: scores = U[u] . V^T (+ biases); items with item_allowed[j] == 0 (nullable mask) and, when
 * remove_seen, the user's seen items become -inf; ranked[(row) * cutoff ...] = the cutoff best items in descending score
 * order, -1 padded where fewer finite scores exist.  scores (nullable, n x n_items) receives the filtered score matrix. */
int mi355rec_scorer_recommend(mi355rec_scorer_t h, const int32_t *user_ids, int32_t n, int32_t cutoff, int32_t remove_seen,
                              const uint8_t *item_allowed, int32_t *ranked, float *scores);
int mi355rec_scorer_get_stats(mi355rec_scorer_t h, mi355rec_stats *stats);
void mi355rec_scorer_destroy(mi355rec_scorer_t h);

/* Similarity models: scores[u] = A[u, :] . B, A (n_users x n_mid) and B (n_mid x n_items) CSR float32.
 * ItemKNN / SLIM: A = URM_train, B = W_sparse (Base/BaseSimilarityMatrixRecommender.py:73-92);
 * UserKNN: A = W_sparse, B = URM_train (:101-116).  Filtering and ranking as mi355rec_scorer_recommend. */
typedef struct mi355rec_spscorer *mi355rec_spscorer_t;
int mi355rec_spscorer_create(mi355rec_spscorer_t *out, int32_t n_users, int32_t n_mid, int32_t n_items,
                             const int32_t *a_indptr, const int32_t *a_indices, const float *a_data,
                             const int32_t *b_indptr, const int32_t *b_indices, const float *b_data,
                             const int32_t *seen_indptr, const int32_t *seen_indices);
int mi355rec_spscorer_recommend(mi355rec_spscorer_t h, const int32_t *user_ids, int32_t n, int32_t cutoff, int32_t remove_seen,
                                const uint8_t *item_allowed, int32_t *ranked, float *scores);
int mi355rec_spscorer_get_stats(mi355rec_spscorer_t h, mi355rec_stats *stats);
void mi355rec_spscorer_destroy(mi355rec_spscorer_t h);

#ifdef __cplusplus
}
#endif
#endif /* MI355REC_H */
