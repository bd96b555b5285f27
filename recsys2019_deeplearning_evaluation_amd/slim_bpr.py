"""SLIM-BPR on MI355X: host front-end of the slim_* entry points of libmi355rec.so.

Mirrors
  SLIM_BPR_Cython_Epoch   SLIM_BPR/Cython/SLIM_BPR_Cython_Epoch.pyx:59 (ctor :87-135, epochIteration_Cython :212,
                          get_S :343-391, _dealloc :195)
  SLIM_BPR_Cython         SLIM_BPR/Cython/SLIM_BPR_Cython.py:50 (fit :78-171)
The dense and the symmetric (triangular) weight stores are on the device.  The sparse-tree training mode
(train_with_sparse_weights=True, Sparse_Matrix_Tree_CSR :582) changes the learned model through its periodic per-row
top-K selection (rebalance_tree :785, at the fifths of every epoch :320-324, and again inside get_S :381-382); those
SEMANTICS run on the same dense device array (cells know whether they "have a node"), float64 like the reference.
As in the reference wrapper the epoch object is always driven with batch_size = 1 (SLIM_BPR_Cython.py:140).
"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sps

from . import _native as N
from .recommender_base import (BaseItemSimilarityMatrixRecommender, Incremental_Training_Early_Stopping, check_matrix,
                               similarityMatrixTopK)
from .scoring import GpuSimilarityScoringMixin


def rows_slabs_to_csr(nbr_idx, nbr_val, n):
    """(n, topK) per-ROW slabs (-1 padded) -> csr_matrix (n, n) float64 like get_S."""
    keep = nbr_idx >= 0
    indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
    return sps.csr_matrix((nbr_val[keep].astype(np.float64), nbr_idx[keep], indptr), shape=(n, n))


class SLIM_BPR_MI355X_Epoch:
    def __init__(self, URM_mask, train_with_sparse_weights=False, final_model_sparse_weights=True, learning_rate=0.01,
                 li_reg=0.0, lj_reg=0.0, batch_size=1, topK=150, symmetric=True, verbose=False, random_seed=None,
                 sgd_mode="adam", gamma=0.995, beta_1=0.9, beta_2=0.999, precision="auto"):
        if sgd_mode not in N.SGD_MODE_CODES:
            raise ValueError("Value for 'sgd_mode' not recognized. Acceptable values are {}, provided was '{}'".format(
                list(N.SGD_MODE_CODES), sgd_mode))
        if batch_size != 1:
            raise NotImplementedError("SLIM_BPR: the reference wrapper always trains with batch_size=1; so does the device path")
        self.train_with_sparse_weights = bool(train_with_sparse_weights)
        if topK is not False and topK is not None and topK < 0:
            raise ValueError("TopK not valid. Acceptable values are either False or a positive integer value. "
                             "Provided value was '{}'".format(topK))
        if precision == "auto":         # dense store: float64 cells and optimiser cells for the adaptive modes (the reference is double
            # throughout); the sparse store's selections compare values, so it keeps the reference's float64 as well.  The symmetric
            # store always holds float32 values in {value, tag} cells and computes in float64 (include/mi355rec.h)
            precision = "fp32" if sgd_mode == "sgd" and not self.train_with_sparse_weights else "fp64"
        if precision not in N.PRECISION_CODES:
            raise ValueError("Value for 'precision' not recognized. Acceptable values are {}, provided was '{}'".format(
                ["auto"] + list(N.PRECISION_CODES), precision))
        if self.train_with_sparse_weights:
            if precision != "fp64":
                raise ValueError("train_with_sparse_weights keeps float64 weights (precision 'auto' or 'fp64')")
            symmetric = False               # .pyx:112-113
        self.precision = precision
        URM_mask = check_matrix(URM_mask, "csr")
        URM_mask = URM_mask.sorted_indices()
        self.n_users, self.n_items = URM_mask.shape
        self.topK = topK
        self.symmetric = bool(symmetric)
        self.final_model_sparse_weights = final_model_sparse_weights
        self.verbose = verbose
        seed = int(random_seed) if random_seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        cfg = N.SlimConfig(int(self.symmetric), N.SGD_MODE_CODES[sgd_mode], learning_rate, li_reg, lj_reg, gamma, beta_1,
                           beta_2, seed & (2 ** 64 - 1), N.PRECISION_CODES[precision], int(self.train_with_sparse_weights),
                           int(topK) if topK else 0, 0)
        indptr, indices = N.as_i32(URM_mask.indptr), N.as_i32(URM_mask.indices)
        self._lib = N.load()
        self._h = C.c_void_p()
        N.check(self._lib.mi355rec_slim_create(C.byref(self._h), C.byref(cfg), self.n_users, self.n_items,
                                               N.ptr(indptr), N.ptr(indices)))

    def _dealloc(self):
        if getattr(self, "_h", None):
            self._lib.mi355rec_slim_destroy(self._h)
            self._h = None

    close = _dealloc

    def __del__(self):
        try:
            self._dealloc()
        except Exception:
            pass

    def epochIteration_Cython(self, n_epochs=1):
        N.check(self._lib.mi355rec_slim_run_epochs(self._h, int(n_epochs)))
        if self.verbose:
            st = self.stats()
            print("Processed {} samples in {:.3f} seconds. BPR loss is {:.2E}. Sample per second: {:.0f}".format(
                st["n_units"], st["call_ms"] * 1e-3, st["loss"] / max(1, st["n_units"]), st["n_units"] / max(1e-9, st["call_ms"] * 1e-3)))
            sys.stdout.flush()

    def replay_samples(self, user, pos_item, neg_item):
        u, i, j = N.as_i32(user), N.as_i32(pos_item), N.as_i32(neg_item)
        N.check(self._lib.mi355rec_slim_run_samples(self._h, N.ptr(u), N.ptr(i), N.ptr(j), len(u)))

    def last_epoch_samples(self):
        """(user, pos_item, neg_item) drawn on the device during the last native epoch."""
        n = C.c_int64(0)
        N.check(self._lib.mi355rec_slim_get_last_samples(self._h, None, None, None, 0, C.byref(n)))
        u = np.empty(n.value, np.int32); i = np.empty(n.value, np.int32); j = np.empty(n.value, np.int32)
        N.check(self._lib.mi355rec_slim_get_last_samples(self._h, N.ptr(u), N.ptr(i), N.ptr(j), n.value, C.byref(n)))
        return u, i, j

    def get_S_dense(self):
        S = np.empty((self.n_items, self.n_items), np.float32)
        N.check(self._lib.mi355rec_slim_get_S_dense(self._h, N.ptr(S)))
        return S

    def get_S_slabs(self, topK):
        k = min(int(topK), self.n_items)
        idx = np.empty((self.n_items, k), np.int32); val = np.empty((self.n_items, k), np.float32)
        N.check(self._lib.mi355rec_slim_get_S_topk(self._h, k, N.ptr(idx), N.ptr(val)))
        return idx, val

    def selects_rows(self):
        """Does get_S() return the per-row top-K selection as a csr_matrix (rather than the dense array, the whole matrix or the
        sparse store's own selection)?"""
        return bool(self.topK) and not self.train_with_sparse_weights and (self.symmetric or self.final_model_sparse_weights)

    def get_S_and_W(self):
        """(get_S(), similarityMatrixTopK(get_S(), k=topK) as canonical csr float32) in ONE device pass (mi355rec_slim_get_W_csr): what
        SLIM_BPR_Cython.py:186-197 computes at every validation.  Only where get_S() is the per-row selection (selects_rows())."""
        assert self.selects_rows() and self.n_items <= 65535
        k = min(int(self.topK), self.n_items)
        idx = np.empty((self.n_items, k), np.int32); val = np.empty((self.n_items, k), np.float32)
        indptr = np.empty(self.n_items + 1, np.int32); indices = np.empty(self.n_items * k, np.int32); data = np.empty(self.n_items * k, np.float32)
        nnz = C.c_int64(0)
        N.check(self._lib.mi355rec_slim_get_W_csr(self._h, k, N.ptr(idx), N.ptr(val), N.ptr(indptr), N.ptr(indices), N.ptr(data), C.byref(nnz)))
        W = sps.csr_matrix((data[:nnz.value], indices[:nnz.value], indptr), shape=(self.n_items, self.n_items))
        W.has_sorted_indices = True
        return rows_slabs_to_csr(idx, val, self.n_items), W

    def get_S(self):
        """Same return convention as the reference (.pyx:343-391): csr with per-row top-K, or the dense array when
        final_model_sparse_weights is False on the dense store, or the full matrix as csr when topK is False."""
        if self.train_with_sparse_weights:
            # get_scipy_csr of the tree (.pyx:363-366, 381-382): the non-zero nodes, after (and keeping) the per-row selection
            if not self.topK:
                return sps.csr_matrix(self.get_S_dense().astype(np.float64))
            idx = np.empty((self.n_items, int(self.topK)), np.int32); val = np.empty((self.n_items, int(self.topK)), np.float32)
            N.check(self._lib.mi355rec_slim_get_S_sparse(self._h, N.ptr(idx), N.ptr(val)))
            return rows_slabs_to_csr(idx, val, self.n_items)
        if not self.topK:
            S = self.get_S_dense().astype(np.float64)
            return S if (not self.symmetric and not self.final_model_sparse_weights) else sps.csr_matrix(S)
        if not self.symmetric and not self.final_model_sparse_weights:
            return self.get_S_dense().astype(np.float64)
        idx, val = self.get_S_slabs(self.topK)
        return rows_slabs_to_csr(idx, val, self.n_items)

    def stats(self):
        st = N.Stats()
        N.check(self._lib.mi355rec_slim_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def schedule_info(self):
        """(rows an owning workgroup kept in LDS, steps that ran on rows in HBM) of the last dense-store launch."""
        a, b = C.c_int32(0), C.c_int32(0)
        N.check(self._lib.mi355rec_slim_schedule_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


class _SLIMLogic:
    """Drop-in for SLIM_BPR_Cython."""
    RECOMMENDER_NAME = "SLIM_BPR_Recommender"

    def __init__(self, URM_train, verbose=True, free_mem_threshold=0.5):
        super(_SLIMLogic, self).__init__(URM_train, verbose=verbose)
        assert 0.0 <= free_mem_threshold <= 1.0, \
            "SLIM_BPR_Recommender: free_mem_threshold must be between 0.0 and 1.0, provided was '{}'".format(free_mem_threshold)
        self.n_users, self.n_items = self.URM_train.shape
        self.free_mem_threshold = free_mem_threshold

    def fit(self, epochs=300, positive_threshold_BPR=None, train_with_sparse_weights=None, symmetric=True, random_seed=None,
            batch_size=1000, lambda_i=0.0, lambda_j=0.0, learning_rate=1e-4, topK=200, sgd_mode="adagrad", gamma=0.995,
            beta_1=0.9, beta_2=0.999, **earlystopping_kwargs):
        self.symmetric = symmetric
        # the reference picks dense vs sparse-tree from free host RAM (.py:97-114); on the device S always lives
        # densely in HBM (288 GB), so "auto" means dense.  train_with_sparse_weights=True selects the sparse store's
        # SEMANTICS (periodic per-row top-K selection), still on the dense device array.
        self.train_with_sparse_weights = bool(train_with_sparse_weights) if train_with_sparse_weights is not None else False
        URM_train_positive = self.URM_train.copy()
        self.positive_threshold_BPR = positive_threshold_BPR
        self.sgd_mode = sgd_mode
        self.epochs = epochs
        if positive_threshold_BPR is not None:
            URM_train_positive.data = URM_train_positive.data >= positive_threshold_BPR
            URM_train_positive.eliminate_zeros()
            assert URM_train_positive.nnz > 0, "SLIM_BPR_Cython: URM_train_positive is empty, positive threshold is too high"
        if topK is not False and topK < 1:
            raise ValueError("TopK not valid. Acceptable values are either False or a positive integer value. "
                             "Provided value was '{}'".format(topK))
        self.epoch_kernel = SLIM_BPR_MI355X_Epoch(URM_train_positive, train_with_sparse_weights=self.train_with_sparse_weights,
                                                  final_model_sparse_weights=True, topK=topK, learning_rate=learning_rate,
                                                  li_reg=lambda_i, lj_reg=lambda_j, batch_size=1, symmetric=symmetric,
                                                  sgd_mode=sgd_mode, verbose=self.verbose, random_seed=random_seed,
                                                  gamma=gamma, beta_1=beta_1, beta_2=beta_2)
        self.topK = topK
        self.batch_size = batch_size
        self.lambda_i = lambda_i
        self.lambda_j = lambda_j
        self.learning_rate = learning_rate
        self.S_incremental = self.epoch_kernel.get_S()
        self.S_best = self.S_incremental.copy()
        self._train_with_early_stopping(epochs, algorithm_name=self.RECOMMENDER_NAME, **earlystopping_kwargs)
        self.get_S_incremental_and_set_W()
        self.epoch_kernel._dealloc()
        sys.stdout.flush()

    def _prepare_model_for_validation(self):
        self.get_S_incremental_and_set_W()

    def _update_best_model(self):
        self.S_best = self.S_incremental.copy()

    def _run_epoch(self, num_epoch):
        self.epoch_kernel.epochIteration_Cython()

    def get_S_incremental_and_set_W(self):
        kernel = self.epoch_kernel
        if getattr(kernel, "selects_rows", None) and kernel.selects_rows() and kernel.n_items <= 65535 and \
                os.environ.get("MI355REC_SLIM_HOST_TOPK") != "1":
            # the column selection of .py:196 next to the row selection, on the device (0.09 s per validation on the host at ML-20M size)
            self.S_incremental, self.W_sparse = kernel.get_S_and_W()
            return
        self.S_incremental = kernel.get_S()
        if self.train_with_sparse_weights or not self.topK:          # .py:193-195: the tree's get_S has selected already
            self.W_sparse = self.S_incremental
        else:
            self.W_sparse = similarityMatrixTopK(self.S_incremental, k=self.topK)
        self.W_sparse = check_matrix(self.W_sparse, format="csr")


class SLIM_BPR_MI355X(_SLIMLogic, GpuSimilarityScoringMixin, BaseItemSimilarityMatrixRecommender, Incremental_Training_Early_Stopping):
    pass
