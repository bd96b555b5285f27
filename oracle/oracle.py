"""CPU oracle for the hot paths -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module, and only as
the checker / the reported CPU baseline.  The product package (recsys2019_deeplearning_evaluation_amd)
never imports it and fails loudly when its HIP library is missing.

Python front-end of oracle/oracle.c (plain C restatement, float64 like the reference) plus NumPy
restatements of the pieces that are NumPy in the reference.  Each class mirrors the constructor / method
surface of the reference object it restates so that tests read like the reference's own usage:

  OracleMF      <- MatrixFactorization_Cython_Epoch   (MatrixFactorization/Cython/MatrixFactorization_Cython_Epoch.pyx:50)
  OracleSLIM    <- SLIM_BPR_Cython_Epoch              (SLIM_BPR/Cython/SLIM_BPR_Cython_Epoch.pyx:59; dense + symmetric stores)
  OracleSimilarity <- Compute_Similarity_Cython       (Base/Similarity/Cython/Compute_Similarity_Cython.pyx:51)
  oracle_ials_epoch / OracleIALS <- IALSRecommender   (MatrixFactorization/IALSRecommender.py:137-210)
  oracle_similarity_topk_rows <- similarityMatrixTopK / Triangular_Matrix.get_scipy_csr (Base/Recommender_utils.py:55,
                                                        SLIM_BPR_Cython_Epoch.pyx:1338-1418)

Parity status: pinned against the compiled reference itself (oracle/_ref) and the golden fixtures in
tests/golden/ (tests/test_oracle_vs_reference.py, tests/test_oracle_golden.py).  The reference has no unit
tests for BPR-MF / FunkSVD / SLIM-BPR / IALS; its similarity known-answer matrices
(Base/Similarity/Compute_similarity_test.py) are re-used in tests/test_similarity_known_answers.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, "liboracle.so")
_SRC_PATH = os.path.join(HERE, "oracle.c")
_lib = None

SGD_MODES = {"sgd": 0, "adagrad": 1, "rmsprop": 2, "adam": 3}


def build(force=False):
    """gcc -O2 (the reference's own optimisation level, CythonCompiler/compile_script.py:42)."""
    if (not force and os.path.isfile(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(_SRC_PATH)):
        return _LIB_PATH
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-Wall", _SRC_PATH, "-o", _LIB_PATH, "-lm"],
                   check=True)
    return _LIB_PATH


_p = C.c_void_p
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_mf_create.restype = _p
        L.orc_mf_create.argtypes = [C.c_int32] * 4 + [_p] * 5 + [C.c_double] * 7 + [C.c_int32, C.c_int32] + [C.c_double] * 3
        L.orc_mf_create_rows.restype = _p
        L.orc_mf_create_rows.argtypes = [C.c_int32] * 5 + [_p] * 5 + [C.c_double] * 7 + [C.c_int32, C.c_int32] + [C.c_double] * 3
        L.orc_mf_epoch_asy.argtypes = [_p]
        L.orc_mf_replay_asy.argtypes = [_p, _p, _p, _p, C.c_int64]
        L.orc_mf_destroy.argtypes = [_p]
        L.orc_srand.argtypes = [C.c_uint]
        L.orc_rand.restype = C.c_int
        L.orc_rand_max.restype = C.c_int
        L.orc_mf_set_recorder.argtypes = [_p, _p, _p, _p, _p, C.c_int64]
        L.orc_mf_recorded.restype = C.c_int64
        L.orc_mf_recorded.argtypes = [_p]
        L.orc_mf_get.argtypes = [_p] * 6
        L.orc_mf_loss.restype = C.c_double
        L.orc_mf_loss.argtypes = [_p]
        L.orc_mf_epoch_bpr.argtypes = [_p]
        L.orc_mf_epoch_funk.argtypes = [_p]
        L.orc_mf_replay_bpr.argtypes = [_p, _p, _p, _p, C.c_int64]
        L.orc_mf_replay_funk.argtypes = [_p, _p, _p, _p, C.c_int64]
        L.orc_slim_create.restype = _p
        L.orc_slim_create.argtypes = [C.c_int32, C.c_int32, _p, _p, C.c_int32, C.c_int32] + [C.c_double] * 6
        L.orc_slim_destroy.argtypes = [_p]
        L.orc_slim_set_recorder.argtypes = [_p, _p, _p, _p, C.c_int64]
        L.orc_slim_recorded.restype = C.c_int64
        L.orc_slim_recorded.argtypes = [_p]
        L.orc_slim_epoch.argtypes = [_p]
        L.orc_slim_replay.argtypes = [_p, _p, _p, _p, C.c_int64]
        L.orc_slim_get_S.argtypes = [_p, _p]
        L.orc_slim_set_sparse.argtypes = [_p, C.c_int32]
        L.orc_slim_sparse_get_S.restype = C.c_int64
        L.orc_slim_sparse_get_S.argtypes = [_p, _p]
        L.orc_slim_sparse_cells.argtypes = [_p, _p, _p]
        L.orc_sim_column.restype = C.c_int32
        L.orc_sim_column.argtypes = ([C.c_int32, C.c_int32] + [_p] * 7 + [C.c_int32] * 3 + [_p] * 3
                                     + [C.c_double] * 2 + [_p] * 3)
        L.orc_sim_build.argtypes = ([C.c_int32] * 4 + [_p] * 7 + [C.c_int32] * 3 + [_p] * 3
                                    + [C.c_double] * 2 + [_p] * 2)
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _sorted_csr(URM):
    URM = sps.csr_matrix(URM)
    URM = URM.sorted_indices()
    return URM


# --------------------------------------------------------------------------------------------------
#                                       matrix factorisation
# --------------------------------------------------------------------------------------------------

class OracleMF:
    """Restates MatrixFactorization_Cython_Epoch for algorithm_name in {"MF_BPR", "FUNK_SVD", "ASY_SVD"}.

    Constructor arguments, defaults, RNG seeding order and factor initialisation follow
    MatrixFactorization_Cython_Epoch.pyx:95-188 (np.random.seed(seed); srand(seed); U then V from
    np.random.normal(init_mean, init_std_dev)).
    """

    def __init__(self, URM_train, n_factors=1, algorithm_name=None, batch_size=1,
                 negative_interactions_quota=0.5, learning_rate=1e-3, use_bias=False,
                 user_reg=0.0, item_reg=0.0, bias_reg=0.0, positive_reg=0.0, negative_reg=0.0,
                 verbose=False, random_seed=None, init_mean=0.0, init_std_dev=0.1,
                 sgd_mode="sgd", gamma=0.995, beta_1=0.9, beta_2=0.999):
        if sgd_mode not in SGD_MODES:
            raise ValueError("Value for 'sgd_mode' not recognized: %r" % (sgd_mode,))
        if algorithm_name not in ("FUNK_SVD", "MF_BPR", "ASY_SVD"):
            raise ValueError("Value for 'algorithm_name' not recognized by the oracle: %r" % (algorithm_name,))
        if algorithm_name == "ASY_SVD":
            assert batch_size == 1, "Batch size other than 1 not supported for ASY_SVD"
        URM = _sorted_csr(URM_train)
        self.n_users, self.n_items = URM.shape
        self.n_factors = int(n_factors)
        self.batch_size = int(batch_size)
        self.algorithm_name = algorithm_name
        self.use_bias = bool(use_bias)
        self._indptr = np.ascontiguousarray(URM.indptr, dtype=np.int32)
        self._indices = np.ascontiguousarray(URM.indices, dtype=np.int32)
        self._data = np.ascontiguousarray(URM.data, dtype=np.float64)
        L = lib()
        if random_seed is not None:
            np.random.seed(seed=random_seed)
            L.orc_srand(C.c_uint(int(random_seed)))
        # AsySVD keeps TWO item-sized matrices (.pyx:163-166): "USER_factors" is then n_items x k
        self.n_u_rows = self.n_items if algorithm_name == "ASY_SVD" else self.n_users
        U0 = np.random.normal(init_mean, init_std_dev, (self.n_u_rows, self.n_factors)).astype(np.float64)
        V0 = np.random.normal(init_mean, init_std_dev, (self.n_items, self.n_factors)).astype(np.float64)
        self.initial_USER_factors = U0.copy()
        self.initial_ITEM_factors = V0.copy()
        self._h = L.orc_mf_create_rows(self.n_users, self.n_items, self.n_u_rows, self.n_factors, self.batch_size,
                                  _ptr(self._indptr), _ptr(self._indices), _ptr(self._data), _ptr(U0), _ptr(V0),
                                  learning_rate, user_reg, item_reg, bias_reg, positive_reg, negative_reg,
                                  negative_interactions_quota, int(self.use_bias), SGD_MODES[sgd_mode],
                                  gamma, beta_1, beta_2)
        self._rec = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_mf_destroy(self._h)
            self._h = None

    def record_samples(self, capacity):
        """Record the (user, item[, neg item | rating]) stream of subsequent epochs for GPU replay."""
        u = np.zeros(capacity, np.int32); i = np.zeros(capacity, np.int32)
        j = np.zeros(capacity, np.int32); r = np.zeros(capacity, np.float64)
        self._rec = (u, i, j, r)
        lib().orc_mf_set_recorder(self._h, _ptr(u), _ptr(i), _ptr(j), _ptr(r), capacity)

    def recorded(self):
        n = lib().orc_mf_recorded(self._h)
        u, i, j, r = self._rec
        return u[:n].copy(), i[:n].copy(), j[:n].copy(), r[:n].copy()

    def epochIteration_Cython(self):
        if self.algorithm_name == "MF_BPR":
            lib().orc_mf_epoch_bpr(self._h)
        elif self.algorithm_name == "ASY_SVD":
            lib().orc_mf_epoch_asy(self._h)
        else:
            lib().orc_mf_epoch_funk(self._h)

    def replay(self, u, i, j=None, rating=None):
        u = np.ascontiguousarray(u, np.int32); i = np.ascontiguousarray(i, np.int32)
        if self.algorithm_name == "MF_BPR":
            j = np.ascontiguousarray(j, np.int32)
            lib().orc_mf_replay_bpr(self._h, _ptr(u), _ptr(i), _ptr(j), len(u))
        else:
            rating = np.ascontiguousarray(rating, np.float64)
            fn = lib().orc_mf_replay_asy if self.algorithm_name == "ASY_SVD" else lib().orc_mf_replay_funk
            fn(self._h, _ptr(u), _ptr(i), _ptr(rating), len(u))

    def _get(self):
        U = np.empty((self.n_u_rows, self.n_factors)); V = np.empty((self.n_items, self.n_factors))
        bu = np.empty(self.n_users); bi = np.empty(self.n_items); mu = C.c_double(0.0)
        lib().orc_mf_get(self._h, _ptr(U), _ptr(V), _ptr(bu), _ptr(bi), C.byref(mu))
        return U, V, bu, bi, mu.value

    def get_USER_factors(self):
        return self._get()[0]

    def get_ITEM_factors(self):
        return self._get()[1]

    def get_USER_bias(self):
        return self._get()[2]

    def get_ITEM_bias(self):
        return self._get()[3]

    def get_GLOBAL_bias(self):
        return np.array(self._get()[4])

    def cumulative_loss(self):
        return lib().orc_mf_loss(self._h)


# --------------------------------------------------------------------------------------------------
#                                            SLIM-BPR
# --------------------------------------------------------------------------------------------------

class OracleSLIM:
    """Restates SLIM_BPR_Cython_Epoch (dense, symmetric-triangular and sparse-tree stores; batch_size is 1 as the wrapper
    hard-codes, SLIM_BPR/Cython/SLIM_BPR_Cython.py:140).  The sparse-tree store (train_with_sparse_weights=True,
    Sparse_Matrix_Tree_CSR .pyx:582-1030) is restated as the dense array plus a node-exists map: per-row top-K selection at
    the rebalance points of the epoch (.pyx:320-324) and inside get_S (.pyx:381-382), which mutates the model."""

    def __init__(self, URM_mask, train_with_sparse_weights=False, final_model_sparse_weights=True,
                 learning_rate=0.01, li_reg=0.0, lj_reg=0.0, batch_size=1, topK=150, symmetric=True,
                 verbose=False, random_seed=None, sgd_mode="adam", gamma=0.995, beta_1=0.9, beta_2=0.999):
        if sgd_mode not in SGD_MODES:
            raise ValueError("Value for 'sgd_mode' not recognized: %r" % (sgd_mode,))
        URM = _sorted_csr(URM_mask)
        self.n_users, self.n_items = URM.shape
        self.topK = topK
        self.sparse = bool(train_with_sparse_weights)
        self.symmetric = bool(symmetric) and not self.sparse                 # .pyx:112-113
        self.final_model_sparse_weights = final_model_sparse_weights
        self._indptr = np.ascontiguousarray(URM.indptr, dtype=np.int32)
        self._indices = np.ascontiguousarray(URM.indices, dtype=np.int32)
        L = lib()
        if random_seed is not None:
            L.orc_srand(C.c_uint(int(random_seed)))
        self._h = L.orc_slim_create(self.n_users, self.n_items, _ptr(self._indptr), _ptr(self._indices),
                                    int(self.symmetric), SGD_MODES[sgd_mode], learning_rate, li_reg, lj_reg,
                                    gamma, beta_1, beta_2)
        if self.sparse:
            L.orc_slim_set_sparse(self._h, int(topK) if topK else 0)
        self._rec = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_slim_destroy(self._h)
            self._h = None

    def record_samples(self, capacity):
        u = np.zeros(capacity, np.int32); i = np.zeros(capacity, np.int32); j = np.zeros(capacity, np.int32)
        self._rec = (u, i, j)
        lib().orc_slim_set_recorder(self._h, _ptr(u), _ptr(i), _ptr(j), capacity)

    def recorded(self):
        n = lib().orc_slim_recorded(self._h)
        return tuple(a[:n].copy() for a in self._rec)

    def epochIteration_Cython(self):
        lib().orc_slim_epoch(self._h)

    def replay(self, u, i, j):
        u = np.ascontiguousarray(u, np.int32); i = np.ascontiguousarray(i, np.int32); j = np.ascontiguousarray(j, np.int32)
        lib().orc_slim_replay(self._h, _ptr(u), _ptr(i), _ptr(j), len(u))

    def get_S_dense(self):
        out = np.empty((self.n_items, self.n_items))
        lib().orc_slim_get_S(self._h, _ptr(out))
        return out

    def get_S(self):
        """get_S (.pyx:343-391): diagonal zeroed, then per-ROW top-K -> CSR (or the dense array)."""
        if self.sparse:                       # the tree's non-zero nodes in column order; the selection is kept in the model
            counts = np.zeros(self.n_items, np.int32)
            total = lib().orc_slim_sparse_get_S(self._h, _ptr(counts))
            cols = np.zeros(total, np.int32); data = np.zeros(total, np.float64)
            lib().orc_slim_sparse_cells(self._h, _ptr(cols), _ptr(data))
            indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
            return sps.csr_matrix((data, cols, indptr), shape=(self.n_items, self.n_items))
        S = self.get_S_dense()
        if not self.final_model_sparse_weights and not self.symmetric:
            return S
        return oracle_similarity_topk_rows(S, self.topK, zeros_compete=self.symmetric)


def oracle_similarity_topk_rows(S, topK, zeros_compete):
    """Per-row top-K of a dense square matrix, returned as CSR float64.

    zeros_compete=True  restates Triangular_Matrix.get_scipy_csr (SLIM_BPR_Cython_Epoch.pyx:1384-1404):
                        the K largest cells of the FULL row are taken (zeros included), exact zeros are then
                        dropped; order inside a row is descending value.
    zeros_compete=False restates similarityMatrixTopK(S.T, k).T (Base/Recommender_utils.py:55-122): only
                        non-zero cells compete (so negative cells survive while fewer than K non-zeros exist).
    Ties are broken towards the lower column index (the reference's are NumPy introselect artefacts).
    """
    n = S.shape[0]
    if not topK:
        return sps.csr_matrix(S)
    k = min(int(topK), n)
    rows, cols, vals = [], [], []
    for r in range(n):
        row = S[r]
        if zeros_compete:
            order = np.lexsort((np.arange(n), -row))[:k]
            order = order[row[order] != 0.0]
        else:
            nz = np.flatnonzero(row != 0.0)
            order = nz[np.lexsort((nz, -row[nz]))][:k]
        rows.extend([r] * len(order)); cols.extend(order.tolist()); vals.extend(row[order].tolist())
    return sps.csr_matrix((vals, (rows, cols)), shape=(n, n))


# --------------------------------------------------------------------------------------------------
#                                       Compute_Similarity
# --------------------------------------------------------------------------------------------------

_SIM_KIND = {"cosine": 0, "adjusted": 0, "pearson": 0, "asymmetric": 1, "jaccard": 2, "tanimoto": 2, "dice": 3,
             "tversky": 4}


class OracleSimilarity:
    """Restates Compute_Similarity_Cython (constructor .pyx:72-213, compute_similarity :411-607)."""

    def __init__(self, dataMatrix, topK=100, shrink=0, normalize=True, asymmetric_alpha=0.5,
                 tversky_alpha=1.0, tversky_beta=1.0, similarity="cosine", row_weights=None):
        if similarity not in _SIM_KIND:
            raise ValueError("Cosine_Similarity: value for parameter 'mode' not recognized: %r" % (similarity,))
        self.n_rows, self.n_columns = dataMatrix.shape
        self.shrink = int(shrink)                       # `cdef int shrink` truncates (.pyx:64)
        self.normalize = bool(normalize)
        self.kind = _SIM_KIND[similarity]
        # float32 round-trip of the three coefficients: they are `cdef float` in the reference (.pyx:65)
        self.asymmetric_alpha = float(np.float32(asymmetric_alpha))
        self.tversky_alpha = float(np.float32(tversky_alpha))
        self.tversky_beta = float(np.float32(tversky_beta))
        set_based = similarity in ("jaccard", "tanimoto", "dice", "tversky")
        if set_based:
            self.normalize = False
        self.TopK = min(int(topK), self.n_columns)
        # The reference pre-processes a COPY in the caller's dtype (float32 for a URM that went through
        # BaseRecommender.__init__) and only afterwards widens the stored values to float64 (.pyx:153-207).
        X = dataMatrix.copy()
        if not sps.issparse(X):
            X = sps.csr_matrix(X)
        if similarity == "adjusted":                    # subtract the row (user) mean of the stored cells
            X = sps.csr_matrix(X)
            cnt = np.diff(X.indptr)
            tot = np.array(X.sum(axis=1), dtype=np.float64).ravel()
            mean = np.divide(tot, cnt, out=np.zeros(self.n_rows), where=cnt > 0)
            X.data = X.data - np.repeat(mean, cnt).astype(X.data.dtype)     # see note below
        elif similarity == "pearson":                   # subtract the column (item) mean of the stored cells
            X = sps.csc_matrix(X)
            cnt = np.diff(X.indptr)
            tot = np.array(X.sum(axis=0), dtype=np.float64).ravel()
            mean = np.divide(tot, cnt, out=np.zeros(self.n_columns), where=cnt > 0)
            # `dataMatrix.data[i] -= colAverage` (.pyx:262,301) runs at Python level on a NumPy scalar: under
            # NumPy >= 2 (NEP 50) the Python-float mean is cast to the data dtype first, i.e. the subtraction
            # is carried out in float32 for a float32 matrix.  Restated as such (this is what oracle/_ref does
            # in this image).
            X.data = X.data - np.repeat(mean, cnt).astype(X.data.dtype)
        elif set_based:
            X.data = np.ones_like(X.data)
        sq = np.array(X.power(2).sum(axis=0), dtype=np.float64).ravel()
        self.sumOfSquared = sq if set_based else np.sqrt(sq)
        self.norm_alpha = self.norm_1ma = None
        if similarity == "asymmetric":
            self.norm_1ma = np.power(self.sumOfSquared, 2 * (1 - self.asymmetric_alpha))
            self.norm_alpha = np.power(self.sumOfSquared, 2 * self.asymmetric_alpha)
        self.row_weights = None
        if row_weights is not None:
            if len(row_weights) != self.n_rows:
                raise ValueError("Cosine_Similarity: provided row_weights and dataMatrix have different number of rows.")
            self.row_weights = np.ascontiguousarray(row_weights, dtype=np.float64)
        csr = sps.csr_matrix(X)
        csc = sps.csc_matrix(csr)
        self.csr = (np.ascontiguousarray(csr.indptr, np.int32), np.ascontiguousarray(csr.indices, np.int32),
                    np.ascontiguousarray(csr.data, np.float64))
        self.csc = (np.ascontiguousarray(csc.indptr, np.int32), np.ascontiguousarray(csc.indices, np.int32),
                    np.ascontiguousarray(csc.data, np.float64))

    def _range(self, start_col, end_col):
        s, e = 0, self.n_columns                       # .pyx:447-451
        if start_col is not None and 0 < start_col < self.n_columns:
            s = start_col
        if end_col is not None and s < end_col < self.n_columns:
            e = end_col
        return s, e

    def _common(self):
        return ([_ptr(a) for a in self.csr] + [_ptr(a) for a in self.csc] + [_ptr(self.row_weights)]
                + [self.kind, int(self.normalize), self.shrink]
                + [_ptr(self.sumOfSquared), _ptr(self.norm_alpha), _ptr(self.norm_1ma)]
                + [self.tversky_alpha, self.tversky_beta])

    def column(self, item):
        """Normalised dense column `item` (float64) and its touched ids in first-touch order."""
        w = np.zeros(self.n_columns); touched = np.zeros(self.n_columns, np.int32); mask = np.zeros(self.n_columns, np.int8)
        nt = lib().orc_sim_column(self.n_columns, int(item), *self._common(), _ptr(w), _ptr(touched), _ptr(mask))
        return w, touched[:nt]

    def compute_similarity(self, start_col=None, end_col=None, exact_numpy_topk=False):
        """Returns csr_matrix (n_columns, n_columns) float32 with column = source item (or a dense ndarray
        when topK == 0), like the reference.  exact_numpy_topk=True replays the reference's
        argpartition/argsort calls on the touched-order array (.pyx:523-545) column by column."""
        s, e = self._range(start_col, end_col)
        n = self.n_columns
        if self.TopK == 0:
            W = np.zeros((n, n))
            for c in range(s, e):
                W[:, c] = self.column(c)[0]
            return W
        if exact_numpy_topk:
            vals, rows, cols = [], [], []
            scratch = np.zeros(n)
            for c in range(s, e):
                w, touched = self.column(c)
                scratch[:] = 0.0
                scratch[:len(touched)] = -w[touched]
                k = min(self.TopK, len(touched))
                if k == 0:
                    continue
                part = np.argpartition(scratch, k - 1)[0:k]
                order = part[np.argsort(scratch[part])]
                for t in order:
                    if t < len(touched) and w[touched[t]] != 0.0:
                        vals.append(w[touched[t]]); rows.append(touched[t]); cols.append(c)
            return sps.csr_matrix((vals, (rows, cols)), shape=(n, n), dtype=np.float32)
        idx, val = self.build_slabs(s, e)
        return slabs_to_csr(idx, val, s, n)

    def compute_similarity_full_column_rule(self, start_col=None, end_col=None):
        """The top-K rule of the reference's SECOND implementation, Compute_Similarity_Python.compute_similarity
        (Base/Similarity/Compute_Similarity_Python.py:346-355): `(-column).argpartition(TopK-1)[0:TopK]` over the FULL column
        -- zeros (untouched cells, the zeroed diagonal) compete with negative similarities -- then exact zeros are dropped.
        Ties (NumPy introselect artefacts in the reference) go to the lower index here."""
        s, e = self._range(start_col, end_col)
        n = self.n_columns
        vals, rows, cols = [], [], []
        k = min(self.TopK, n)
        for c in range(s, e):
            w = self.column(c)[0]
            order = np.lexsort((np.arange(n), -w))[:k]
            order = order[w[order] != 0.0]
            vals.extend(w[order]); rows.extend(order); cols.extend([c] * len(order))
        return sps.csr_matrix((vals, (rows, cols)), shape=(n, n), dtype=np.float32)

    def build_slabs(self, start_col, end_col):
        ncol = end_col - start_col
        idx = np.empty((ncol, self.TopK), np.int32); val = np.empty((ncol, self.TopK), np.float32)
        lib().orc_sim_build(self.n_columns, start_col, end_col, self.TopK, *self._common(), _ptr(idx), _ptr(val))
        return idx, val


class OracleSimilarityEuclidean:
    """Restates Compute_Similarity_Euclidean (Base/Similarity/Compute_Similarity_Euclidean.py:13; compute_similarity
    :87-248) in NumPy, in the dtype of the input like the reference (float32 for a URM): squared distances from the
    Gram matrix (:167-172), optional division by the product of the norms (:178-179) and by n_rows (:181-182), square
    root (:184), the three distance -> similarity maps (:186-196), zero diagonal (:202), top-K of the whole column with
    zeros dropped (:213-224).  row_weights (:62-72): the Gram matrix comes from the row-weighted copy of the data (:153) and the
    distance vector over the COLUMNS is multiplied by the weights of the ROWS (:174-175) -- NumPy only allows that on square inputs
    and the restatement inherits the error for every other shape.  `dense()` returns every column; `compute_similarity()` the
    csr_matrix the reference returns."""

    MODES = ("lin", "log", "exp")

    def __init__(self, dataMatrix, topK=100, shrink=0, normalize=False, normalize_avg_row=False,
                 similarity_from_distance_mode="lin", row_weights=None):
        if similarity_from_distance_mode not in self.MODES:
            raise ValueError("Compute_Similarity_Euclidean: value for parameter 'mode' not recognized.")
        self.row_weights = None
        if row_weights is not None:
            if dataMatrix.shape[0] != len(row_weights):
                raise ValueError("Compute_Similarity_Euclidean: provided row_weights and dataMatrix have different number of rows.")
            self.row_weights = row_weights.copy()                                              # (:70)
            self.X_weighted = dataMatrix.T.dot(sps.diags(self.row_weights)).T                   # (:71-73)
        self.X = dataMatrix.copy()      # the caller's sparse format: float32 summation order depends on it (:44)
        self.n_rows, self.n_columns = self.X.shape
        self.TopK = min(int(topK), self.n_columns)
        self.shrink, self.normalize, self.normalize_avg_row = shrink, bool(normalize), bool(normalize_avg_row)
        self.mode = similarity_from_distance_mode
        self.sq = np.asarray(self.X.power(2).sum(axis=0)).ravel()          # item_distance_initial (:112)
        self.rt = np.sqrt(self.sq)                                         # sumOfSquared (:113)

    def columns(self, start, end):
        """Similarity columns [start, end) as an (n_columns, end - start) array of the input dtype."""
        block = self.X[:, start:end].toarray()
        gram = np.asarray((self.X if self.row_weights is None else self.X_weighted).T.dot(block))      # (:153 / :156)
        out = np.empty_like(gram)
        for k, c in enumerate(range(start, end)):
            d2 = self.sq.copy()
            d2 += self.sq[c]
            d2 -= 2 * gram[:, k]
            d2[c] = 0.0
            if self.row_weights is not None:
                d2 = np.multiply(d2, self.row_weights)                     # (:174-175; float64 weights promote the vector)
            if self.normalize:
                d2 /= self.rt[c] * self.rt
            if self.normalize_avg_row:
                d2 /= self.n_rows
            with np.errstate(invalid="ignore"):
                d = np.sqrt(d2)
            if self.mode == "exp":
                sim = 1 / (np.exp(d) + self.shrink + 1e-9)
            elif self.mode == "lin":
                sim = 1 / (d + self.shrink + 1e-9)
            else:
                sim = 1 / (np.log(d + 1) + self.shrink + 1e-9)
            sim[c] = 0.0
            out[:, k] = sim
        return out

    def dense(self):
        return self.columns(0, self.n_columns)

    def compute_similarity(self):
        W = self.dense()
        rows, cols, vals = [], [], []
        for c in range(self.n_columns):
            w = W[:, c]
            part = (-w).argpartition(self.TopK - 1)[0:self.TopK]
            top = part[np.argsort(-w[part])]
            top = top[w[top] != 0.0]
            rows.extend(top); cols.extend([c] * len(top)); vals.extend(w[top])
        return sps.csr_matrix((vals, (rows, cols)), shape=(self.n_columns, self.n_columns), dtype=np.float32)


def slabs_to_csr(idx, val, start_col, n_columns):
    """(n_local, topK) neighbour/value slabs (-1 padded) -> csr_matrix with column = source item (.pyx:603-605)."""
    keep = idx >= 0
    cols = np.broadcast_to(np.arange(start_col, start_col + idx.shape[0], dtype=np.int32)[:, None], idx.shape)[keep]
    return sps.csr_matrix((val[keep], (idx[keep], cols)), shape=(n_columns, n_columns), dtype=np.float32)


# --------------------------------------------------------------------------------------------------
#                                              IALS
# --------------------------------------------------------------------------------------------------

def oracle_ials_confidence(URM, confidence_scaling="linear", alpha=1.0, epsilon=1.0):
    """_linear_scaling_confidence / _log_scaling_confidence (IALSRecommender.py:111-123); float32 like the reference."""
    Cm = sps.csr_matrix(URM, dtype=np.float32, copy=True)
    if confidence_scaling == "linear":
        Cm.data = 1.0 + alpha * Cm.data
    else:
        Cm.data = 1.0 + alpha * np.log(1.0 + Cm.data / epsilon)
    Cm.data = Cm.data.astype(np.float32)
    return Cm


def _ials_update_row(profile, confidence, Y, YtY, reg_diag):
    """_update_row (IALSRecommender.py:170-201): x = inv(YtY + Y_I^T (C-1) Y_I + reg I) . (Y_I^T c)."""
    Yi = Y[profile, :]
    A = Yi.T.dot(((confidence - 1) * Yi.T).T)
    B = YtY + A + reg_diag
    return np.dot(np.linalg.inv(B), Yi.T.dot(confidence))


def oracle_ials_epoch(C_csr, C_csc, U, V, reg):
    """_run_epoch (IALSRecommender.py:137-166): user pass against V (Jacobi inside the pass), then item pass
    against the UPDATED U.  U and V are float64 arrays updated in place; only warm rows are touched."""
    k = V.shape[1]
    reg_diag = np.diag(reg * np.ones(k))
    try:        # thousands of small k x k products and inverses: one BLAS thread (a 256-core host spends its time handing them out)
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except ImportError:
        limit = None
    try:
        VV = V.T.dot(V)
        for u in np.flatnonzero(np.diff(C_csr.indptr) > 0):
            s, e = C_csr.indptr[u], C_csr.indptr[u + 1]
            U[u, :] = _ials_update_row(C_csr.indices[s:e], C_csr.data[s:e], V, VV, reg_diag)
        UU = U.T.dot(U)
        for i in np.flatnonzero(np.diff(C_csc.indptr) > 0):
            s, e = C_csc.indptr[i], C_csc.indptr[i + 1]
            V[i, :] = _ials_update_row(C_csc.indices[s:e], C_csc.data[s:e], U, UU, reg_diag)
    finally:
        if limit is not None:
            limit.restore_original_limits()
    return U, V
