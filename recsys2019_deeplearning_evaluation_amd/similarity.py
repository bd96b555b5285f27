"""Compute_Similarity on MI355X: host front-end of the sim_* entry points of libmi355rec.so.

Mirrors
  Compute_Similarity_Cython   Base/Similarity/Cython/Compute_Similarity_Cython.pyx:51 (ctor :72, compute_similarity :411)
  Compute_Similarity          Base/Similarity/Compute_Similarity.py:32  (the dispatcher every KNN recommender calls)
with the same constructor arguments, the same `compute_similarity(start_col=None, end_col=None)` signature,
the same return types (csr_matrix (n_cols, n_cols) float32 with column = source item; dense ndarray when
topK == 0) and the same ValueError on bad enum arguments.  All arithmetic happens on the device; there is
no CPU fallback.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sps

from . import _native as N
from .recommender_base import check_matrix


def slabs_to_csr(nbr_idx, nbr_val, start_col, n_columns):
    """(n_local, topK) neighbour / value slabs (-1 padded, value-descending) -> csr_matrix whose column
    start_col + r holds row r of the slabs: the COO assembly of Compute_Similarity_Cython.pyx:603-605."""
    keep = nbr_idx >= 0
    per_col = keep.sum(axis=1)
    indptr = np.zeros(n_columns + 1, dtype=np.int64)
    indptr[start_col + 1:start_col + 1 + len(per_col)] = per_col
    np.cumsum(indptr, out=indptr)
    W = sps.csc_matrix((nbr_val[keep], nbr_idx[keep], indptr), shape=(n_columns, n_columns), dtype=np.float32)
    return W.tocsr()


class Compute_Similarity_MI355X:
    """Drop-in for Compute_Similarity_Cython backed by the gfx950 kernels."""

    SIMILARITY_VALUES = ("cosine", "pearson", "adjusted", "asymmetric", "jaccard", "tanimoto", "dice", "tversky")

    def __init__(self, dataMatrix, topK=100, shrink=0, normalize=True, asymmetric_alpha=0.5, tversky_alpha=1.0,
                 tversky_beta=1.0, similarity="cosine", row_weights=None, unit_column_side=False,
                 normalize_avg_row=False, similarity_from_distance_mode="lin", feature_weighting="none",
                 weighting_documents="columns", K1=1.2, B=0.75, resident=None):
        """feature_weighting ("none" / "BM25" / "TF-IDF"), weighting_documents ("columns" / "rows"), K1, B: the re-weighting the
        KNN recommenders apply to the matrix before the build (Base/IR_feature_weighting.py), as a device pre-pass on the
        uploaded values; `weighted_matrix()` hands the re-weighted matrix back.  resident: a `_native.ResidentURM` holding THIS
        dataMatrix on the device (checked): the constructor copies it there instead of uploading it.  Not arguments of the
        reference class."""
        if similarity not in self.SIMILARITY_VALUES and similarity != "euclidean":
            raise ValueError("Cosine_Similarity: value for parameter 'mode' not recognized."
                             " Allowed values are: 'cosine', 'pearson', 'adjusted', 'asymmetric', 'jaccard', 'tanimoto',"
                             "dice, tversky. Passed value was '{}'".format(similarity))
        if feature_weighting not in N.FEATURE_WEIGHTING_CODES:
            raise ValueError("Value for 'feature_weighting' not recognized. Acceptable values are {}, provided was '{}'".format(
                list(N.FEATURE_WEIGHTING_CODES), feature_weighting))
        if feature_weighting == "BM25":
            assert 0 < B < 1, "okapi_BM_25: B must be in (0,1)"
            assert K1 > 0, "okapi_BM_25: K1 must be > 0"
        if feature_weighting == "TF-IDF":
            assert np.all(dataMatrix.data >= 0.0), \
                "TF_IDF: Data matrix contains {} negative values, computing the square root is not possible.".format(np.sum(dataMatrix.data < 0.0))
        self.n_rows, self.n_columns = dataMatrix.shape
        self.TopK = min(int(topK), self.n_columns)
        self.similarity = similarity
        if row_weights is not None and self.n_rows != len(row_weights):
            raise ValueError("Cosine_Similarity: provided row_weights and dataMatrix have different number of rows."
                             "Row_weights has {} rows, dataMatrix has {}.".format(len(row_weights), self.n_rows))
        # the reference sums the squares behind the norms in float32, in an order that follows the sparse format it is handed
        # (include/mi355rec.h, norm_sum_order): CSC input and pearson (whose pre-pass converts to CSC) -> NumPy's pairwise reduceat
        # order, everything else (CSR, the adjusted-cosine pre-pass, ndarray / COO input) -> one square after the other in row order
        norm_sum_order = 1 if (similarity == "pearson" or (sps.isspmatrix_csc(dataMatrix) and similarity != "adjusted")) else 0
        csr = check_matrix(dataMatrix, "csr", dtype=np.float32)
        if not csr.has_sorted_indices:
            csr = csr.sorted_indices()
        indptr, indices, data = N.as_i32(csr.indptr), N.as_i32(csr.indices), N.as_f32(csr.data)
        rw = None if row_weights is None else N.as_f32(row_weights)
        cfg = N.SimConfig(self.TopK, int(shrink), int(bool(normalize)), N.SIMILARITY_CODES[similarity],
                          float(asymmetric_alpha), float(tversky_alpha), float(tversky_beta), int(bool(unit_column_side)),
                          int(bool(normalize_avg_row)), N.EUCLIDEAN_MODE_CODES.get(similarity_from_distance_mode, -1),
                          N.FEATURE_WEIGHTING_CODES[feature_weighting], int(weighting_documents == "rows"), float(K1), float(B),
                          norm_sum_order, 0)
        assert weighting_documents in ("columns", "rows")
        self._weighted_structure = (csr.indptr, csr.indices, csr.shape) if feature_weighting != "none" else None
        self._lib = N.load()
        self._h = C.c_void_p()
        if resident is not None:
            if not resident.matches(csr):
                raise ValueError("Compute_Similarity_MI355X: `resident` does not hold this dataMatrix (shape, nnz or contents differ)")
            N.check(self._lib.mi355rec_sim_create_resident(C.byref(self._h), C.byref(cfg), self.n_rows, self.n_columns,
                                                           resident.indptr.ptr, resident.indices.ptr, resident.data.ptr, N.ptr(rw)))
        else:
            N.check(self._lib.mi355rec_sim_create(C.byref(self._h), C.byref(cfg), self.n_rows, self.n_columns,
                                                  N.ptr(indptr), N.ptr(indices), N.ptr(data), N.ptr(rw)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355rec_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def weighted_matrix(self):
        """The BM25 / TF-IDF re-weighted dataMatrix (csr, float32), computed on the device by the constructor."""
        if self._weighted_structure is None:
            raise ValueError("Compute_Similarity_MI355X was created with feature_weighting='none'")
        indptr, indices, shape = self._weighted_structure
        data = np.empty(len(indices), np.float32)
        N.check(self._lib.mi355rec_sim_get_weighted_values(self._h, N.ptr(data)))
        return sps.csr_matrix((data, indices.copy(), indptr.copy()), shape=shape)

    def _range(self, start_col, end_col):
        # same acceptance rule as Compute_Similarity_Cython.pyx:447-451
        s, e = 0, self.n_columns
        if start_col is not None and 0 < start_col < self.n_columns:
            s = int(start_col)
        if end_col is not None and s < end_col < self.n_columns:
            e = int(end_col)
        return s, e

    def compute_slabs(self, start_col=None, end_col=None):
        """Raw device output for columns [start_col, end_col): (idx int32 (n_local, topK), val float32)."""
        s, e = self._range(start_col, end_col)
        idx = np.empty((e - s, self.TopK), dtype=np.int32)
        val = np.empty((e - s, self.TopK), dtype=np.float32)
        N.check(self._lib.mi355rec_sim_compute(self._h, s, e, N.ptr(idx), N.ptr(val)))
        return idx, val, s

    def compute_slabs_device(self, start_col, end_col, d_idx_ptr, d_val_ptr):
        """Asynchronous variant writing into device buffers (data_ptr() of int32 / float32 tensors)."""
        s, e = self._range(start_col, end_col)
        N.check(self._lib.mi355rec_sim_compute_device(self._h, s, e, C.c_void_p(d_idx_ptr), C.c_void_p(d_val_ptr)))
        return s, e

    def compute_part_device(self, part, n_parts, d_idx_ptr, d_val_ptr):
        """Asynchronous build of interleaved part `part` of `n_parts` (equal column counts and equal cost per part) into device
        slabs of ceil(n_columns / n_parts) rows; row q holds column part_columns(part, n_parts)[q]."""
        N.check(self._lib.mi355rec_sim_compute_part_device(self._h, int(part), int(n_parts), C.c_void_p(d_idx_ptr), C.c_void_p(d_val_ptr)))

    def compute_part_chunk_device(self, part, n_parts, slot_first, slot_count, d_idx_ptr, d_val_ptr):
        """Rows [slot_first, slot_first + slot_count) of interleaved part `part` only, into rows 0 .. slot_count - 1 of the slabs."""
        N.check(self._lib.mi355rec_sim_compute_part_chunk_device(self._h, int(part), int(n_parts), int(slot_first), int(slot_count),
                                                                 C.c_void_p(d_idx_ptr), C.c_void_p(d_val_ptr)))

    def pack_slab_device(self, d_idx_ptr, d_val_ptr, n_cells, d_packed_ptr):
        """(idx, val) device slabs of n_cells cells -> the 6-byte exchange cells (values, then 16-bit ids); asynchronous on the handle's stream."""
        N.check(self._lib.mi355rec_sim_pack_slab_device(self._h, C.c_void_p(d_idx_ptr), C.c_void_p(d_val_ptr), int(n_cells), C.c_void_p(d_packed_ptr)))

    def unpack_slab_device(self, d_packed_ptr, n_cells, d_idx_ptr, d_val_ptr):
        N.check(self._lib.mi355rec_sim_unpack_slab_device(self._h, C.c_void_p(d_packed_ptr), int(n_cells), C.c_void_p(d_idx_ptr), C.c_void_p(d_val_ptr)))

    def part_columns(self, part, n_parts):
        n = C.c_int32()
        N.check(self._lib.mi355rec_sim_part_columns(self._h, int(part), int(n_parts), None, C.byref(n)))
        cols = np.empty(n.value, np.int32)
        N.check(self._lib.mi355rec_sim_part_columns(self._h, int(part), int(n_parts), N.ptr(cols), C.byref(n)))
        return cols

    def synchronize(self):
        N.check(self._lib.mi355rec_sim_sync(self._h))

    def column_costs(self):
        cost = np.empty(self.n_columns, dtype=np.int64)
        N.check(self._lib.mi355rec_sim_column_costs(self._h, N.ptr(cost)))
        return cost

    def schedule_info(self):
        """(work items, split columns, parts) of the last compute call."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        N.check(self._lib.mi355rec_sim_schedule_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def accumulator_info(self):
        """("uint32" | "int64-fixed" | "float64" | "int32-exact", scale): type of the in-LDS column accumulator (diagnostics)."""
        kind, scale = C.c_int32(), C.c_double()
        N.check(self._lib.mi355rec_sim_accumulator_info(self._h, C.byref(kind), C.byref(scale)))
        return ("uint32", "int64-fixed", "float64", "int32-exact")[kind.value], scale.value

    def selection_info(self):
        """(columns selected threshold-first, their candidates in total, fall-backs to the full selection) of the last compute call."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        N.check(self._lib.mi355rec_sim_selection_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def stats(self):
        st = N.Stats()
        N.check(self._lib.mi355rec_sim_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def compute_similarity(self, start_col=None, end_col=None):
        if self.TopK == 0:
            s, e = self._range(start_col, end_col)
            # the device writes columns [s, e) straight into W (row pitch n_columns): no second n x n array on the host
            W = np.zeros((self.n_columns, self.n_columns), dtype=np.float32) if (s, e) != (0, self.n_columns) else \
                np.empty((self.n_columns, self.n_columns), dtype=np.float32)
            N.check(self._lib.mi355rec_sim_compute_dense(self._h, s, e, C.c_void_p(W.ctypes.data + 4 * s), self.n_columns))
            return W
        # CSR assembled on the device: SciPy's COO/CSC -> CSR conversion of the result would cost more than the build
        s, e = self._range(start_col, end_col)
        cap = (e - s) * self.TopK
        indptr = np.empty(self.n_columns + 1, dtype=np.int32)
        indices = np.empty(cap, dtype=np.int32)
        data = np.empty(cap, dtype=np.float32)
        nnz = C.c_int64()
        N.check(self._lib.mi355rec_sim_compute_csr(self._h, s, e, N.ptr(indptr), N.ptr(indices), N.ptr(data), C.byref(nnz)))
        W = sps.csr_matrix((data[:nnz.value], indices[:nnz.value], indptr), shape=(self.n_columns, self.n_columns))
        W.has_sorted_indices = True
        return W


class Compute_Similarity_Euclidean_MI355X(Compute_Similarity_MI355X):
    """Drop-in for Compute_Similarity_Euclidean (Base/Similarity/Compute_Similarity_Euclidean.py:13): same constructor
    keywords and defaults (normalize=False!), same `compute_similarity(start_col, end_col)` and csr_matrix float32
    result.  similarity = 1 / (f(distance) + shrink + 1e-9) with f = identity / log(1 + .) / exp for "lin" / "log" / "exp",
    every pair of columns has one (no co-occurrence needed), the diagonal is 0.  `row_weights` (:62-72) weight the dot
    products and multiply the distances to the n_cols columns by the n_rows weights (:174-175), as in the reference; that
    product only exists for square inputs -- any other shape raises ValueError here, in the constructor, where the reference
    fails in NumPy's broadcasting inside compute_similarity."""

    def __init__(self, dataMatrix, topK=100, shrink=0, normalize=False, normalize_avg_row=False,
                 similarity_from_distance_mode="lin", row_weights=None, feature_weighting="none", weighting_documents="columns",
                 K1=1.2, B=0.75, **args):
        if similarity_from_distance_mode not in N.EUCLIDEAN_MODE_CODES:
            raise ValueError("Compute_Similarity_Euclidean: value for parameter 'mode' not recognized."
                             " Allowed values are: 'exp', 'lin', 'log'."
                             " Passed value was '{}'".format(similarity_from_distance_mode))
        if row_weights is not None and dataMatrix.shape[0] != len(row_weights):
            raise ValueError("Compute_Similarity_Euclidean: provided row_weights and dataMatrix have different number of rows."
                             "row_weights has {} rows, dataMatrix has {}.".format(len(row_weights), dataMatrix.shape[0]))
        super().__init__(dataMatrix, topK=topK, shrink=shrink, normalize=normalize, similarity="euclidean",
                         row_weights=row_weights, normalize_avg_row=normalize_avg_row,
                         similarity_from_distance_mode=similarity_from_distance_mode,
                         # the KNN recommenders' BM25 / TF-IDF pre-pass runs before the Euclidean set-up like before any other
                         # build (run_parameter_search.py:219-239 searches feature_weighting for euclidean, too)
                         feature_weighting=feature_weighting, weighting_documents=weighting_documents, K1=K1, B=B)


class Compute_Similarity:
    """Dispatcher with the reference's signature (Compute_Similarity.py:32).  `use_implementation` accepts
    "mi355x" (also chosen by the reference's default "density" rule and by "cython"): on this path every
    implementation name resolves to the device kernels, similarity="euclidean" goes to the Euclidean front-end as in
    the reference (:59-62); "python" raises NotImplementedError instead of silently running on the CPU."""

    def __init__(self, dataMatrix, use_implementation="density", similarity=None, **args):
        # (the reference asserts np.isfinite over the whole data array here, Compute_Similarity.py:34-36: two host passes over 80 MB at
        # ML-20M shape, half of an ItemKNN fit on this path.  The library's constructor finds non-finite values in the pass over the
        # uploaded values it makes anyway and refuses; the count for the reference's message is then taken on the host.)
        def non_finite():
            return AssertionError("Compute_Similarity: Data matrix contains {} non finite values".format(
                np.sum(np.logical_not(np.isfinite(dataMatrix.data)))))
        if similarity == "euclidean" and not np.all(np.isfinite(dataMatrix.data)):        # (that front-end squares the values first)
            raise non_finite()
        assert similarity == "euclidean" or not (dataMatrix.shape[0] == 1 and dataMatrix.nnz == dataMatrix.shape[1]), \
            "Compute_Similarity: data has only 1 feature (shape: {}) with dense values," \
            " vector and set based similarities are not defined on 1-dimensional dense data," \
            " use Euclidean similarity instead.".format(dataMatrix.shape)
        if similarity is not None:
            args["similarity"] = similarity
        if use_implementation not in ("density", "cython", "mi355x"):
            if use_implementation == "python":
                raise NotImplementedError("Compute_Similarity: the NumPy implementation is not provided here")
            raise ValueError("Compute_Similarity: value for argument 'use_implementation' not recognized")
        if isinstance(dataMatrix, np.ndarray):
            dataMatrix = sps.csr_matrix(dataMatrix)
        if similarity == "euclidean":
            args.pop("similarity", None)
            args.pop("resident", None)              # (the Euclidean front-end uploads its own, squared, copy)
            self.compute_similarity_object = Compute_Similarity_Euclidean_MI355X(dataMatrix, **args)
        else:
            try:
                self.compute_similarity_object = Compute_Similarity_MI355X(dataMatrix, **args)
            except ValueError as exc:
                if "non finite" in str(exc):
                    raise non_finite() from None
                raise

    def compute_similarity(self, **args):
        return self.compute_similarity_object.compute_similarity(**args)
