"""TEST INFRASTRUCTURE ONLY (oracle side) -- import shim for the compiled reference modules.

The reference's Cython modules do `from Base.Recommender_utils import check_matrix[, similarityMatrixTopK]`
at import time (MatrixFactorization_Cython_Epoch.pyx:18, SLIM_BPR_Cython_Epoch.pyx:34,
Compute_Similarity_Cython.pyx:43).  On the GPU box /root/reference does not exist, so
`oracle/ref_loader.py` puts this directory on sys.path *after* /root/reference: when the reference tree is
present its own module wins, otherwise these two independent re-implementations of the documented
behaviour (Base/Recommender_utils.py:13 `check_matrix`, :55 `similarityMatrixTopK`) are used.
"""
import numpy as np
import scipy.sparse as sps

_CONVERT = {"csc": (sps.csc_matrix, "tocsc"), "csr": (sps.csr_matrix, "tocsr"), "coo": (sps.coo_matrix, "tocoo")}


def check_matrix(X, format="csc", dtype=np.float32):
    """Return X in the requested sparse format; dtype is applied whenever a conversion happens
    (a matrix that already has the format is returned as-is, as in the reference)."""
    if isinstance(X, np.ndarray) and format != "npy":
        X = sps.csr_matrix(X, dtype=dtype)
        X.eliminate_zeros()
    if format == "npy":
        return X.toarray().astype(dtype) if sps.issparse(X) else np.array(X)
    cls, meth = _CONVERT[format]
    if isinstance(X, cls):
        return X
    return getattr(X, meth)().astype(dtype)


def similarityMatrixTopK(item_weights, k=100, verbose=False):
    """Column-wise top-k of a square matrix; exact zeros are dropped; returns CSC float32."""
    n = item_weights.shape[1]
    assert item_weights.shape[0] == n
    k = min(k, n)
    dense = isinstance(item_weights, np.ndarray)
    if not dense:
        item_weights = check_matrix(item_weights, "csc", dtype=np.float32)
    data, rows, indptr = [], [], [0]
    all_rows = np.arange(n, dtype=np.int32)
    for c in range(n):
        if dense:
            col, ridx = item_weights[:, c], all_rows
        else:
            s, e = item_weights.indptr[c], item_weights.indptr[c + 1]
            col, ridx = item_weights.data[s:e], item_weights.indices[s:e]
        nz = col != 0
        col, ridx = col[nz], ridx[nz]
        top = np.argsort(col)[-k:]
        data.extend(col[top]); rows.extend(ridx[top]); indptr.append(len(data))
    return sps.csc_matrix((data, rows, indptr), shape=(n, n), dtype=np.float32)
