"""Shared helpers of the test-suite (comparators, fixture loading)."""
import json
import os

import numpy as np
import scipy.sparse as sps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cases = json.loads(str(z["cases"]))
    return z, cases


def unpack_csr(z, prefix):
    shape = tuple(int(x) for x in z[prefix + "_shape"])
    return sps.csr_matrix((z[prefix + "_data"], z[prefix + "_indices"], z[prefix + "_indptr"]), shape=shape)


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the '1e-5 relative on float32 factor matrices / similarity values' of north_star."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() / scale


def elementwise_close(a, b, rtol, atol_frac=1e-6):
    """|a-b| <= rtol*|b| + atol_frac*max|b| everywhere."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    tol = rtol * np.abs(b) + atol_frac * np.abs(b).max()
    bad = np.abs(a - b) > tol
    return not bad.any(), (np.abs(a - b) - tol).max()


def check_topk_against_dense(idx, val, dense_col, topK, rtol=1e-5):
    """Tie-aware top-K check of ONE column.

    idx/val : device (or oracle) output, -1 padded, value-descending
    dense_col: the full normalised column (float64) from the oracle
    Rule (SURVEY section 7 hard part 4): with t = K-th largest value of the column (zeros compete, then are dropped),
    every cell strictly above t (beyond tolerance) must be present, nothing below t (beyond tolerance) may be
    present, the emitted values must match the oracle's values for the emitted indices, order must be
    non-increasing, and no zero may be emitted.
    """
    n = len(dense_col)
    K = min(topK, n)
    got = idx[idx >= 0]
    got_val = val[:len(got)]
    assert (idx[len(got):] == -1).all(), "padding must be trailing"
    assert len(np.unique(got)) == len(got), "duplicate neighbour"
    scale = max(np.abs(dense_col).max(), 1e-30)
    tol = rtol * scale
    order = np.sort(dense_col)[::-1]
    t = order[K - 1]
    must = np.flatnonzero((dense_col > t + tol) & (dense_col != 0))
    assert np.isin(must, got).all(), "a cell strictly above the K-th value is missing"
    allowed = dense_col[got]
    assert (allowed >= t - tol).all(), "a cell strictly below the K-th value was emitted"
    assert (allowed != 0).all(), "a zero similarity was emitted"
    expected_count = min(K, int(((dense_col >= t - tol) & (dense_col != 0)).sum()))
    strict_count = int(((dense_col > t + tol) & (dense_col != 0)).sum())
    assert strict_count <= len(got) <= max(expected_count, strict_count)
    np.testing.assert_allclose(got_val, allowed, rtol=rtol, atol=tol)
    assert (np.diff(got_val) <= tol).all(), "values must be non-increasing"


def csr_columns_as_slabs(W, topK):
    """csr/csc matrix with <= topK entries per column -> (idx, val) slabs sorted by descending value."""
    W = sps.csc_matrix(W)
    n = W.shape[1]
    idx = -np.ones((n, topK), np.int32); val = np.zeros((n, topK), np.float32)
    for c in range(n):
        s, e = W.indptr[c], W.indptr[c + 1]
        o = np.argsort(-W.data[s:e], kind="stable")
        idx[c, :e - s] = W.indices[s:e][o]; val[c, :e - s] = W.data[s:e][o]
    return idx, val
