"""EASE_R with its Gram step on MI355X.

The reference (EASE_R/EASE_R_Recommender.py:55-65) asks `Compute_Similarity(URM, shrink=0, topK=n_items, normalize=False,
similarity="cosine")` for X^T X and densifies it.  Here the same product comes straight from the similarity kernel's dense
(topK = 0) path.  What follows is the closed form of Steck's model, B = I - P diag(1 / diag P) with P = (X^T X + l2 I)^-1,
evaluated like the reference does it: one float32 `np.linalg.inv` on the host (a third-party LAPACK solve, outside the
hot path) and the division of every column by its own diagonal entry.
"""
import numpy as np
import scipy.sparse as sps

from .recommender_base import BaseItemSimilarityMatrixRecommender, similarityMatrixTopK
from .similarity import Compute_Similarity_MI355X


def _unit_l2(X, axis):
    """Scale the rows (axis=1) or columns (axis=0) of a sparse matrix to unit Euclidean length; empty ones stay empty."""
    M = (sps.csr_matrix if axis == 1 else sps.csc_matrix)(X, dtype=np.float32)
    length = np.sqrt(np.asarray(M.multiply(M).sum(axis=axis), dtype=np.float64)).ravel()
    length[length == 0.0] = 1.0
    M.data = (M.data / np.repeat(length, np.diff(M.indptr))).astype(np.float32)
    return M


class EASE_R_Recommender(BaseItemSimilarityMatrixRecommender):
    """Same surface as the reference class (EASE_R_Recommender.py:20): `fit(topK=None, l2_norm=1e3,
    normalize_matrix=False)`; `W_sparse` is the dense n_items x n_items weight matrix when topK is None, otherwise its
    column-wise top-K (`similarityMatrixTopK`) as a csr_matrix."""

    RECOMMENDER_NAME = "EASE_R_Recommender"

    def __init__(self, URM_train, verbose=True):
        super(EASE_R_Recommender, self).__init__(URM_train, verbose=verbose)

    def _gram_matrix(self):
        """X^T X with a zero diagonal, float32, from the device."""
        builder = Compute_Similarity_MI355X(self.URM_train, topK=0, shrink=0, normalize=False, similarity="cosine")
        try:
            gram = builder.compute_similarity()
            self.similarity_stats = builder.stats()
        finally:
            builder.close()
        return gram

    def fit(self, topK=None, l2_norm=1e3, normalize_matrix=False, verbose=True):
        self.verbose = verbose
        if normalize_matrix:                    # unit rows first, then unit columns of the result (:47-51)
            self.URM_train = sps.csr_matrix(_unit_l2(_unit_l2(self.URM_train, axis=1), axis=0))
        gram = self._gram_matrix()
        n_items = gram.shape[0]
        on_diagonal = slice(None, None, n_items + 1)             # stride of the diagonal in the flattened matrix
        # the diagonal of X^T X is taken as the number of stored cells of each item (:63), plus the ridge term
        gram.flat[on_diagonal] = np.diff(self.URM_train.tocsc().indptr) + l2_norm
        precision = np.linalg.inv(gram)
        weights = precision / -precision.diagonal()              # column j over -P[j, j]
        weights.flat[on_diagonal] = 0.0
        if topK is None:
            self.W_sparse = weights
        else:
            self.W_sparse = sps.csr_matrix(similarityMatrixTopK(weights, k=topK, verbose=False))

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        """profiles . W for a dense or a sparse W; items outside `items_to_compute` get -inf."""
        scores = self.URM_train[user_id_array] @ self.W_sparse
        scores = scores.toarray() if sps.issparse(scores) else np.asarray(scores)
        if items_to_compute is None:
            return scores
        masked = np.full(scores.shape, -np.inf, dtype=np.float32)
        masked[:, items_to_compute] = scores[:, items_to_compute]
        return masked
