"""P3alpha / RP3beta on MI355X: the random-walk similarity
    W[i, j] = sum_u Piu[i, u] * Pui[u, j] (* degree[j]^-beta),   row-wise top-K
of GraphBased/P3alphaRecommender.py:31-131 and GraphBased/RP3betaRecommender.py:30-140 has the same shape as the
similarity build (for every item, walk its users, add their profiles, keep the K best): it runs on `sim_column_kernel`.
Piu is the column-normalised BOOLEAN URM transposed, so Piu[i, u] = n_i^-alpha does not depend on u: the device sums
M[u, j] = Pui[u, j]^alpha * degree[j] over the users of item i with a unit column side, and the host applies the
per-row factor n_i^-alpha, the optional l1 row normalisation and the reference's final column-wise
similarityMatrixTopK (all O(n_items * topK) post-steps, as in the reference).
"""
import numpy as np
import scipy.sparse as sps

from .recommender_base import BaseItemSimilarityMatrixRecommender, check_matrix, similarityMatrixTopK
from .scoring import GpuSimilarityScoringMixin
from .similarity import Compute_Similarity_MI355X
from .slim_bpr import rows_slabs_to_csr


class _RandomWalkLogic:

    def _fit_walk(self, topK, alpha, beta, min_rating, implicit, normalize_similarity):
        if min_rating > 0:
            self.URM_train.data[self.URM_train.data < min_rating] = 0
            self.URM_train.eliminate_zeros()
            if implicit:
                self.URM_train.data = np.ones(self.URM_train.data.size, dtype=np.float32)
        URM = self.URM_train
        n_items = URM.shape[1]
        # Pui: l1 row-normalised URM (sklearn.normalize in the reference), to the power alpha
        row_sum = np.asarray(abs(URM).sum(axis=1)).ravel()
        scale = np.divide(1.0, row_sum, out=np.zeros_like(row_sum, dtype=np.float64), where=row_sum != 0)
        M = sps.csr_matrix(URM, dtype=np.float64, copy=True)
        M.data = M.data * np.repeat(scale, np.diff(M.indptr))
        if alpha != 1.0:
            M.data = np.power(M.data, alpha)
        item_count = np.diff(URM.tocsc().indptr).astype(np.float64)        # boolean column sums
        if beta is not None:                                                # RP3beta: penalise popular items (:56-64)
            degree = np.zeros(n_items)
            degree[item_count != 0] = np.power(item_count[item_count != 0], -beta)
            M.data = M.data * degree[M.indices]
        M = M.astype(np.float32)
        builder = Compute_Similarity_MI355X(M, topK=topK, shrink=0, normalize=False, similarity="cosine", unit_column_side=True)
        idx, val, _ = builder.compute_slabs()
        self.similarity_stats = builder.stats()
        builder.close()
        # Piu[i, u] = (1 / n_i)^alpha: one factor per ROW of W
        row_factor = np.zeros(n_items)
        row_factor[item_count != 0] = np.power(1.0 / item_count[item_count != 0], alpha)
        W = rows_slabs_to_csr(idx, (val * row_factor[:, None]).astype(np.float32), n_items)
        self.W_rowwise = W                  # per-row top-K of the walk, before normalisation / the column-wise cut
        if normalize_similarity:
            norm = np.asarray(abs(W).sum(axis=1)).ravel()
            inv = np.divide(1.0, norm, out=np.zeros_like(norm), where=norm != 0)
            W = sps.diags(inv).dot(W).tocsr()
        if topK is not False:
            W = similarityMatrixTopK(W, k=topK)
        self.W_sparse = check_matrix(W, format="csr")


class _P3alphaLogic(_RandomWalkLogic):
    """Drop-in for GraphBased/P3alphaRecommender.py:19."""
    RECOMMENDER_NAME = "P3alphaRecommender"

    def fit(self, topK=100, alpha=1., min_rating=0, implicit=False, normalize_similarity=False):
        self.topK, self.alpha, self.min_rating, self.implicit = topK, alpha, min_rating, implicit
        self.normalize_similarity = normalize_similarity
        self._fit_walk(topK, alpha, None, min_rating, implicit, normalize_similarity)


class _RP3betaLogic(_RandomWalkLogic):
    """Drop-in for GraphBased/RP3betaRecommender.py:16."""
    RECOMMENDER_NAME = "RP3betaRecommender"

    def fit(self, alpha=1., beta=0.6, min_rating=0, topK=100, implicit=False, normalize_similarity=True):
        self.alpha, self.beta, self.min_rating, self.topK, self.implicit = alpha, beta, min_rating, topK, implicit
        self.normalize_similarity = normalize_similarity
        self._fit_walk(topK, alpha, beta, min_rating, implicit, normalize_similarity)


class P3alphaRecommender(_P3alphaLogic, GpuSimilarityScoringMixin, BaseItemSimilarityMatrixRecommender):
    pass


class RP3betaRecommender(_RP3betaLogic, GpuSimilarityScoringMixin, BaseItemSimilarityMatrixRecommender):
    pass
