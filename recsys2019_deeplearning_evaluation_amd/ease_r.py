"""EASE_R on MI355X: the Gram step of EASE_R/EASE_R_Recommender.py:55-65 -- which the
reference obtains from `Compute_Similarity(URM, shrink=0, topK=n_items, normalize=False, similarity="cosine")` -- runs on
the similarity kernel's dense (topK = 0) path; the k x k inverse stays the reference's own `np.linalg.inv` call on the
host (a third-party LAPACK solve, outside the hot path), in the float32 the reference feeds it.
"""
import numpy as np
import scipy.sparse as sps

from .recommender_base import BaseItemSimilarityMatrixRecommender, similarityMatrixTopK
from .similarity import Compute_Similarity_MI355X


def _l2_normalize(X, axis):
    """sklearn.preprocessing.normalize(X, norm='l2', axis=axis) for a sparse matrix (EASE_R_Recommender.py:47-51)."""
    X = sps.csr_matrix(X, dtype=np.float32) if axis == 1 else sps.csc_matrix(X, dtype=np.float32)
    norms = np.sqrt(np.asarray(X.multiply(X).sum(axis=axis), dtype=np.float64)).ravel()
    norms[norms == 0.0] = 1.0
    X.data = (X.data / np.repeat(norms, np.diff(X.indptr))).astype(np.float32)
    return X


class EASE_R_Recommender(BaseItemSimilarityMatrixRecommender):
    """Drop-in for EASE_R/EASE_R_Recommender.py:20: same `fit(topK=None, l2_norm=1e3, normalize_matrix=False)`;
    `W_sparse` is the dense item-item matrix B (topK=None) or its column-wise top-K (`similarityMatrixTopK`) as a csr_matrix."""

    RECOMMENDER_NAME = "EASE_R_Recommender"

    def __init__(self, URM_train, verbose=True):
        super(EASE_R_Recommender, self).__init__(URM_train, verbose=verbose)

    def fit(self, topK=None, l2_norm=1e3, normalize_matrix=False, verbose=True):
        self.verbose = verbose
        if normalize_matrix:                                    # rows, then columns (:47-51)
            self.URM_train = sps.csr_matrix(_l2_normalize(_l2_normalize(self.URM_train, 1), 0))
        builder = Compute_Similarity_MI355X(self.URM_train, topK=0, shrink=0, normalize=False, similarity="cosine")
        gram = builder.compute_similarity()                     # dense (n_items, n_items) float32, zero diagonal
        self.similarity_stats = builder.stats()
        builder.close()
        diag = np.diag_indices(gram.shape[0])
        item_popularity = np.ediff1d(self.URM_train.tocsc().indptr)        # sic (:63): the stored-cell count
        gram[diag] = item_popularity + l2_norm
        P = np.linalg.inv(gram)
        B = P / (-np.diag(P))
        B[diag] = 0.0
        if topK is None:
            self.W_sparse = B
            self._compute_item_score = self._compute_score_W_dense
        else:
            self.W_sparse = sps.csr_matrix(similarityMatrixTopK(B, k=topK, verbose=False))

    def _compute_score_W_dense(self, user_id_array, items_to_compute=None):
        user_profile_array = self.URM_train[user_id_array]
        if items_to_compute is not None:
            item_scores = -np.ones((len(user_id_array), self.URM_train.shape[1]), dtype=np.float32) * np.inf
            item_scores_all = user_profile_array.dot(self.W_sparse)
            item_scores[:, items_to_compute] = item_scores_all[:, items_to_compute]
        else:
            item_scores = user_profile_array.dot(self.W_sparse)
        return item_scores

    def load_model(self, folder_path, file_name=None):
        super(EASE_R_Recommender, self).load_model(folder_path, file_name=file_name)
        if not sps.issparse(self.W_sparse):
            self._compute_item_score = self._compute_score_W_dense
