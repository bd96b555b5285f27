"""KNN collaborative-filtering recommenders whose similarity build runs on MI355X.

Mirrors KNN/ItemKNNCFRecommender.py:17 (fit :31-54) and KNN/UserKNNCFRecommender.py:17 (fit :31-55): same fit()
keywords, same W_sparse attribute, same scoring through the base classes.  Feature weighting (BM25 / TF-IDF,
Base/IR_feature_weighting.py) is the reference's host-side NumPy pre-step (feature_weighting.py); the similarity
build that follows runs on the device.
"""
import numpy as np

from .recommender_base import (BaseItemSimilarityMatrixRecommender, BaseUserSimilarityMatrixRecommender, check_matrix)
from .feature_weighting import apply_feature_weighting
from .scoring import GpuSimilarityScoringMixin
from .similarity import Compute_Similarity


class _KNNCFMixin:
    FEATURE_WEIGHTING_VALUES = ["BM25", "TF-IDF", "none"]

    def _check_weighting(self, feature_weighting):
        if feature_weighting not in self.FEATURE_WEIGHTING_VALUES:
            raise ValueError("Value for 'feature_weighting' not recognized. Acceptable values are {}, provided was '{}'".format(
                self.FEATURE_WEIGHTING_VALUES, feature_weighting))


class _ItemKNNLogic(_KNNCFMixin):
    """ItemKNN recommender: W_sparse = top-K item-item similarity of the URM columns."""
    RECOMMENDER_NAME = "ItemKNNCFRecommender"

    def __init__(self, URM_train, verbose=True):
        super(_ItemKNNLogic, self).__init__(URM_train, verbose=verbose)

    def fit(self, topK=50, shrink=100, similarity="cosine", normalize=True, feature_weighting="none", **similarity_args):
        self.topK = topK
        self.shrink = shrink
        self._check_weighting(feature_weighting)
        self.URM_train = apply_feature_weighting(self.URM_train, feature_weighting, user_major=False)
        builder = Compute_Similarity(self.URM_train, shrink=shrink, topK=topK, normalize=normalize,
                                     similarity=similarity, **similarity_args)
        self.W_sparse = builder.compute_similarity()
        self.W_sparse = check_matrix(self.W_sparse, format="csr")
        self.similarity_stats = builder.compute_similarity_object.stats()
        builder.compute_similarity_object.close()


class _UserKNNLogic(_KNNCFMixin):
    """UserKNN recommender: the same build on URM.T (columns = users); user bases wider than the LDS accumulator
    (32 256 cells) are handled by the kernel's accumulator tiling."""
    RECOMMENDER_NAME = "UserKNNCFRecommender"
    _SCORER_USER_BASED = True

    def __init__(self, URM_train, verbose=True):
        super(_UserKNNLogic, self).__init__(URM_train, verbose=verbose)

    def fit(self, topK=50, shrink=100, similarity="cosine", normalize=True, feature_weighting="none", **similarity_args):
        self.topK = topK
        self.shrink = shrink
        self._check_weighting(feature_weighting)
        self.URM_train = apply_feature_weighting(self.URM_train, feature_weighting, user_major=True)
        builder = Compute_Similarity(self.URM_train.T, shrink=shrink, topK=topK, normalize=normalize,
                                     similarity=similarity, **similarity_args)
        self.W_sparse = builder.compute_similarity()
        self.W_sparse = check_matrix(self.W_sparse, format="csr")
        builder.compute_similarity_object.close()


class ItemKNNCFRecommender(_ItemKNNLogic, GpuSimilarityScoringMixin, BaseItemSimilarityMatrixRecommender):
    pass


class UserKNNCFRecommender(_UserKNNLogic, GpuSimilarityScoringMixin, BaseUserSimilarityMatrixRecommender):
    pass
