"""KNN collaborative-filtering recommenders whose similarity build runs on MI355X.

Mirrors KNN/ItemKNNCFRecommender.py:17 (fit :31-54) and KNN/UserKNNCFRecommender.py:17 (fit :31-55): same fit()
keywords, same W_sparse attribute, same scoring through the base classes.  Feature weighting (BM25 / TF-IDF,
Base/IR_feature_weighting.py) is a device pre-pass of the similarity constructor on the values it has just uploaded
(at ML-20M shape the reference's NumPy version costs more than a hundred similarity builds); the recommender's URM_train is
replaced by the re-weighted matrix, as in the reference.
"""
import numpy as np

from .recommender_base import (BaseItemSimilarityMatrixRecommender, BaseUserSimilarityMatrixRecommender, check_matrix)
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

    def fit(self, topK=50, shrink=100, similarity="cosine", normalize=True, feature_weighting="none", resident_urm=None,
            **similarity_args):
        """resident_urm (not an argument of the reference): a `ResidentURM` of this URM_train, uploaded once for a whole search --
        the build then starts from the device copy (the constructor verifies that it holds the same matrix)."""
        self.topK = topK
        self.shrink = shrink
        self._check_weighting(feature_weighting)
        if resident_urm is not None:
            similarity_args["resident"] = resident_urm
        # okapi_BM_25(URM.T).T / TF_IDF(URM.T).T (ItemKNNCFRecommender.py:40-48): documents = items = columns of the URM
        builder = Compute_Similarity(self.URM_train.astype(np.float32, copy=False), shrink=shrink, topK=topK, normalize=normalize,
                                     similarity=similarity, feature_weighting=feature_weighting, weighting_documents="columns",
                                     **similarity_args)
        if feature_weighting != "none":
            self.URM_train = builder.compute_similarity_object.weighted_matrix()
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
        # the same okapi_BM_25(URM.T).T / TF_IDF(URM.T).T (UserKNNCFRecommender.py:40-48): the build runs on URM.T, whose ROWS
        # are the documents (items)
        builder = Compute_Similarity(self.URM_train.astype(np.float32).T, shrink=shrink, topK=topK, normalize=normalize,
                                     similarity=similarity, feature_weighting=feature_weighting, weighting_documents="rows",
                                     **similarity_args)
        if feature_weighting != "none":
            self.URM_train = check_matrix(builder.compute_similarity_object.weighted_matrix().T, "csr")
        self.W_sparse = builder.compute_similarity()
        self.W_sparse = check_matrix(self.W_sparse, format="csr")
        builder.compute_similarity_object.close()


class ItemKNNCFRecommender(_ItemKNNLogic, GpuSimilarityScoringMixin, BaseItemSimilarityMatrixRecommender):
    pass


class UserKNNCFRecommender(_UserKNNLogic, GpuSimilarityScoringMixin, BaseUserSimilarityMatrixRecommender):
    pass
