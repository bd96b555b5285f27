"""MI355X-native (gfx950) training kernels for the baseline recommenders of
MaurizioFD/RecSys2019_DeepLearning_Evaluation, behind the reference's own recommender surface.

Hot path only: Compute_Similarity (ItemKNN build), BPR-MF / FunkSVD SGD epochs, SLIM-BPR epoch,
IALS solve step.  Python host code + ctypes C-ABI (include/mi355rec.h) + hand-written HIP kernels (csrc/).
Nothing here imports torch; torch.distributed is only used by `sharding` for the multi-GPU gather.
"""
from ._native import ResidentURM  # noqa: F401
from .similarity import Compute_Similarity, Compute_Similarity_MI355X, Compute_Similarity_Euclidean_MI355X  # noqa: F401
from .knn import ItemKNNCFRecommender, UserKNNCFRecommender  # noqa: F401
from .matrix_factorization import (MatrixFactorization_MI355X_Epoch, MatrixFactorization_BPR_MI355X,  # noqa: F401
                                   MatrixFactorization_FunkSVD_MI355X, MatrixFactorization_AsySVD_MI355X,
                                   MatrixFactorization_MI355X_Group)

from .slim_bpr import SLIM_BPR_MI355X_Epoch, SLIM_BPR_MI355X  # noqa: F401,E402
from .scoring import MI355XScorer, MI355XSparseScorer, GpuScoringMixin, GpuSimilarityScoringMixin  # noqa: F401,E402
from .graph_based import P3alphaRecommender, RP3betaRecommender  # noqa: F401,E402
from .ease_r import EASE_R_Recommender  # noqa: F401,E402
from .ials import IALS_MI355X_Epoch, IALSRecommender  # noqa: F401,E402

__all__ = ["ResidentURM", "EASE_R_Recommender", "P3alphaRecommender", "RP3betaRecommender", "MI355XScorer", "MI355XSparseScorer", "GpuScoringMixin", "GpuSimilarityScoringMixin", "SLIM_BPR_MI355X_Epoch", "SLIM_BPR_MI355X", "IALS_MI355X_Epoch", "IALSRecommender", "Compute_Similarity", "Compute_Similarity_MI355X", "Compute_Similarity_Euclidean_MI355X", "ItemKNNCFRecommender", "UserKNNCFRecommender",
           "MatrixFactorization_MI355X_Epoch", "MatrixFactorization_MI355X_Group", "MatrixFactorization_BPR_MI355X", "MatrixFactorization_FunkSVD_MI355X", "MatrixFactorization_AsySVD_MI355X"]
