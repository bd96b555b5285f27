"""The device recommenders as subclasses of the REFERENCE's own base classes (last row).

The recommenders of this package are compositions `(logic mixin, device-scoring mixin, recommender base[, early stopping])`.
By default the last two are this package's re-provided copies of the reference's plugin surface (recommender_base.py), so
the package works where the reference tree is not importable.  Inside the reference tree a maintainer wants the REAL bases
-- `Base.BaseRecommender`, `Base.BaseMatrixFactorizationRecommender`, `Base.BaseSimilarityMatrixRecommender`,
`Base.Incremental_Training_Early_Stopping` -- so that `DataIO` persistence, `SearchBayesianSkopt`, `EvaluatorHoldout` and
`isinstance` checks see their own classes.  `bind()` builds exactly those subclasses:

    from Base.BaseMatrixFactorizationRecommender import BaseMatrixFactorizationRecommender
    from Base.BaseSimilarityMatrixRecommender import BaseItemSimilarityMatrixRecommender, BaseUserSimilarityMatrixRecommender
    from Base.Incremental_Training_Early_Stopping import Incremental_Training_Early_Stopping
    from recsys2019_deeplearning_evaluation_amd.reference_binding import bind
    R = bind(BaseMatrixFactorizationRecommender, BaseItemSimilarityMatrixRecommender, BaseUserSimilarityMatrixRecommender,
             Incremental_Training_Early_Stopping)
    rec = R.MatrixFactorization_BPR_MI355X(URM_train); rec.fit(epochs=300, num_factors=128, ...)

`device_scoring=False` leaves `recommend()` to the reference's own host implementation (Base/BaseRecommender.py:131).
"""
from types import SimpleNamespace

from .graph_based import _P3alphaLogic, _RP3betaLogic
from .ials import _IALSLogic
from .knn import _ItemKNNLogic, _UserKNNLogic
from .matrix_factorization import _AsySVDLogic, _BPRLogic, _FunkSVDLogic
from .scoring import GpuScoringMixin, GpuSimilarityScoringMixin
from .slim_bpr import _SLIMLogic


def bind(BaseMatrixFactorizationRecommender, BaseItemSimilarityMatrixRecommender, BaseUserSimilarityMatrixRecommender,
         Incremental_Training_Early_Stopping, device_scoring=True):
    """Returns a namespace with every recommender of this package rebuilt on the given (reference) base classes."""
    mf_score = (GpuScoringMixin,) if device_scoring else ()
    sim_score = (GpuSimilarityScoringMixin,) if device_scoring else ()
    mf = mf_score + (BaseMatrixFactorizationRecommender, Incremental_Training_Early_Stopping)
    table = {
        "MatrixFactorization_BPR_MI355X": (_BPRLogic,) + mf,
        "MatrixFactorization_FunkSVD_MI355X": (_FunkSVDLogic,) + mf,
        "MatrixFactorization_AsySVD_MI355X": (_AsySVDLogic,) + mf,
        "IALSRecommender": (_IALSLogic,) + mf,
        "SLIM_BPR_MI355X": (_SLIMLogic,) + sim_score + (BaseItemSimilarityMatrixRecommender, Incremental_Training_Early_Stopping),
        "ItemKNNCFRecommender": (_ItemKNNLogic,) + sim_score + (BaseItemSimilarityMatrixRecommender,),
        "UserKNNCFRecommender": (_UserKNNLogic,) + sim_score + (BaseUserSimilarityMatrixRecommender,),
        "P3alphaRecommender": (_P3alphaLogic,) + sim_score + (BaseItemSimilarityMatrixRecommender,),
        "RP3betaRecommender": (_RP3betaLogic,) + sim_score + (BaseItemSimilarityMatrixRecommender,),
    }
    return SimpleNamespace(**{name: type(name, bases, {"__doc__": bases[0].__doc__}) for name, bases in table.items()})
