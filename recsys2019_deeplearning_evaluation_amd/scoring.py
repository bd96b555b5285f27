"""Device scoring + ranking for factor models.

`MI355XScorer` keeps USER/ITEM factors (and biases) plus the "seen" CSR in HBM and answers the two questions the
reference's evaluation loop asks on every validation (Base/Evaluation/Evaluator.py:436):
  scores  = BaseMatrixFactorizationRecommender._compute_item_score   (BaseMatrixFactorizationRecommender.py:38-70)
  ranking = the filter + top-cutoff half of BaseRecommender.recommend (BaseRecommender.py:131-222)
`GpuScoringMixin` plugs it under `recommend()` of a BaseMatrixFactorizationRecommender without changing its signature;
`_compute_item_score` itself is left untouched (host NumPy), so either path can be checked against the other.
"""
import ctypes as C

import numpy as np

from . import _native as N


class MI355XScorer:
    def __init__(self, USER_factors, ITEM_factors, URM_seen, USER_bias=None, ITEM_bias=None, GLOBAL_bias=0.0):
        U, V = N.as_f32(USER_factors), N.as_f32(ITEM_factors)
        assert U.shape[1] == V.shape[1], "User and Item factors have inconsistent shape"
        self.n_users, self.n_items, self.n_factors = U.shape[0], V.shape[0], U.shape[1]
        self.use_bias = USER_bias is not None
        seen = URM_seen.tocsr()
        assert seen.shape == (self.n_users, self.n_items)
        indptr, indices = N.as_i32(seen.indptr), N.as_i32(seen.indices)
        bu = N.as_f32(USER_bias) if self.use_bias else None
        bi = N.as_f32(ITEM_bias) if self.use_bias else None
        self._lib = N.load()
        self._h = C.c_void_p()
        N.check(self._lib.mi355rec_scorer_create(C.byref(self._h), self.n_users, self.n_items, self.n_factors, N.ptr(U), N.ptr(V),
                                                 int(self.use_bias), N.ptr(bu), N.ptr(bi), float(np.asarray(GLOBAL_bias)),
                                                 N.ptr(indptr), N.ptr(indices)))

    def update(self, USER_factors, ITEM_factors, USER_bias=None, ITEM_bias=None, GLOBAL_bias=0.0):
        U, V = N.as_f32(USER_factors), N.as_f32(ITEM_factors)
        assert U.shape == (self.n_users, self.n_factors) and V.shape == (self.n_items, self.n_factors)
        bu = N.as_f32(USER_bias) if self.use_bias else None
        bi = N.as_f32(ITEM_bias) if self.use_bias else None
        N.check(self._lib.mi355rec_scorer_update(self._h, N.ptr(U), N.ptr(V), N.ptr(bu), N.ptr(bi), float(np.asarray(GLOBAL_bias))))

    def recommend(self, user_id_array, cutoff, remove_seen=True, allowed_items=None, return_scores=False):
        """ranked: int32 (n, cutoff) with -1 padding; scores: float32 (n, n_items) with -inf for filtered items."""
        users = N.as_i32(np.atleast_1d(user_id_array))
        cutoff = int(min(cutoff, self.n_items))
        ranked = np.empty((len(users), cutoff), np.int32)
        scores = np.empty((len(users), self.n_items), np.float32) if return_scores else None
        mask = None if allowed_items is None else np.ascontiguousarray(allowed_items, dtype=np.uint8)
        N.check(self._lib.mi355rec_scorer_recommend(self._h, N.ptr(users), len(users), cutoff, int(bool(remove_seen)),
                                                    N.ptr(mask), N.ptr(ranked), N.ptr(scores)))
        return ranked, scores

    def stats(self):
        st = N.Stats()
        N.check(self._lib.mi355rec_scorer_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355rec_scorer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _fingerprint(array):
    """Cheap content signature of a host array (256 strided elements + shape): an optimiser step moves practically every
    cell of a factor matrix, so in-place edits of an object the cache already holds are noticed without hashing it all."""
    a = np.asarray(array).ravel()
    if a.size == 0:
        return (0,)
    step = max(1, a.size // 256)
    return (np.asarray(array).shape, a[::step][:256].tobytes(), a[-1].tobytes())


class GpuScoringMixin:
    """recommend() of BaseRecommender (same signature, same return values) served by MI355XScorer.  The scorer is
    (re)built lazily from the host attributes USER_factors / ITEM_factors[/biases] and URM_train, so it follows
    early-stopping's _prepare_model_for_validation / best-model swaps and set_URM_train automatically.  The cache holds
    STRONG references to the objects it was built from and compares by identity (an id() of a freed object can be reused
    by its successor) plus a content fingerprint (in-place edits)."""
    _scorer = None
    _scorer_src = None

    def invalidate_scorer(self):
        """Forget the device copy of the model: the next recommend() uploads USER_factors / ITEM_factors again.  The cache notices
        REPLACED arrays (identity) and wholesale in-place changes (a strided 256-element fingerprint -- an optimiser step moves
        practically every cell); code that edits a few rows of a factor matrix in place (folding in cold users, say) must call
        this.  The package's own training loops call it from _prepare_model_for_validation."""
        self._scorer_src = None
        if self._scorer is not None:
            self._scorer.close()
            self._scorer = None

    def _scorer_sources(self):
        src = [self.USER_factors, self.ITEM_factors]
        if self.use_bias:
            src += [self.USER_bias, self.ITEM_bias, np.asarray(self.GLOBAL_bias)]
        return src

    def _get_scorer(self):
        src = self._scorer_sources()
        prints = [_fingerprint(a) for a in src]
        bias = dict(USER_bias=self.USER_bias, ITEM_bias=self.ITEM_bias, GLOBAL_bias=self.GLOBAL_bias) if self.use_bias else {}
        old = self._scorer_src
        rebuild = (self._scorer is None or self._scorer.use_bias != bool(self.use_bias) or old["urm"] is not self.URM_train
                   or self._scorer.n_factors != np.asarray(self.USER_factors).shape[1])
        if rebuild:
            if self._scorer is not None:
                self._scorer.close()
            self._scorer = MI355XScorer(self.USER_factors, self.ITEM_factors, self.URM_train, **bias)    # uploads the seen CSR too
        elif len(old["src"]) != len(src) or any(a is not b for a, b in zip(old["src"], src)) or old["prints"] != prints:
            self._scorer.update(self.USER_factors, self.ITEM_factors, **bias)
        self._scorer_src = {"src": src, "prints": prints, "urm": self.URM_train}
        return self._scorer

    def recommend(self, user_id_array, cutoff=None, remove_seen_flag=True, items_to_compute=None, remove_top_pop_flag=False,
                  remove_custom_items_flag=False, return_scores=False):
        single_user = np.isscalar(user_id_array)
        users = np.atleast_1d(user_id_array)
        if cutoff is None:
            cutoff = self.URM_train.shape[1] - 1
        allowed = None
        if items_to_compute is not None or remove_top_pop_flag or remove_custom_items_flag:
            allowed = np.zeros(self.n_items, np.uint8) if items_to_compute is not None else np.ones(self.n_items, np.uint8)
            if items_to_compute is not None:
                allowed[np.asarray(items_to_compute)] = 1
            if remove_top_pop_flag:
                allowed[self.filterTopPop_ItemsID] = 0
            if remove_custom_items_flag:
                allowed[self.items_to_ignore_ID] = 0
        scorer = self._get_scorer()
        assert scorer.n_users > np.max(users), \
            "{}: Cold users not allowed. Users in trained model are {}, requested prediction for users up to {}".format(
                self.RECOMMENDER_NAME, scorer.n_users, np.max(users))
        ranked, scores = scorer.recommend(users, cutoff, remove_seen_flag, allowed, return_scores)
        # (-1 pads rows whose user has fewer than `cutoff` admissible items: rare -- one C-level tolist() otherwise, a tenth of the
        # per-row masks' time on blocks of 1000 users)
        ranking_list = ranked.tolist() if ranked.size and int(ranked.min()) >= 0 else [row[row >= 0].tolist() for row in ranked]
        if single_user:
            ranking_list = ranking_list[0]
        return (ranking_list, scores) if return_scores else ranking_list


class MI355XSparseScorer:
    """scores[u] = A[u, :] . B for sparse A, B (ItemKNN / SLIM: A = URM_train, B = W_sparse; UserKNN: A = W_sparse,
    B = URM_train), filtered and ranked on the device like MI355XScorer."""

    def __init__(self, A, B, URM_seen):
        A, B, seen = A.tocsr(), B.tocsr(), URM_seen.tocsr()
        assert A.shape[1] == B.shape[0] and seen.shape == (A.shape[0], B.shape[1])
        self.n_users, self.n_items = A.shape[0], B.shape[1]
        arrs = [N.as_i32(A.indptr), N.as_i32(A.indices), N.as_f32(A.data), N.as_i32(B.indptr), N.as_i32(B.indices),
                N.as_f32(B.data), N.as_i32(seen.indptr), N.as_i32(seen.indices)]
        self._lib = N.load()
        self._h = C.c_void_p()
        N.check(self._lib.mi355rec_spscorer_create(C.byref(self._h), A.shape[0], A.shape[1], B.shape[1], *[N.ptr(a) for a in arrs]))

    def recommend(self, user_id_array, cutoff, remove_seen=True, allowed_items=None, return_scores=False):
        users = N.as_i32(np.atleast_1d(user_id_array))
        cutoff = int(min(cutoff, self.n_items))
        ranked = np.empty((len(users), cutoff), np.int32)
        scores = np.empty((len(users), self.n_items), np.float32) if return_scores else None
        mask = None if allowed_items is None else np.ascontiguousarray(allowed_items, dtype=np.uint8)
        N.check(self._lib.mi355rec_spscorer_recommend(self._h, N.ptr(users), len(users), cutoff, int(bool(remove_seen)),
                                                      N.ptr(mask), N.ptr(ranked), N.ptr(scores)))
        return ranked, scores

    def stats(self):
        st = N.Stats()
        N.check(self._lib.mi355rec_spscorer_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355rec_spscorer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuSimilarityScoringMixin:
    """recommend() for BaseItemSimilarityMatrixRecommender / BaseUserSimilarityMatrixRecommender subclasses, served by
    MI355XSparseScorer.  `_SCORER_USER_BASED` selects the operand order.  The scorer is rebuilt whenever W_sparse or
    URM_train is replaced (fit, early-stopping validation)."""
    _SCORER_USER_BASED = False
    _sp_scorer = None
    _sp_scorer_src = None

    def _get_sparse_scorer(self):
        # strong references + identity (SLIM's get_S_incremental_and_set_W assigns W_sparse twice per validation: the address of
        # the first, freed matrix can be handed to its successor, so an id() key would score with stale weights)
        W = self.W_sparse
        print_now = (W.shape, W.nnz, _fingerprint(W.data), _fingerprint(W.indices)) if hasattr(W, "nnz") else _fingerprint(W)
        old = self._sp_scorer_src
        if self._sp_scorer is None or old["W"] is not W or old["urm"] is not self.URM_train or old["print"] != print_now:
            if self._sp_scorer is not None:
                self._sp_scorer.close()
            A, B = (self.W_sparse, self.URM_train) if self._SCORER_USER_BASED else (self.URM_train, self.W_sparse)
            self._sp_scorer = MI355XSparseScorer(A, B, self.URM_train)
            self._sp_scorer_src = {"W": W, "urm": self.URM_train, "print": print_now}
        return self._sp_scorer

    def recommend(self, user_id_array, cutoff=None, remove_seen_flag=True, items_to_compute=None, remove_top_pop_flag=False,
                  remove_custom_items_flag=False, return_scores=False):
        single_user = np.isscalar(user_id_array)
        users = np.atleast_1d(user_id_array)
        if cutoff is None:
            cutoff = self.URM_train.shape[1] - 1
        allowed = None
        if items_to_compute is not None or remove_top_pop_flag or remove_custom_items_flag:
            allowed = np.zeros(self.n_items, np.uint8) if items_to_compute is not None else np.ones(self.n_items, np.uint8)
            if items_to_compute is not None:
                allowed[np.asarray(items_to_compute)] = 1
            if remove_top_pop_flag:
                allowed[self.filterTopPop_ItemsID] = 0
            if remove_custom_items_flag:
                allowed[self.items_to_ignore_ID] = 0
        ranked, scores = self._get_sparse_scorer().recommend(users, cutoff, remove_seen_flag, allowed, return_scores)
        # (-1 pads rows whose user has fewer than `cutoff` admissible items: rare -- one C-level tolist() otherwise, a tenth of the
        # per-row masks' time on blocks of 1000 users)
        ranking_list = ranked.tolist() if ranked.size and int(ranked.min()) >= 0 else [row[row >= 0].tolist() for row in ranked]
        if single_user:
            ranking_list = ranking_list[0]
        return (ranking_list, scores) if return_scores else ranking_list
