"""BPR-MF / FunkSVD SGD on MI355X: host front-end of the mf_* entry points of libmi355rec.so.

Mirrors
  MatrixFactorization_Cython_Epoch      MatrixFactorization/Cython/MatrixFactorization_Cython_Epoch.pyx:50
                                        (ctor :95-148, epochIteration_Cython :273, get_* :685-702)
  _MatrixFactorization_Cython           MatrixFactorization/Cython/MatrixFactorization_Cython.py:20 (fit :37)
  MatrixFactorization_BPR_Cython        :171      MatrixFactorization_FunkSVD_Cython  :192
The epoch object keeps the reference constructor's argument names / defaults / ValueErrors; factors are
initialised on the host exactly as the reference does (np.random.seed(seed); normal(init_mean, init_std_dev)
for U then V, .pyx:142-175).  On the device factors and optimiser moments are float32 for plain sgd and float64 for
adagrad / rmsprop / adam (`precision="auto"`; the reference computes in double and the adaptive normalisation amplifies
float32 rounding); the getters return float32 either way.  Sampling happens on the device
(counter-based RNG seeded by random_seed); `replay_samples` runs the identical arithmetic on a given sample
stream (parity mode).  AsySVD ("ASY_SVD", batch_size 1) runs its steps strictly in order on one workgroup.
"""
import ctypes as C
import sys

import numpy as np

from . import _native as N
from .recommender_base import (BaseMatrixFactorizationRecommender, Incremental_Training_Early_Stopping,
                               check_matrix)
from .scoring import GpuScoringMixin


class MatrixFactorization_MI355X_Epoch:
    SGD_MODE_VALUES = ["sgd", "adam", "adagrad", "rmsprop"]
    ALGORITHM_NAME_VALUES = ["FUNK_SVD", "ASY_SVD", "MF_BPR"]

    def __init__(self, URM_train, n_factors=1, algorithm_name=None, batch_size=1, negative_interactions_quota=0.5,
                 learning_rate=1e-3, use_bias=False, user_reg=0.0, item_reg=0.0, bias_reg=0.0, positive_reg=0.0,
                 negative_reg=0.0, verbose=False, random_seed=None, init_mean=0.0, init_std_dev=0.1,
                 sgd_mode="sgd", gamma=0.995, beta_1=0.9, beta_2=0.999,
                 initial_USER_factors=None, initial_ITEM_factors=None, precision="auto"):
        if sgd_mode not in self.SGD_MODE_VALUES:
            raise ValueError("Value for 'sgd_mode' not recognized. Acceptable values are {}, provided was '{}'".format(
                self.SGD_MODE_VALUES, sgd_mode))
        if algorithm_name not in self.ALGORITHM_NAME_VALUES:
            raise ValueError("Value for 'algorithm_name' not recognized. Acceptable values are {}, provided was '{}'".format(
                self.ALGORITHM_NAME_VALUES, algorithm_name))
        if algorithm_name == "ASY_SVD":
            assert batch_size == 1, "Batch size other than 1 not supported for ASY_SVD"
        if precision == "auto":
            # plain sgd on mini-batches holds the 1e-5 bar in float32; the adaptive optimisers need float64 state (see above), and so
            # does ASY_SVD: an epoch is nnz + 1 strictly sequential steps that each rewrite a whole profile of rows, so float32
            # rounding compounds a million times per epoch (element-wise error 2e-6 of max|Y| after ONE ML-1M epoch) -- and the
            # ordered kernel is latency-bound, not bandwidth-bound, so the wider type costs nothing
            precision = "fp32" if sgd_mode == "sgd" and algorithm_name != "ASY_SVD" else "fp64"
        if precision not in N.PRECISION_CODES:
            raise ValueError("Value for 'precision' not recognized. Acceptable values are {}, provided was '{}'".format(
                ["auto"] + list(N.PRECISION_CODES), precision))
        self.precision = precision
        URM_train = check_matrix(URM_train, "csr")
        URM_train = URM_train.sorted_indices()
        self.n_users, self.n_items = URM_train.shape
        self.n_factors = int(n_factors)
        self.batch_size = int(batch_size)
        self.algorithm_name = algorithm_name
        self.use_bias = bool(use_bias)
        self.verbose = verbose
        if random_seed is not None:
            np.random.seed(seed=random_seed)
        # same draw order as .pyx:174-175; an explicit initial model (parity tests) overrides the draw.  AsySVD keeps two
        # item-sized matrices (.pyx:163-166): its "USER_factors" is the n_items x k matrix Y
        self.n_user_rows = self.n_items if algorithm_name == "ASY_SVD" else self.n_users
        U0 = np.random.normal(init_mean, init_std_dev, (self.n_user_rows, self.n_factors))
        V0 = np.random.normal(init_mean, init_std_dev, (self.n_items, self.n_factors))
        if initial_USER_factors is not None:
            U0 = np.asarray(initial_USER_factors)
        if initial_ITEM_factors is not None:
            V0 = np.asarray(initial_ITEM_factors)
        assert U0.shape == (self.n_user_rows, self.n_factors) and V0.shape == (self.n_items, self.n_factors)
        seed = int(random_seed) if random_seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        cfg = N.MFConfig(N.ALGORITHM_CODES[algorithm_name], self.n_factors, self.batch_size, int(self.use_bias),
                         N.SGD_MODE_CODES[sgd_mode], learning_rate, user_reg, item_reg, bias_reg, positive_reg,
                         negative_reg, negative_interactions_quota, gamma, beta_1, beta_2, seed & (2 ** 64 - 1),
                         N.PRECISION_CODES[precision], 0)
        indptr, indices, data = N.as_i32(URM_train.indptr), N.as_i32(URM_train.indices), N.as_f32(URM_train.data)
        U0, V0 = (N.as_f64(U0), N.as_f64(V0)) if precision == "fp64" else (N.as_f32(U0), N.as_f32(V0))
        self._lib = N.load()
        self._h = C.c_void_p()
        N.check(self._lib.mi355rec_mf_create(C.byref(self._h), C.byref(cfg), self.n_users, self.n_items,
                                             N.ptr(indptr), N.ptr(indices), N.ptr(data), N.ptr(U0), N.ptr(V0)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355rec_mf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- training ----
    def epochIteration_Cython(self, n_epochs=1):
        """One reference epoch (n_users/B+1 or nnz/B+1 mini-batches); n_epochs>1 fuses several into one call."""
        N.check(self._lib.mi355rec_mf_run_epochs(self._h, int(n_epochs)))
        if self.verbose:
            st = self.stats()
            print("{}: Processed {} samples in {:.3f} seconds. loss {:.2E}. Sample per second: {:.0f}".format(
                self.algorithm_name, st["n_units"], st["call_ms"] * 1e-3, st["loss"] / max(1, st["n_units"]),
                st["n_units"] / max(1e-9, st["call_ms"] * 1e-3)))
            sys.stdout.flush()

    def replay_samples(self, user, item, neg_item=None, rating=None):
        user, item = N.as_i32(user), N.as_i32(item)
        neg_item = None if neg_item is None else N.as_i32(neg_item)
        rating = None if rating is None else N.as_f32(rating)
        N.check(self._lib.mi355rec_mf_run_samples(self._h, N.ptr(user), N.ptr(item), N.ptr(neg_item), N.ptr(rating), len(user)))

    def last_epoch_samples(self):
        """(user, item, neg_item | rating) drawn on the device during the last epoch of the last native call."""
        n = C.c_int64(0)
        N.check(self._lib.mi355rec_mf_get_last_samples(self._h, None, None, None, None, 0, C.byref(n)))
        u = np.empty(n.value, np.int32); i = np.empty(n.value, np.int32)
        j = np.empty(n.value, np.int32); r = np.empty(n.value, np.float32)
        N.check(self._lib.mi355rec_mf_get_last_samples(self._h, N.ptr(u), N.ptr(i), N.ptr(j), N.ptr(r), n.value, C.byref(n)))
        return (u, i, j) if self.algorithm_name == "MF_BPR" else (u, i, r)

    # ---- exact multi-GPU mini-batches (sharding.sharded_bpr_epoch drives these) ----
    def shard_begin_epoch(self, rank, world):
        """Draws and schedules one epoch; returns (send address, receive address, bytes per rank, mini-batches)."""
        send, recv, nbytes, nb = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_int32()
        N.check(self._lib.mi355rec_mf_shard_begin_epoch(self._h, int(rank), int(world), C.byref(send), C.byref(recv), C.byref(nbytes), C.byref(nb)))
        return send.value, recv.value, nbytes.value, nb.value

    def shard_batch(self, batch):
        N.check(self._lib.mi355rec_mf_shard_batch(self._h, int(batch)))

    def shard_merge(self, batch):
        N.check(self._lib.mi355rec_mf_shard_merge(self._h, int(batch)))

    def shard_end_epoch(self):
        N.check(self._lib.mi355rec_mf_shard_end_epoch(self._h))

    def stats(self):
        st = N.Stats()
        N.check(self._lib.mi355rec_mf_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def set_profiling(self, max_timed_launches):
        """Attach start/stop events to (at most) that many gradient-kernel dispatches of every following call."""
        N.check(self._lib.mi355rec_mf_set_profiling(self._h, int(max_timed_launches)))

    # ---- model read-back (fresh host copies; float64 like the reference's getters, .pyx:685-702, when the device state is
    # float64 -- adagrad / rmsprop / adam, AsySVD -- and the float32 state as it is otherwise) ----
    def _download(self, want_bias):
        f64 = self.precision == "fp64"
        dt = np.float64 if f64 else np.float32
        U = np.empty((self.n_user_rows, self.n_factors), dt)
        V = np.empty((self.n_items, self.n_factors), dt)
        bu = bi = mu = None
        if want_bias:
            bu = np.empty(self.n_users, dt); bi = np.empty(self.n_items, dt); mu = np.empty(1, dt)
        get = self._lib.mi355rec_mf_get_factors_f64 if f64 else self._lib.mi355rec_mf_get_factors
        N.check(get(self._h, N.ptr(U), N.ptr(V), N.ptr(bu), N.ptr(bi), N.ptr(mu)))
        return U, V, bu, bi, mu

    def get_factors(self):
        U, V, _, _, _ = self._download(False)
        return U, V

    def get_USER_factors(self):
        return self._download(False)[0]

    def get_ITEM_factors(self):
        return self._download(False)[1]

    def get_USER_bias(self):
        return self._download(True)[2]

    def get_ITEM_bias(self):
        return self._download(True)[3]

    def get_GLOBAL_bias(self):
        return np.array(self._download(True)[4][0])


class MatrixFactorization_MI355X_Group:
    """R independent epoch objects trained side by side: mini-batch b of ALL members is ONE launch (mi355rec_mf_group_*).

    The reference's hyper-parameter search runs this path as a pool of workers with one model each
    (ParameterTuning/run_parameter_search.py:498-503); one model's epoch is a chain of dependent ~3 MB mini-batches that
    fills a tenth of an MI355X, so the device-side form of that pool is a group.  Members keep their own factors,
    hyper-parameters, optimiser, seed and sample stream and end bit-identical to training alone; they must share
    algorithm_name (MF_BPR / FUNK_SVD), precision, batch_size, the URM's number of mini-batches per epoch and the kernel
    instance n_factors selects (see include/mi355rec.h).  The group does not own its members."""

    def __init__(self, members):
        self.members = list(members)
        assert len(self.members) >= 1
        self._lib = N.load()
        handles = (C.c_void_p * len(self.members))(*[m._h for m in self.members])
        self._member_handles = [m._h.value for m in self.members]      # the native group keeps these raw handles
        self._g = C.c_void_p()
        N.check(self._lib.mi355rec_mf_group_create(C.byref(self._g), handles, len(self.members)))

    def epochIteration_Cython(self, n_epochs=1):
        # a member that was closed (or closed and re-created) since the group was built would be a dangling handle on the device side
        for n, (m, h) in enumerate(zip(self.members, self._member_handles)):
            if getattr(m, "_h", None) is None or m._h.value != h:
                raise RuntimeError("MatrixFactorization_MI355X_Group: member %d was closed after the group was created; "
                                   "build a new group from live epoch objects" % n)
        N.check(self._lib.mi355rec_mf_group_run_epochs(self._g, int(n_epochs)))

    def set_profiling(self, max_timed_launches):
        N.check(self._lib.mi355rec_mf_group_set_profiling(self._g, int(max_timed_launches)))

    def stats(self):
        st = N.Stats()
        N.check(self._lib.mi355rec_mf_group_get_stats(self._g, C.byref(st)))
        return st.as_dict()

    def close(self):
        if getattr(self, "_g", None):
            self._lib.mi355rec_mf_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _MatrixFactorizationLogic:
    """fit() / early-stopping hooks of MatrixFactorization_Cython.py:20-170 over the device epoch object.  A mixin without
    bases: composed below with this package's re-provided recommender bases, and by reference_binding.bind() with the
    reference's own Base classes."""
    RECOMMENDER_NAME = "MatrixFactorization_MI355X_Recommender"

    def __init__(self, URM_train, verbose=True, algorithm_name="MF_BPR"):
        super(_MatrixFactorizationLogic, self).__init__(URM_train, verbose=verbose)
        self.n_users, self.n_items = self.URM_train.shape
        self.normalize = False
        self.algorithm_name = algorithm_name

    def fit(self, epochs=300, batch_size=1000, num_factors=10, positive_threshold_BPR=None, learning_rate=0.001,
            use_bias=True, sgd_mode="sgd", negative_interactions_quota=0.0, init_mean=0.0, init_std_dev=0.1,
            user_reg=0.0, item_reg=0.0, bias_reg=0.0, positive_reg=0.0, negative_reg=0.0, random_seed=None,
            **earlystopping_kwargs):
        self.num_factors = num_factors
        self.use_bias = use_bias
        self.sgd_mode = sgd_mode
        self.positive_threshold_BPR = positive_threshold_BPR
        self.learning_rate = learning_rate
        assert 0.0 <= negative_interactions_quota < 1.0, \
            "{}: negative_interactions_quota must be a float value >=0 and < 1.0, provided was '{}'".format(
                self.RECOMMENDER_NAME, negative_interactions_quota)
        self.negative_interactions_quota = negative_interactions_quota
        common = dict(algorithm_name=self.algorithm_name, n_factors=num_factors, learning_rate=learning_rate,
                      sgd_mode=sgd_mode, user_reg=user_reg, batch_size=batch_size, use_bias=use_bias,
                      init_mean=init_mean, init_std_dev=init_std_dev, verbose=self.verbose, random_seed=random_seed)
        if self.algorithm_name in ("FUNK_SVD", "ASY_SVD"):
            # as in the reference wrapper (.py:63-77) positive_reg is NOT forwarded, so items end up unregularised
            self.epoch_kernel = MatrixFactorization_MI355X_Epoch(
                self.URM_train, item_reg=item_reg, bias_reg=bias_reg,
                negative_interactions_quota=negative_interactions_quota, **common)
        else:
            URM_positive = self.URM_train.copy()
            if positive_threshold_BPR is not None:
                URM_positive.data = URM_positive.data >= positive_threshold_BPR
                URM_positive.eliminate_zeros()
                assert URM_positive.nnz > 0, \
                    "MatrixFactorization_Cython: URM_train_positive is empty, positive threshold is too high"
            self.epoch_kernel = MatrixFactorization_MI355X_Epoch(
                URM_positive, positive_reg=positive_reg, negative_reg=negative_reg, **common)
        self._prepare_model_for_validation()
        self._update_best_model()
        self._train_with_early_stopping(epochs, algorithm_name=self.algorithm_name, **earlystopping_kwargs)
        self.USER_factors = self.USER_factors_best
        self.ITEM_factors = self.ITEM_factors_best
        if self.use_bias:
            self.USER_bias = self.USER_bias_best
            self.ITEM_bias = self.ITEM_bias_best
            self.GLOBAL_bias = self.GLOBAL_bias_best
        self.epoch_kernel.close()
        sys.stdout.flush()

    def _forget_device_scorer(self):
        invalidate = getattr(self, "invalidate_scorer", None)         # (the device scorer of the scoring mixin, if the class has one)
        if invalidate is not None:
            invalidate()

    def _prepare_model_for_validation(self):     # the only device -> host copy of the training loop
        self.USER_factors, self.ITEM_factors = self.epoch_kernel.get_factors()
        self._forget_device_scorer()
        if self.use_bias:
            self.USER_bias = self.epoch_kernel.get_USER_bias()
            self.ITEM_bias = self.epoch_kernel.get_ITEM_bias()
            self.GLOBAL_bias = self.epoch_kernel.get_GLOBAL_bias()

    def _update_best_model(self):
        self.USER_factors_best = self.USER_factors.copy()
        self.ITEM_factors_best = self.ITEM_factors.copy()
        if self.use_bias:
            self.USER_bias_best = self.USER_bias.copy()
            self.ITEM_bias_best = self.ITEM_bias.copy()
            self.GLOBAL_bias_best = self.GLOBAL_bias

    def _run_epoch(self, num_epoch):
        self.epoch_kernel.epochIteration_Cython()


class _BPRLogic(_MatrixFactorizationLogic):
    """Drop-in for MatrixFactorization_BPR_Cython (forces use_bias=False, negative_interactions_quota=0)."""
    RECOMMENDER_NAME = "MatrixFactorization_BPR_MI355X_Recommender"

    def __init__(self, *pos_args, **key_args):
        super(_BPRLogic, self).__init__(*pos_args, algorithm_name="MF_BPR", **key_args)

    def fit(self, **key_args):
        key_args["use_bias"] = False
        key_args["negative_interactions_quota"] = 0.0
        super(_BPRLogic, self).fit(**key_args)


class _FunkSVDLogic(_MatrixFactorizationLogic):
    """Drop-in for MatrixFactorization_FunkSVD_Cython."""
    RECOMMENDER_NAME = "MatrixFactorization_FunkSVD_MI355X_Recommender"

    def __init__(self, *pos_args, **key_args):
        super(_FunkSVDLogic, self).__init__(*pos_args, algorithm_name="FUNK_SVD", **key_args)

    def fit(self, **key_args):
        super(_FunkSVDLogic, self).fit(**key_args)


class _AsySVDLogic(_MatrixFactorizationLogic):
    """Drop-in for MatrixFactorization_AsySVD_Cython (MatrixFactorization_Cython.py:219): two item-sized factor matrices;
    a user's factors are the sum of the Y rows of its profile divided by sqrt(profile length)."""
    RECOMMENDER_NAME = "MatrixFactorization_AsySVD_MI355X_Recommender"

    def __init__(self, *pos_args, **key_args):
        super(_AsySVDLogic, self).__init__(*pos_args, algorithm_name="ASY_SVD", **key_args)

    def fit(self, **key_args):
        if key_args.get("batch_size", 1) > 1:
            print("{}: batch_size not supported for this recommender, setting to default value 1.".format(self.RECOMMENDER_NAME))
        key_args["batch_size"] = 1
        super(_AsySVDLogic, self).fit(**key_args)

    def _prepare_model_for_validation(self):
        self.ITEM_factors_Y, self.ITEM_factors = self.epoch_kernel.get_factors()
        self.USER_factors = self._estimate_user_factors(self.ITEM_factors_Y)
        self._forget_device_scorer()
        if self.use_bias:
            self.USER_bias = self.epoch_kernel.get_USER_bias()
            self.ITEM_bias = self.epoch_kernel.get_ITEM_bias()
            self.GLOBAL_bias = self.epoch_kernel.get_GLOBAL_bias()

    def _update_best_model(self):
        super(_AsySVDLogic, self)._update_best_model()
        self.ITEM_factors_Y_best = self.ITEM_factors_Y.copy()

    def set_URM_train(self, URM_train_new, estimate_item_similarity_for_cold_users=False, **kwargs):
        """MatrixFactorization_Cython.py:257-278: the user factors of AsySVD are a function of the URM, so a new URM (the
        evaluator's cold-user scenarios) re-estimates them from the learned Y."""
        super(_AsySVDLogic, self).set_URM_train(URM_train_new, **kwargs)
        if estimate_item_similarity_for_cold_users:           # (sic: the reference's keyword; it re-estimates USER_factors)
            self._print("Estimating USER_factors for cold users...")
            self.USER_factors = self._estimate_user_factors(self.ITEM_factors_Y_best)
            self._print("Estimating USER_factors for cold users... done!")

    def _estimate_user_factors(self, ITEM_factors_Y):
        # MatrixFactorization_Cython.py:281-304: URM . Y, every row divided by sqrt(profile length)
        length_sqrt = np.sqrt(np.ediff1d(self.URM_train.indptr))
        USER_factors = self.URM_train.dot(ITEM_factors_Y)
        warm = length_sqrt > 0
        USER_factors[warm] /= length_sqrt[warm][:, None]
        return USER_factors


_BASES = (GpuScoringMixin, BaseMatrixFactorizationRecommender, Incremental_Training_Early_Stopping)


class MatrixFactorization_BPR_MI355X(_BPRLogic, *_BASES):
    pass


class MatrixFactorization_FunkSVD_MI355X(_FunkSVDLogic, *_BASES):
    pass


class MatrixFactorization_AsySVD_MI355X(_AsySVDLogic, *_BASES):
    pass
