"""IALS on MI355X: host front-end of the ials_* entry points of libmi355rec.so.

Mirrors IALSRecommender (MatrixFactorization/IALSRecommender.py:23): same fit() keywords, same confidence
scaling (:111-123, computed on the host exactly as the reference does, float32), same factor initialisation
(:204-210: ITEM_factors = k^-0.5 * U(0,1) drawn from NumPy's global stream; USER_factors are "don't care" and start
at zero here instead of np.empty garbage), same early-stopping hooks.  _run_epoch (:137) is what moves to the
device: both half-steps (Gramian, per-row normal equations, solve) run in float64 on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .recommender_base import (BaseMatrixFactorizationRecommender, Incremental_Training_Early_Stopping, check_matrix)
from .scoring import GpuScoringMixin


class IALS_MI355X_Epoch:
    """Device-resident IALS state: one object per fit(), like the Cython epoch objects of the SGD recommenders."""

    def __init__(self, C_csr, num_factors, reg, ITEM_factors, USER_factors=None):
        C_csr = check_matrix(C_csr, "csr", dtype=np.float32)
        if not C_csr.has_sorted_indices:
            C_csr = C_csr.sorted_indices()
        self.n_users, self.n_items = C_csr.shape
        self.num_factors = int(num_factors)
        indptr, indices, conf = N.as_i32(C_csr.indptr), N.as_i32(C_csr.indices), N.as_f32(C_csr.data)
        V0 = np.ascontiguousarray(ITEM_factors, dtype=np.float64)
        U0 = None if USER_factors is None else np.ascontiguousarray(USER_factors, dtype=np.float64)
        assert V0.shape == (self.n_items, self.num_factors)
        self._lib = N.load()
        self._h = C.c_void_p()
        N.check(self._lib.mi355rec_ials_create(C.byref(self._h), self.n_users, self.n_items, self.num_factors, float(reg),
                                               N.ptr(indptr), N.ptr(indices), N.ptr(conf), N.ptr(U0), N.ptr(V0)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355rec_ials_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_epochs(self, n_epochs=1):
        N.check(self._lib.mi355rec_ials_run_epochs(self._h, int(n_epochs)))

    def user_half(self, u0, u1):
        N.check(self._lib.mi355rec_ials_user_half(self._h, int(u0), int(u1)))

    def item_half(self, i0, i1):
        N.check(self._lib.mi355rec_ials_item_half(self._h, int(i0), int(i1)))

    def synchronize(self):
        N.check(self._lib.mi355rec_ials_sync(self._h))

    def device_factor_pointers(self):
        dU, dV = C.c_void_p(), C.c_void_p()
        N.check(self._lib.mi355rec_ials_device_factors(self._h, C.byref(dU), C.byref(dV)))
        return dU.value, dV.value

    def get_factors(self):
        U = np.empty((self.n_users, self.num_factors), np.float64)
        V = np.empty((self.n_items, self.num_factors), np.float64)
        N.check(self._lib.mi355rec_ials_get_factors(self._h, N.ptr(U), N.ptr(V)))
        return U, V

    def schedule_info(self):
        """(rows split over several workgroups, parts) of the last half-step."""
        a, b = C.c_int32(), C.c_int32()
        N.check(self._lib.mi355rec_ials_schedule_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats(self):
        st = N.Stats()
        N.check(self._lib.mi355rec_ials_get_stats(self._h, C.byref(st)))
        return st.as_dict()


class _IALSLogic:
    """Drop-in for the reference IALSRecommender with _run_epoch on the GPU (a mixin without bases, see matrix_factorization.py)."""
    RECOMMENDER_NAME = "IALSRecommender"
    AVAILABLE_CONFIDENCE_SCALING = ["linear", "log"]

    def fit(self, epochs=300, num_factors=20, confidence_scaling="linear", alpha=1.0, epsilon=1.0, reg=1e-3,
            init_mean=0.0, init_std=0.1, **earlystopping_kwargs):
        if confidence_scaling not in self.AVAILABLE_CONFIDENCE_SCALING:
            raise ValueError("Value for 'confidence_scaling' not recognized. Acceptable values are {}, provided was '{}'".format(
                self.AVAILABLE_CONFIDENCE_SCALING, confidence_scaling))
        self.num_factors = num_factors
        self.alpha = alpha
        self.epsilon = epsilon
        self.reg = reg
        self.USER_factors = np.zeros((self.n_users, num_factors))
        self.ITEM_factors = self.num_factors ** -0.5 * np.random.random_sample((self.n_items, num_factors))
        self._build_confidence_matrix(confidence_scaling)
        self.epoch_kernel = IALS_MI355X_Epoch(self.C, num_factors, reg, self.ITEM_factors, self.USER_factors)
        self._update_best_model()
        self._train_with_early_stopping(epochs, algorithm_name=self.RECOMMENDER_NAME, **earlystopping_kwargs)
        self.USER_factors = self.USER_factors_best
        self.ITEM_factors = self.ITEM_factors_best
        self.epoch_kernel.close()

    def _build_confidence_matrix(self, confidence_scaling):
        Cm = check_matrix(self.URM_train, format="csr", dtype=np.float32).copy()
        if confidence_scaling == "linear":
            Cm.data = 1.0 + self.alpha * Cm.data
        else:
            Cm.data = 1.0 + self.alpha * np.log(1.0 + Cm.data / self.epsilon)
        Cm.data = Cm.data.astype(np.float32)
        self.C = Cm

    def _prepare_model_for_validation(self):
        self.USER_factors, self.ITEM_factors = self.epoch_kernel.get_factors()
        invalidate = getattr(self, "invalidate_scorer", None)         # (the device scorer of the scoring mixin, if the class has one)
        if invalidate is not None:
            invalidate()

    def _update_best_model(self):
        self.USER_factors_best = self.USER_factors.copy()
        self.ITEM_factors_best = self.ITEM_factors.copy()

    def _run_epoch(self, num_epoch):
        self.epoch_kernel.run_epochs(1)


class IALSRecommender(_IALSLogic, GpuScoringMixin, BaseMatrixFactorizationRecommender, Incremental_Training_Early_Stopping):
    pass
