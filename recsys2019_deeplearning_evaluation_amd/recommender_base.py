"""Host-side recommender surface (the reference's plugin API), re-provided so the device recommenders are
usable where /root/reference is not importable, and duck-type compatible with the reference's harness.

Mirrors (same method names, argument meaning and error behaviour):
  BaseRecommender                     Base/BaseRecommender.py:14            (__init__, recommend :131, set_items_to_ignore ...)
  BaseMatrixFactorizationRecommender  Base/BaseMatrixFactorizationRecommender.py:15  (_compute_item_score :38, save_model :81)
  BaseItemSimilarityMatrixRecommender Base/BaseSimilarityMatrixRecommender.py:69     (_compute_item_score :73)
  BaseUserSimilarityMatrixRecommender Base/BaseSimilarityMatrixRecommender.py:97
  Incremental_Training_Early_Stopping Base/Incremental_Training_Early_Stopping.py:15 (_train_with_early_stopping :91)
  check_matrix / similarityMatrixTopK Base/Recommender_utils.py:13 / :55

`EvaluatorHoldout.evaluateRecommender(rec)` (Base/Evaluation/Evaluator.py:225) only calls
rec.recommend(users, remove_seen_flag, cutoff, remove_top_pop_flag, remove_custom_items_flag,
return_scores=True), rec.get_URM_train(), rec.set_items_to_ignore / reset_items_to_ignore -- all provided
here -- so these classes drop into the reference evaluator unchanged.  When used *inside* the reference tree a
maintainer can equally mix the kernel front-ends into the reference's own base classes.
"""
import io
import json
import os
import time
import zipfile

import numpy as np
import scipy.sparse as sps


def check_matrix(X, format="csc", dtype=np.float32):
    """Convert to the requested sparse format (dtype applied on conversion only), Recommender_utils.py:13."""
    if isinstance(X, np.ndarray) and format != "npy":
        X = sps.csr_matrix(X, dtype=dtype)
        X.eliminate_zeros()
    if format == "npy":
        return X.toarray().astype(dtype) if sps.issparse(X) else np.array(X)
    wanted = {"csr": sps.csr_matrix, "csc": sps.csc_matrix, "coo": sps.coo_matrix}[format]
    if isinstance(X, wanted):
        return X
    return getattr(X, "to" + format)().astype(dtype)


def similarityMatrixTopK(item_weights, k=100, verbose=False):
    """Keep, column by column, the k largest non-zero cells (Recommender_utils.py:55). Returns CSC float32.

    Vectorised: columns that already hold at most k non-zero cells are kept as they are (the usual case after a
    per-row top-K), only the others are ranked; a dense input is ranked with one argpartition along the rows."""
    n = item_weights.shape[1]
    assert item_weights.shape[0] == n, "selectTopK: ItemWeights is not a square matrix"
    k = min(k, n)
    if isinstance(item_weights, np.ndarray):
        W = np.asarray(item_weights, dtype=np.float32)
        if k < n:
            ranked = np.where(W != 0, -W, np.inf)                   # zero cells do not compete (:100-104)
            top = np.argpartition(ranked, k - 1, axis=0)[:k]        # (k, n) row ids of the k largest non-zero cells per column
        else:
            top = np.broadcast_to(np.arange(n)[:, None], (n, n))
        vals = np.take_along_axis(W, top, axis=0)
        keep = vals != 0
        cols = np.broadcast_to(np.arange(n)[None, :], top.shape)
        return sps.csc_matrix((vals[keep], (top[keep], cols[keep])), shape=(n, n), dtype=np.float32)
    W = check_matrix(item_weights, "csc", dtype=np.float32)
    if W is item_weights:                      # (a conversion has made a new matrix already: zeros are eliminated in place below)
        W = W.copy()
    W.eliminate_zeros()
    counts = np.diff(W.indptr)
    keep = np.ones(W.nnz, dtype=bool)
    # Over-full columns, ranked together: a stable ascending sort of a column drops its first len - k entries -- i.e. keeps every
    # entry above the k-th largest value T and, of the entries equal to T, the LAST ones (highest rows).  One argsort call per column
    # was 9 000 calls and 0.09 s at ML-20M size (every validation of a SLIM fit pays it); here the columns are padded into a few
    # 2-D blocks by length (powers of two: at most twice the entries), T comes from one argpartition per block and the tie rule
    # from a reversed cumulative count.
    full = np.flatnonzero(counts > k)
    if len(full):
        lengths = counts[full]
        bucket = np.ceil(np.log2(lengths)).astype(np.int64)
        for b in np.unique(bucket):
            cols = full[bucket == b]
            width = int(counts[cols].max())
            starts, lens = W.indptr[cols].astype(np.int64), counts[cols].astype(np.int64)
            pos = starts[:, None] + np.arange(width, dtype=np.int64)[None, :]
            real = np.arange(width)[None, :] < lens[:, None]
            vals = np.where(real, W.data[np.minimum(pos, W.nnz - 1)], -np.inf)
            kth = -np.partition(-vals, k - 1, axis=1)[:, k - 1]                      # the k-th largest value of every column
            above = vals > kth[:, None]
            tied = real & (vals == kth[:, None])
            wanted = k - above.sum(axis=1)                                            # how many of the tied entries belong to the top k
            from_end = np.cumsum(tied[:, ::-1], axis=1)[:, ::-1]                      # (a tied entry's rank among the tied ones, counted from the end)
            kept = above | (tied & (from_end <= wanted[:, None]))
            drop = real & ~kept
            keep[pos[drop]] = False
    indptr = np.concatenate([[0], np.cumsum(np.minimum(counts, k))])
    return sps.csc_matrix((W.data[keep], W.indices[keep], indptr), shape=(n, n), dtype=np.float32)


class BaseRecommender(object):
    RECOMMENDER_NAME = "Recommender_Base_Class"

    def __init__(self, URM_train, verbose=True):
        super(BaseRecommender, self).__init__()
        self.URM_train = check_matrix(URM_train.copy(), "csr", dtype=np.float32)
        self.URM_train.eliminate_zeros()
        self.n_users, self.n_items = self.URM_train.shape
        self.verbose = verbose
        self.filterTopPop = False
        self.filterTopPop_ItemsID = np.array([], dtype=int)
        self.items_to_ignore_flag = False
        self.items_to_ignore_ID = np.array([], dtype=int)
        self._cold_user_mask = np.ediff1d(self.URM_train.indptr) == 0
        if self._cold_user_mask.any():
            self._print("URM Detected {} ({:.2f} %) cold users.".format(
                self._cold_user_mask.sum(), self._cold_user_mask.sum() / self.n_users * 100))
        self._cold_item_mask = np.ediff1d(self.URM_train.tocsc().indptr) == 0
        if self._cold_item_mask.any():
            self._print("URM Detected {} ({:.2f} %) cold items.".format(
                self._cold_item_mask.sum(), self._cold_item_mask.sum() / self.n_items * 100))

    def _get_cold_user_mask(self):
        return self._cold_user_mask

    def _get_cold_item_mask(self):
        return self._cold_item_mask

    def _print(self, string):
        if self.verbose:
            print("{}: {}".format(self.RECOMMENDER_NAME, string))

    def fit(self):
        pass

    def get_URM_train(self):
        return self.URM_train.copy()

    def set_URM_train(self, URM_train_new, **kwargs):
        assert self.URM_train.shape == URM_train_new.shape, \
            "{}: set_URM_train old and new URM train have different shapes".format(self.RECOMMENDER_NAME)
        if kwargs:
            self._print("set_URM_train keyword arguments not supported for this recommender class. Received: {}".format(kwargs))
        self.URM_train = check_matrix(URM_train_new.copy(), "csr", dtype=np.float32)
        self.URM_train.eliminate_zeros()
        self._cold_user_mask = np.ediff1d(self.URM_train.indptr) == 0

    def set_items_to_ignore(self, items_to_ignore):
        self.items_to_ignore_flag = True
        self.items_to_ignore_ID = np.array(items_to_ignore, dtype=int)

    def reset_items_to_ignore(self):
        self.items_to_ignore_flag = False
        self.items_to_ignore_ID = np.array([], dtype=int)

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        raise NotImplementedError("BaseRecommender: compute_item_score not assigned for current recommender")

    def recommend(self, user_id_array, cutoff=None, remove_seen_flag=True, items_to_compute=None,
                  remove_top_pop_flag=False, remove_custom_items_flag=False, return_scores=False):
        """Ranked item lists (BaseRecommender.py:131-222): -inf marks excluded items, which are dropped."""
        single_user = np.isscalar(user_id_array)
        if single_user:
            user_id_array = np.atleast_1d(user_id_array)
        if cutoff is None:
            cutoff = self.URM_train.shape[1] - 1
        scores = self._compute_item_score(user_id_array, items_to_compute=items_to_compute)
        if remove_seen_flag:
            assert self.URM_train.getformat() == "csr"
            for row, user in enumerate(user_id_array):
                seen = self.URM_train.indices[self.URM_train.indptr[user]:self.URM_train.indptr[user + 1]]
                scores[row, seen] = -np.inf
        if remove_top_pop_flag:
            scores[:, self.filterTopPop_ItemsID] = -np.inf
        if remove_custom_items_flag:
            scores[:, self.items_to_ignore_ID] = -np.inf
        part = (-scores).argpartition(cutoff, axis=1)[:, 0:cutoff]
        rows = np.arange(scores.shape[0])[:, None]
        order = np.argsort(-scores[rows, part], axis=1)
        ranking = part[rows, order]
        ranking_list = []
        for row in range(len(user_id_array)):
            items = ranking[row]
            ranking_list.append(items[np.isfinite(scores[row, items])].tolist())
        if single_user:
            ranking_list = ranking_list[0]
        return (ranking_list, scores) if return_scores else ranking_list

    # ---- persistence: the file layout of the reference's DataIO (Base/DataIO.py:102-186), so that models saved here load
    # with `DataIO.load_data` / the reference's `load_model` and vice versa: a zip with one member per attribute (.npy for
    # arrays, .npz for sparse matrices, .json for the rest) plus ".DataIO_attribute_to_file_name.json" mapping attribute -> member
    _DATAIO_INDEX = ".DataIO_attribute_to_file_name"

    def _save_dict(self, folder_path, file_name, data):
        if file_name is None:
            file_name = self.RECOMMENDER_NAME
        os.makedirs(folder_path, exist_ok=True)
        path = os.path.join(folder_path, file_name + ("" if file_name.endswith(".zip") else ".zip"))
        self._print("Saving model in file '{}'".format(path))
        index = {}
        with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED) as z:
            for name, value in data.items():
                if sps.issparse(value):
                    index[name] = name + ".npz"
                    with z.open(index[name], "w") as f:
                        sps.save_npz(f, value)
                elif isinstance(value, np.ndarray):
                    index[name] = name + ".npy"
                    with z.open(index[name], "w") as f:
                        np.save(f, value, allow_pickle=False)
                else:
                    index[name] = name + ".json"
                    z.writestr(index[name], json.dumps(value if not isinstance(value, np.generic) else value.item()))
            z.writestr(self._DATAIO_INDEX + ".json", json.dumps(index))
        self._print("Saving complete")

    def save_model(self, folder_path, file_name=None):
        raise NotImplementedError("BaseRecommender: save_model not implemented")

    def load_model(self, folder_path, file_name=None):
        if file_name is None:
            file_name = self.RECOMMENDER_NAME
        path = os.path.join(folder_path, file_name + ("" if file_name.endswith(".zip") else ".zip"))
        self._print("Loading model from file '{}'".format(path))
        with zipfile.ZipFile(path) as z:
            names = z.namelist()
            index_member = next((m for m in (self._DATAIO_INDEX + ".json", "__DataIO_attribute_to_file_name.json") if m in names), None)
            if index_member is not None:
                index = json.loads(z.read(index_member).decode())
            else:       # (files written by round-1 builds of this package carry no index)
                index = {os.path.splitext(m)[0]: m for m in names}
            for name, member in index.items():
                loaded, value = self._load_dataio_member(z.read(member), os.path.splitext(member)[1])
                if loaded:
                    setattr(self, name, value)
                else:
                    self._print("load_model: skipping attribute '{}' (member '{}': format not handled)".format(name, member))
        self._print("Loading complete")

    def _load_dataio_member(self, raw, ext):
        """One member of a DataIO archive (Base/DataIO.py:102-186 writes .npy / .npz / .json, .csv for DataFrames and a nested .zip
        for dictionaries whose values need their own members).  Returns (handled, value)."""
        if ext == ".npz":
            return True, sps.load_npz(io.BytesIO(raw))
        if ext == ".npy":
            return True, np.load(io.BytesIO(raw), allow_pickle=False)
        if ext == ".json":
            return True, json.loads(raw.decode())
        if ext == ".csv":
            try:
                import pandas as pd
            except ImportError:
                return False, None
            return True, pd.read_csv(io.BytesIO(raw), index_col=False)
        if ext == ".zip":
            value = {}
            with zipfile.ZipFile(io.BytesIO(raw)) as inner:
                names = inner.namelist()
                index_member = next((m for m in (self._DATAIO_INDEX + ".json", "__DataIO_attribute_to_file_name.json") if m in names), None)
                index = json.loads(inner.read(index_member).decode()) if index_member else {os.path.splitext(m)[0]: m for m in names}
                for key, member in index.items():
                    ok, item = self._load_dataio_member(inner.read(member), os.path.splitext(member)[1])
                    if ok:
                        value[key] = item
            return True, value
        return False, None


class BaseMatrixFactorizationRecommender(BaseRecommender):
    """score = USER_factors[u] . ITEM_factors^T (+ biases); BaseMatrixFactorizationRecommender.py:38-70."""

    def __init__(self, URM_train, verbose=True):
        super(BaseMatrixFactorizationRecommender, self).__init__(URM_train, verbose=verbose)
        self.use_bias = False

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        assert self.USER_factors.shape[1] == self.ITEM_factors.shape[1], \
            "{}: User and Item factors have inconsistent shape".format(self.RECOMMENDER_NAME)
        assert self.USER_factors.shape[0] > np.max(user_id_array), \
            "{}: Cold users not allowed. Users in trained model are {}, requested prediction for users up to {}".format(
                self.RECOMMENDER_NAME, self.USER_factors.shape[0], np.max(user_id_array))
        if items_to_compute is not None:
            item_scores = -np.ones((len(user_id_array), self.ITEM_factors.shape[0]), dtype=np.float32) * np.inf
            item_scores[:, items_to_compute] = np.dot(self.USER_factors[user_id_array], self.ITEM_factors[items_to_compute, :].T)
        else:
            item_scores = np.dot(self.USER_factors[user_id_array], self.ITEM_factors.T)
        if self.use_bias:
            item_scores += self.ITEM_bias + self.GLOBAL_bias
            item_scores = (item_scores.T + self.USER_bias[user_id_array]).T
        return item_scores

    def save_model(self, folder_path, file_name=None):
        data = {"USER_factors": self.USER_factors, "ITEM_factors": self.ITEM_factors, "use_bias": bool(self.use_bias)}
        if self.use_bias:
            data["ITEM_bias"] = self.ITEM_bias
            data["USER_bias"] = self.USER_bias
            data["GLOBAL_bias"] = np.asarray(self.GLOBAL_bias)
        self._save_dict(folder_path, file_name, data)


class BaseSimilarityMatrixRecommender(BaseRecommender):
    def __init__(self, URM_train, verbose=True):
        super(BaseSimilarityMatrixRecommender, self).__init__(URM_train, verbose=verbose)

    def save_model(self, folder_path, file_name=None):
        self._save_dict(folder_path, file_name, {"W_sparse": self.W_sparse})


class BaseItemSimilarityMatrixRecommender(BaseSimilarityMatrixRecommender):
    """score = URM[u] . W_sparse; BaseSimilarityMatrixRecommender.py:73-92."""

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        profiles = self.URM_train[user_id_array]
        all_scores = profiles.dot(self.W_sparse).toarray()
        if items_to_compute is None:
            return all_scores
        item_scores = -np.ones((len(user_id_array), self.URM_train.shape[1]), dtype=np.float32) * np.inf
        item_scores[:, items_to_compute] = all_scores[:, items_to_compute]
        return item_scores


class BaseUserSimilarityMatrixRecommender(BaseSimilarityMatrixRecommender):
    """score = W_sparse[u] . URM; BaseSimilarityMatrixRecommender.py:101-116."""

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        weights = self.W_sparse[user_id_array]
        all_scores = weights.dot(self.URM_train).toarray()
        if items_to_compute is None:
            return all_scores
        item_scores = -np.ones((len(user_id_array), self.URM_train.shape[1]), dtype=np.float32) * np.inf
        item_scores[:, items_to_compute] = all_scores[:, items_to_compute]
        return item_scores


class Incremental_Training_Early_Stopping(object):
    """Epoch loop with optional periodic validation and early stopping
    (Base/Incremental_Training_Early_Stopping.py:91-262).  Subclasses provide _run_epoch(num_epoch),
    _prepare_model_for_validation() and _update_best_model()."""

    def get_early_stopping_final_epochs_dict(self):
        return {"epochs": self.epochs_best}

    def _run_epoch(self, num_epoch):
        raise NotImplementedError()

    def _prepare_model_for_validation(self):
        raise NotImplementedError()

    def _update_best_model(self):
        raise NotImplementedError()

    def _train_with_early_stopping(self, epochs_max, epochs_min=0, validation_every_n=None, stop_on_validation=False,
                                   validation_metric=None, lower_validations_allowed=None, evaluator_object=None,
                                   algorithm_name="Incremental_Training_Early_Stopping"):
        assert epochs_max >= 0, "{}: Number of epochs_max must be >= 0, passed was {}".format(algorithm_name, epochs_max)
        assert epochs_min >= 0, "{}: Number of epochs_min must be >= 0, passed was {}".format(algorithm_name, epochs_min)
        assert epochs_min <= epochs_max, "{}: epochs_min must be <= epochs_max".format(algorithm_name)
        validating = evaluator_object is not None
        assert (not validating
                or (not stop_on_validation and validation_every_n is not None and validation_metric is not None)
                or (stop_on_validation and validation_every_n is not None and validation_metric is not None
                    and lower_validations_allowed is not None)), \
            "{}: Inconsistent parameters passed, please check the supported uses".format(algorithm_name)

        verbose = getattr(self, "verbose", True)
        started = time.time()
        self.best_validation_metric = None
        self.epochs_best = 0
        worse_in_a_row = 0
        converged = False
        epoch = 0
        while epoch < epochs_max and not converged:
            self._run_epoch(epoch)
            if not validating:
                self.epochs_best = epoch
            elif (epoch + 1) % validation_every_n == 0:
                self._prepare_model_for_validation()
                results, results_string = evaluator_object.evaluateRecommender(self)
                first_cutoff = results[list(results.keys())[0]]
                value = first_cutoff[validation_metric]
                if verbose:
                    print("{}: {}".format(algorithm_name, results_string))
                assert np.isfinite(value), "{}: metric value is not a finite number, terminating!".format(
                    getattr(self, "RECOMMENDER_NAME", algorithm_name))
                if self.best_validation_metric is None or self.best_validation_metric < value:
                    self.best_validation_metric = value
                    self._update_best_model()
                    self.epochs_best = epoch + 1
                    worse_in_a_row = 0
                else:
                    worse_in_a_row += 1
                if stop_on_validation and worse_in_a_row >= lower_validations_allowed and epoch >= epochs_min:
                    converged = True
                    if verbose:
                        print("{}: Convergence reached! Terminating at epoch {}. Best value for '{}' at epoch {} is {:.4f}. "
                              "Elapsed time {:.2f} sec".format(algorithm_name, epoch + 1, validation_metric, self.epochs_best,
                                                               self.best_validation_metric, time.time() - started))
            if verbose:
                print("{}: Epoch {} of {}. Elapsed time {:.2f} sec".format(algorithm_name, epoch + 1, epochs_max,
                                                                            time.time() - started))
            epoch += 1
        if not validating:
            self._prepare_model_for_validation()
            self._update_best_model()
        if not converged and verbose:
            print("{}: Terminating at epoch {}. Elapsed time {:.2f} sec".format(algorithm_name, epoch, time.time() - started))
