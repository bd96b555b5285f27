"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone) -- NumPy restatement of the reference's BM25 / TF-IDF re-weighting of the
interaction matrix (Base/IR_feature_weighting.py:13 okapi_BM_25, :55 TF_IDF; called from KNN/ItemKNNCFRecommender.py:40-48 and
KNN/UserKNNCFRecommender.py:40-48), pinned to reference outputs by tests/golden/feature_weighting.npz.  The product runs the
weighting as a device pre-pass of the similarity constructor (csrc/sim.hip: weighting_stats_kernel / weighting_apply_kernel);
this file is its checker.  "Documents" are the matrix rows."""
import numpy as np
import scipy.sparse as sps


def _document_frequencies(coo):
    # idf per column ("term"): log(N / (1 + number of rows storing that column))
    return np.log(float(coo.shape[0]) / (1 + np.bincount(coo.col)))


def okapi_BM_25(dataMatrix, K1=1.2, B=0.75):
    assert 0 < B < 1, "okapi_BM_25: B must be in (0,1)"
    assert K1 > 0, "okapi_BM_25: K1 must be > 0"
    assert np.all(np.isfinite(dataMatrix.data)), "okapi_BM_25: Data matrix contains non finite values"
    coo = sps.coo_matrix(dataMatrix)
    idf = _document_frequencies(coo)
    row_sums = np.ravel(coo.sum(axis=1))
    length_norm = (1.0 - B) + B * row_sums / row_sums.mean()
    denominator = K1 * length_norm[coo.row] + coo.data
    denominator[denominator == 0.0] += 1e-9
    coo.data = coo.data * (K1 + 1.0) / denominator * idf[coo.col]
    return coo.tocsr()


def TF_IDF(dataMatrix):
    assert np.all(np.isfinite(dataMatrix.data)), "TF_IDF: Data matrix contains non finite values."
    assert np.all(dataMatrix.data >= 0.0), "TF_IDF: Data matrix contains negative values, computing the square root is not possible."
    coo = sps.coo_matrix(dataMatrix)
    coo.data = np.sqrt(coo.data) * _document_frequencies(coo)[coo.col]
    return coo.tocsr()


def apply_feature_weighting(URM_train, feature_weighting, user_major):
    """What the KNN recommenders do to self.URM_train: `user_major` False = ItemKNN (weights URM.T, documents = items),
    True = UserKNN (same call in the reference: both weight URM.T and transpose back)."""
    if feature_weighting == "none":
        return URM_train
    weigh = okapi_BM_25 if feature_weighting == "BM25" else TF_IDF
    weighted = weigh(URM_train.astype(np.float32).T).T
    return sps.csr_matrix(weighted).astype(np.float32)
