"""The C-ABI library loads and exports every symbol include/mi355rec.h declares (no compute calls, no GPU)."""
import os
import re

import pytest

from recsys2019_deeplearning_evaluation_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355rec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355rec_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_declares_expected_groups():
    names = declared_symbols()
    for group in ("sim", "mf", "slim", "ials"):
        assert any(n.startswith("mi355rec_%s_create" % group) for n in names), group
    assert len(names) >= 30


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    for name in declared_symbols():
        assert hasattr(lib, name), "libmi355rec.so does not export %s" % name


def test_binding_covers_header_exactly():
    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_device_count_does_not_need_a_gpu():
    assert _native.device_count() >= 0


def test_no_cpu_fallback_without_device():
    """Without a device a create call must raise -- never compute on the host."""
    if _native.device_count() > 0:
        pytest.skip("a device is present")
    import numpy as np
    import scipy.sparse as sps
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X, MatrixFactorization_MI355X_Epoch
    X = sps.random(20, 10, 0.3, format="csr", dtype=np.float32, random_state=0)
    with pytest.raises(_native.NativeLibraryError):
        Compute_Similarity_MI355X(X, topK=3)
    with pytest.raises(_native.NativeLibraryError):
        MatrixFactorization_MI355X_Epoch(X, n_factors=4, algorithm_name="MF_BPR", batch_size=4, random_seed=1)


def test_bad_enum_arguments_raise_value_error_before_touching_the_device():
    import numpy as np
    import scipy.sparse as sps
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X, MatrixFactorization_MI355X_Epoch
    X = sps.random(20, 10, 0.3, format="csr", dtype=np.float32, random_state=0)
    with pytest.raises(ValueError):
        Compute_Similarity_MI355X(X, similarity="nope")
    with pytest.raises(ValueError):
        MatrixFactorization_MI355X_Epoch(X, algorithm_name="MF_BPR", sgd_mode="nope")
    with pytest.raises(ValueError):
        MatrixFactorization_MI355X_Epoch(X, algorithm_name="nope")
