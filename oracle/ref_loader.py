"""TEST INFRASTRUCTURE ONLY -- loads the compiled reference modules built by oracle/build_ref.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BUILD = os.path.join(HERE, "_ref")
REF_SHIM = os.path.join(HERE, "ref_shim")
REFERENCE_ROOT = os.environ.get("RECSYS_REFERENCE_ROOT", "/root/reference")

_cache = {}


def _prepare():
    # NumPy 2 removed the aliases the reference still uses at call time
    # (MatrixFactorization_Cython_Epoch.pyx:709-713, BaseRecommender.py:30).
    for name, typ in (("int", int), ("float", float), ("bool", bool)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    paths = [REF_BUILD]
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "Base")):
        paths.append(REFERENCE_ROOT)
    paths.append(REF_SHIM)
    for p in paths:
        if p not in sys.path:
            sys.path.append(p)


def reference_tree_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "Base"))


def load(name):
    """name in {'mf', 'slim', 'sim'}; returns the reference cdef class or None if not built."""
    if name in _cache:
        return _cache[name]
    _prepare()
    mod_path, cls = {
        "mf": ("MatrixFactorization.Cython.MatrixFactorization_Cython_Epoch", "MatrixFactorization_Cython_Epoch"),
        "slim": ("SLIM_BPR.Cython.SLIM_BPR_Cython_Epoch", "SLIM_BPR_Cython_Epoch"),
        "sim": ("Base.Similarity.Cython.Compute_Similarity_Cython", "Compute_Similarity_Cython"),
    }[name]
    try:
        obj = getattr(importlib.import_module(mod_path), cls)
    except ImportError:
        obj = None
    _cache[name] = obj
    return obj


def load_python_reference(dotted, attr):
    """Import a pure-Python reference object (only possible where /root/reference exists)."""
    if not reference_tree_available():
        return None
    _prepare()
    return getattr(importlib.import_module(dotted), attr)
