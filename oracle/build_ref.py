#!/usr/bin/env python3
"""Recipe that compiles the REFERENCE's own Cython hot-path sources into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path may import from oracle/.

The three hot-path kernels of MaurizioFD/RecSys2019_DeepLearning_Evaluation are Cython
(`.pyx` -> C -> `.so`).  This script compiles them *from where they lie* under /root/reference
(no reference source is copied into this repository): the generated C goes to a temporary
directory, only the resulting extension modules (`.so`) land in `oracle/_ref/`, which is listed in
.gitignore (so no derived artefact enters history) but NOT in .gpurunignore (so the built oracle
travels to the GPU box, where /root/reference does not exist).

Sources compiled (reference file -> module placed under oracle/_ref/ with the same package path):
  MatrixFactorization/Cython/MatrixFactorization_Cython_Epoch.pyx   (BPR-MF, FunkSVD, AsySVD epochs)
  SLIM_BPR/Cython/SLIM_BPR_Cython_Epoch.pyx                         (SLIM-BPR epoch; needs legacy_implicit_noexcept)
  Base/Similarity/Cython/Compute_Similarity_Cython.pyx              (similarity build)

Equivalent of the reference's own CythonCompiler/compile_script.py:40-49 (Extension(..., '-O2',
numpy include)); we do not run that script because it writes next to the (read-only) sources.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

import numpy

REFERENCE_ROOT = os.environ.get("RECSYS_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_ROOT = os.path.join(HERE, "_ref")

PYX = [
    ("MatrixFactorization/Cython", "MatrixFactorization_Cython_Epoch", []),
    ("SLIM_BPR/Cython", "SLIM_BPR_Cython_Epoch", ["-X", "legacy_implicit_noexcept=True"]),
    ("Base/Similarity/Cython", "Compute_Similarity_Cython", []),
]


def reference_available():
    return all(os.path.isfile(os.path.join(REFERENCE_ROOT, d, n + ".pyx")) for d, n, _ in PYX)


def built():
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return all(os.path.isfile(os.path.join(OUT_ROOT, d, n + suffix)) for d, n, _ in PYX)


def build(force=False, verbose=True):
    """Build oracle/_ref/*.so.  Returns True if the modules are present afterwards."""
    if built() and not force:
        return True
    if not reference_available():
        if verbose:
            print("[oracle/build_ref] reference sources not present at %s; nothing to build" % REFERENCE_ROOT)
        return built()

    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    py_inc = sysconfig.get_paths()["include"]
    np_inc = numpy.get_include()
    tmp = tempfile.mkdtemp(prefix="recsys_ref_build_")
    try:
        for rel_dir, name, cy_flags in PYX:
            src = os.path.join(REFERENCE_ROOT, rel_dir, name + ".pyx")
            c_file = os.path.join(tmp, name + ".c")
            out_dir = os.path.join(OUT_ROOT, rel_dir)
            os.makedirs(out_dir, exist_ok=True)
            so_file = os.path.join(out_dir, name + suffix)
            cmd = [sys.executable, "-m", "cython", "-3"] + cy_flags + ["-o", c_file, src]
            if verbose:
                print("[oracle/build_ref]", " ".join(cmd))
            subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            cmd = ["gcc", "-O2", "-fPIC", "-shared", "-w", "-fno-strict-aliasing",
                   "-I", py_inc, "-I", np_inc, c_file, "-o", so_file, "-lm"]
            if verbose:
                print("[oracle/build_ref]", " ".join(cmd))
            subprocess.run(cmd, check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return built()


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref built:", ok)
    sys.exit(0 if ok else 1)
