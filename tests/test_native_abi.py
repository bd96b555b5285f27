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


def test_config_structs_have_the_layout_of_the_header(tmp_path):
    """The ctypes mirrors of the configuration structs against the C compiler's view of include/mi355rec.h: size and the offset
    of every field (a struct that grows in the header but not in _native.py would make the library read past the caller's
    memory)."""
    import ctypes as C
    import re
    import subprocess
    header = os.path.join(ROOT, "include", "mi355rec.h")
    N = _native
    structs = {"mi355rec_sim_config": N.SimConfig, "mi355rec_mf_config": N.MFConfig, "mi355rec_slim_config": N.SlimConfig,
               "mi355rec_stats": N.Stats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % header, 'int main(void) {']
    for c_name, cls in structs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (c_name, c_name))
        for field, _ in cls._fields_:
            lines.append('printf(" %s:%%zu", offsetof(%s, %s));' % (field, c_name, field))
        lines.append('printf("\\n");')
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        c_name, size, *fields = line.split()
        cls = structs[c_name]
        assert int(size) == C.sizeof(cls), "%s: header %s bytes, ctypes %d" % (c_name, size, C.sizeof(cls))
        for item in fields:
            name, off = item.split(":")
            assert getattr(cls, name).offset == int(off), "%s.%s: header offset %s, ctypes %d" % (c_name, name, off, getattr(cls, name).offset)
    # and the header has no field the mirrors lack
    text = open(header).read()
    for c_name, cls in structs.items():
        end = text.index("} %s;" % c_name)
        body = text[text.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        declared = [n.strip() for decl in body.split(";") if decl.strip() for n in decl.strip().split(None, 1)[1].split(",")]
        assert declared == [f for f, _ in cls._fields_], (c_name, declared)
