"""Build-level check behind DESIGN 3.4: the scratch (spill) instructions of the IALS row kernel's k = 200 instance sit in the
prologue and in per-row blocks, never in a block that holds an MFMA.  Compiles ials.hip to assembly for gfx950 (hipcc
cross-compiles without a GPU, ~10 s) and reads the assembler's loop-depth comments (scripts/spill_audit.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_ials_row_kernel_spills_are_per_row_not_in_the_mfma_loops(tmp_path):
    from spill_audit import audit
    asm = str(tmp_path / "ials.s")
    src = os.path.join(ROOT, "recsys2019_deeplearning_evaluation_amd", "csrc", "ials.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                    src, "-o", asm], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    found = audit(asm, "ials_row_kernelILi12E", "v_mfma")
    assert len(found) == 1
    k = next(iter(found.values()))
    mfma_depths = set(k["hot"])
    assert mfma_depths and min(mfma_depths) >= 2, k["hot"]          # rows at depth 1, chunks / panels below
    scratch_depths = {d for (_, d) in k["scratch"]}
    assert not (scratch_depths & mfma_depths), (k["scratch"], k["hot"])
    assert max(scratch_depths, default=0) <= 1
    # the instances below 12 slots do not spill at all
    for slots in (1, 2, 4, 6, 8, 10):
        small = audit(asm, "ials_row_kernelILi%dE" % slots, "v_mfma")
        assert len(small) == 1 and not next(iter(small.values()))["scratch"], slots
