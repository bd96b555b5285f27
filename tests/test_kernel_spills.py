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
    found = audit(asm, "ials_row_kernelILi12ELi0ELi512E", "v_mfma")          # the one-kernel epoch at k = 200: <SLOTS 12, STAGE 0, 512 threads>
    assert len(found) == 1
    k = next(iter(found.values()))
    mfma_depths = set(k["hot"])
    assert mfma_depths and min(mfma_depths) >= 2, k["hot"]          # rows at depth 1, chunks / panels below
    scratch_depths = {d for (_, d) in k["scratch"]}
    assert not (scratch_depths & mfma_depths), (k["scratch"], k["hot"])
    assert max(scratch_depths, default=0) <= 1
    # the instances below 12 slots do not spill at all
    for slots in (1, 2, 4, 6, 8, 10):
        small = audit(asm, "ials_row_kernelILi%dELi0ELi512E" % slots, "v_mfma")
        assert len(small) == 1 and not next(iter(small.values()))["scratch"], slots
    # round 5, two-stage epochs.  The Gramian stage (1024 threads = 128 registers per lane, 6 tile slots at k = 200, 32 profile rows
    # staged per round) keeps its few spilled words at row level, never in a block with an MFMA; the solve stage must fit 128 registers (two workgroups per CU is the point of it) and keeps what it spills out of the
    # blocks that hold its MFMAs' operand loads: at most a handful of reloads per panel, none deeper.
    gram = audit(asm, "ials_row_kernelILi6ELi1ELi1024E", "v_mfma")
    assert len(gram) == 1
    gk = next(iter(gram.values()))
    assert max((d for (_, d) in gk["scratch"]), default=0) <= 1 and min(gk["hot"]) >= 2, (gk["scratch"], gk["hot"])
    assert _resource(asm, "ials_row_kernelILi6ELi1ELi1024E", "NumVgprs") <= 128
    solve = audit(asm, "ials_solve_kernelILi13E", "v_mfma")
    assert len(solve) == 1
    sk = next(iter(solve.values()))
    assert _resource(asm, "ials_solve_kernelILi13E", "NumVgprs") <= 128
    assert _resource(asm, "ials_solve_kernelILi13E", "Occupancy") >= 4           # 4 wavefronts per SIMD = two 512-thread workgroups per CU
    assert max((d for (_, d) in sk["scratch"]), default=0) <= 2 and sum(n for (kind, d), n in sk["scratch"].items() if d == 2) <= 32, sk["scratch"]


def _kernel_text(asm_path, mangled_fragment):
    """The instructions of the one kernel whose mangled name holds the fragment (label .. s_endpgm), comments dropped."""
    lines = open(asm_path).read().split("\n")
    starts = [n for n, ln in enumerate(lines) if ln.split(";")[0].strip().endswith(":") and mangled_fragment in ln.split(";")[0]
              and not ln.startswith(("\t", " ", "."))]
    assert len(starts) == 1, (mangled_fragment, len(starts))
    body = []
    for ln in lines[starts[0] + 1:]:
        text = ln.split(";")[0].strip()
        if text:
            body.append(text)
        if text.startswith("s_endpgm"):
            break
    return body


def _resource(asm_path, mangled_fragment, key):
    """`; key: value` from the kernel's resource comment block (VGPRs, ScratchSize, Occupancy ...)."""
    text = open(asm_path).read()
    at = text.index(".amdhsa_kernel", text.index(mangled_fragment))
    import re
    m = re.search(r";\s*%s:\s*(\d+)" % re.escape(key), text[at:at + 20000])
    assert m, key
    return int(m.group(1))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_mini_batch_kernel_prologue_and_plain_sgd_instances(tmp_path):
    """What the round-4 speed-up of the mini-batch chain rests on, checked on the assembly (no GPU): the plain-sgd instances carry no
    optimiser arithmetic (no square root) and no scratch memory; every kernel argument the prologue needs is
    loaded before the header is asked for (at most two waits on scalar loads in front of the first vector load), and the header, a pair
    task's slot records and -- FunkSVD -- the batch index and the global-bias ring entry are all requested before the first wait on
    a vector load."""
    asm = str(tmp_path / "mf.s")
    src = os.path.join(ROOT, "recsys2019_deeplearning_evaluation_amd", "csrc", "mf.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                    src, "-o", asm], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for algo, min_loads in ((0, 3), (1, 5)):             # BPR: header (2 x 16 B) + slot record; FunkSVD: + batch index, ring state, ring terms
        name = "15mf_batch_kernelILi%dEfLi4ELi32ELi1ELb1EE" % algo
        body = _kernel_text(asm, name)
        first_vload = next(n for n, t in enumerate(body) if t.startswith("global_load"))
        first_vwait = next(n for n, t in enumerate(body) if t.startswith("s_waitcnt vmcnt"))
        scalar_waits = sum(1 for t in body[:first_vload] if t.startswith("s_waitcnt lgkmcnt"))
        assert scalar_waits <= 3, (algo, scalar_waits)                      # (one of them belongs to the optional clock stamp)
        loads_before_wait = sum(1 for t in body[first_vload:first_vwait] if t.startswith("global_load"))
        assert loads_before_wait >= min_loads, (algo, loads_before_wait)
        assert _resource(asm, name, "ScratchSize") == 0
    plain = _kernel_text(asm, "15mf_batch_kernelILi0EfLi4ELi32ELi1ELb1EE")
    general = _kernel_text(asm, "15mf_batch_kernelILi0EfLi4ELi32ELi1ELb0EE")
    assert not any(t.startswith("v_sqrt_f32") for t in plain)           # (the sigmoid keeps its one division)
    assert any(t.startswith("v_sqrt_f32") for t in general)
    assert len(plain) * 2 < len(general)
    # the replica-batched plain-sgd instance: no scratch, at most 80 registers (6 wavefronts per SIMD) would be wasted on spills
    assert _resource(asm, "21mf_group_batch_kernelILi0EfLi4ELi32ELi1ELb1EE", "ScratchSize") == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_slim_dataflow_kernels_fit_their_launch_shape(tmp_path):
    """1024-thread workgroups leave 128 vector registers per lane: the SLIM dataflow kernels must fit without scratch memory (a spilled
    granule would put a memory round trip inside a chain link)."""
    asm = str(tmp_path / "slim.s")
    src = os.path.join(ROOT, "recsys2019_deeplearning_evaluation_amd", "csrc", "slim.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                    src, "-o", asm], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for name in ("slim_sym_flow_kernel", "slim_dense_flow_kernelIdE", "slim_dense_flow_kernelIfE"):
        assert _resource(asm, name, "ScratchSize") == 0, name
        assert _resource(asm, name, "NumVgprs") <= 128, name
