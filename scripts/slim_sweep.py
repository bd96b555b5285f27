"""SLIM-BPR at BASELINE config 3 (ML-20M shape, adagrad): epoch time against the knobs of the two dataflow kernels -- rows owned in
LDS (dense store) and steps in flight (symmetric store) -- and, with MI355REC_SLIM_PROF=1, where the cycles of a step go.
Usage: slim_sweep.py [dense|symmetric|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys2019_deeplearning_evaluation_amd import SLIM_BPR_MI355X_Epoch
from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm

which = sys.argv[1] if len(sys.argv) > 1 else "both"
X = named_urm("ml20m", "binary")


def run(symmetric, label, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    ep = SLIM_BPR_MI355X_Epoch(X, symmetric=symmetric, topK=100, learning_rate=1e-4, sgd_mode="adagrad", random_seed=3)
    ep.epochIteration_Cython()
    os.environ.pop("MI355REC_SLIM_PROF", None)
    n = 4
    t = time.perf_counter()
    ep.epochIteration_Cython(n)
    dt = (time.perf_counter() - t) / n
    st = ep.stats()
    print("%-9s %-28s %8.3f ms per epoch (flow kernel %.3f ms) %6.2f M samples/s  owned rows / cold steps %s" % (
        "symmetric" if symmetric else "dense", label, dt * 1e3, st["kernel_ms"] / n, (X.shape[0] + 1) / dt / 1e6, ep.schedule_info()), flush=True)
    ep._dealloc()
    for k in env:
        os.environ.pop(k, None)


if which in ("dense", "both"):
    run(False, "default + phase clocks", MI355REC_SLIM_PROF=1)
    for owners in (0, 32, 64, 96, 128, 160, 192):
        run(False, "owners %d" % owners, MI355REC_SLIM_OWNERS=owners)
    for cus in (192, 128):
        run(False, "owners 64 on %d CUs" % cus, MI355REC_SLIM_OWNERS=64, MI355REC_SLIM_CUS=cus)
    for steps in (8, 48):
        run(False, "owners 128, rows with >= %d steps" % steps, MI355REC_SLIM_OWNER_MIN_STEPS=steps)
if which == "turn":
    run(False, "default + phase clocks", MI355REC_SLIM_PROF=1)
    run(False, "default")
    run(False, "owners 64", MI355REC_SLIM_OWNERS=64)
    run(True, "default + phase clocks", MI355REC_SLIM_PROF=1)
    run(True, "default")
if which in ("symmetric", "both"):
    run(True, "default + phase clocks", MI355REC_SLIM_PROF=1)
    for wgs in (8, 16, 32, 64, 128, 256):
        run(True, "%d workgroups (%d steps in flight)" % (wgs, wgs * 16), MI355REC_SLIM_SYM_WGS=wgs)
if which == "presched":
    # the next epoch's schedule behind the running kernel (default) against one thing after the other, one epoch per call and four
    for sym in (False, True):
        run(sym, "schedule ahead (default)")
        run(sym, "no schedule ahead", MI355REC_SLIM_NO_PRESCHED=1)
    run(True, "default + phase clocks", MI355REC_SLIM_PROF=1)
    for spare in (0, 8, 16, 64, 96):
        run(True, "%d compute units left free" % spare, MI355REC_SLIM_SYM_SPARE_CUS=spare)
    run(True, "32 free, 96 workgroups for long profiles", MI355REC_SLIM_SYM_LONG_WGS=96)
    ep = SLIM_BPR_MI355X_Epoch(X, symmetric=True, topK=100, learning_rate=1e-4, sgd_mode="adagrad", random_seed=3)
    ep.epochIteration_Cython()
    t = time.perf_counter()
    for _ in range(4):
        ep.epochIteration_Cython()
    print("symmetric, one epoch per call: %.3f ms per epoch" % ((time.perf_counter() - t) / 4 * 1e3))
if which == "nap":
    for sym in (False, True):
        for nap in (0, 1, 2, 3):
            run(sym, "nap %d" % nap, MI355REC_SLIM_NAP=nap)
