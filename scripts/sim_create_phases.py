"""Where the similarity constructor's wall time goes at ML-20M shape: the library's own phase clocks (MI355REC_SIM_CREATE_PHASES=1, each
phase drained), then -- without them -- the Python front-end's steps, the ctypes call and the first synchronisation, for host arrays
(PCIe upload) and for a resident URM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from bench import load_urm, TOPK
from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X, ResidentURM, _native as N
urm = load_urm("ml20m")
res = ResidentURM(urm)
Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0).close()          # warm-up (code objects, block cache)
os.environ["MI355REC_SIM_CREATE_PHASES"] = "1"
for label, kw in (("host arrays (PCIe upload)", {}), ("resident URM", {"resident": res})):
    print("----", label, flush=True)
    s = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, **kw)
    s.close()
del os.environ["MI355REC_SIM_CREATE_PHASES"]
lib = N.load()
real_create, real_resident = lib.mi355rec_sim_create, lib.mi355rec_sim_create_resident
for label, kw in (("host arrays (PCIe upload)", {}), ("resident URM", {"resident": res})):
    for rep in range(3):
        spent = {}
        def timed(fn, name):
            def call(*a):
                t = time.perf_counter(); r = fn(*a); spent[name] = time.perf_counter() - t; return r
            return call
        lib.mi355rec_sim_create, lib.mi355rec_sim_create_resident = timed(real_create, "ctypes call"), timed(real_resident, "ctypes call")
        t0 = time.perf_counter()
        s = Compute_Similarity_MI355X(urm, topK=TOPK, shrink=0, **kw)
        t1 = time.perf_counter()
        s.synchronize()
        t2 = time.perf_counter()
        lib.mi355rec_sim_create, lib.mi355rec_sim_create_resident = real_create, real_resident
        print("%-26s rep %d: constructor %.3f ms = Python around the call %.3f + library call %.3f; synchronize %.3f ms" %
              (label, rep, (t1 - t0) * 1e3, (t1 - t0 - spent["ctypes call"]) * 1e3, spent["ctypes call"] * 1e3, (t2 - t1) * 1e3), flush=True)
        t3 = time.perf_counter(); s.close(); print("    close %.3f ms" % ((time.perf_counter() - t3) * 1e3))
