"""Can RCCL create a ONE-rank communicator on this box, and how?  Tries ncclCommInitRank and ncclCommInitAll under a few
environment settings, each in its own process under a timeout, and prints RCCL's own complaint for the ones that fail.
Run on the GPU box: python scripts/rccl_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r)
import numpy as np
from recsys2019_deeplearning_evaluation_amd import _native as N, rccl_direct as R
how = %r
N.check(N.load().mi355rec_device_synchronize())
lib = R._load_rccl()
comm = C.c_void_p()
if how == "init_all":
    lib.ncclCommInitAll.restype = C.c_int
    lib.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    dev = (C.c_int * 1)(0)
    rc = lib.ncclCommInitAll(C.byref(comm), 1, dev)
else:
    uid = R._UniqueId()
    rc = lib.ncclGetUniqueId(C.byref(uid))
    assert rc == 0, "ncclGetUniqueId %%d" %% rc
    rc = lib.ncclCommInitRank(C.byref(comm), 1, uid, 0)
print("init rc", rc, lib.ncclGetErrorString(rc).decode(), flush=True)
if rc == 0:
    a, b = N.DeviceArray(1000), N.DeviceArray(1000)
    src = np.arange(1000, dtype=np.int32)
    N.check(N.load().mi355rec_device_memcpy(a.ptr, N.ptr(src), 4000, 1))
    rc = lib.ncclAllGather(C.c_void_p(a.address()), C.c_void_p(b.address()), 1000, 2, comm, None)
    N.check(N.load().mi355rec_device_synchronize())
    print("all_gather rc", rc, "data ok", bool((b.to_host() == src).all()), flush=True)
'''
CONFIGS = [
    ("init_rank", {}),
    ("init_all", {}),
    ("init_rank", {"NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1"}),
    ("init_all", {"NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1"}),
    ("init_rank", {"RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0", "NCCL_DMABUF_ENABLE": "1"}),
    ("init_all", {"NCCL_TOPO_FILE": "", "NCCL_IGNORE_CPU_AFFINITY": "1", "NCCL_NET": "Socket"}),
]
for how, extra in CONFIGS:
    env = dict(os.environ)
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    env["NCCL_DEBUG"] = "WARN"
    env.update(extra)
    try:
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, how)], env=env, capture_output=True, text=True, timeout=60)
        out = (r.stdout + r.stderr).strip().splitlines()
        keep = [ln for ln in out if "init rc" in ln or "all_gather" in ln or "WARN" in ln or "rror" in ln][-6:]
        print("%-10s %-60s rc %d\n    %s" % (how, extra, r.returncode, "\n    ".join(k[:300] for k in keep)), flush=True)
    except subprocess.TimeoutExpired:
        print("%-10s %-60s TIMEOUT" % (how, extra), flush=True)
