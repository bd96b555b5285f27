"""RCCL through ctypes (the transport of the sharded similarity build without PyTorch)."""
import threading

import numpy as np
import pytest

from recsys2019_deeplearning_evaluation_amd import rccl_direct


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_unique_id_rendezvous_over_tcp():
    """Rank 0 serves its 128-byte id; the other ranks may come up before or after it listens."""
    port = _free_port()
    payload = bytes(range(128))
    got = {}

    def run(rank):
        got[rank] = rccl_direct.exchange_unique_id(payload if rank == 0 else b"", rank, 4, "127.0.0.1", port, timeout=30)

    threads = [threading.Thread(target=run, args=(r,)) for r in (2, 3, 0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(40)
    assert all(got[r] == payload for r in range(4))
    assert rccl_direct.exchange_unique_id(payload, 0, 1, "127.0.0.1", port) == payload


def test_unique_id_travels_whole():
    """The id holds a socket address: NUL bytes in the middle (this is what kept RCCL from initialising through the binding)."""
    import ctypes as C
    uid = rccl_direct._UniqueId()
    raw = bytes([2, 0, 0x7f, 0, 0, 1] + [0] * 10 + list(range(1, 113)))
    C.memmove(C.byref(uid), raw, 128)
    assert rccl_direct._unique_id_bytes(uid) == raw


def test_librccl_exports_what_the_binding_uses():
    lib = rccl_direct._load_rccl()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString", "ncclCommCount",
                 "ncclCommUserRank", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd"):
        assert hasattr(lib, name)
    assert rccl_direct._UniqueId.__dict__ is not None and rccl_direct.NCCL_UNIQUE_ID_BYTES == 128


def test_rendezvous_ignores_strangers_and_answers_retries():
    """A connection without the job's hello must not use up a place; a second hello of a rank already answered (a retry after the
    rank gave up on its first connection) is answered again and does not use up a place either."""
    import socket
    port = _free_port()
    payload = bytes(range(128))
    got = {}

    def server():
        got[0] = rccl_direct.exchange_unique_id(payload, 0, 3, "127.0.0.1", port, timeout=30)

    t0 = threading.Thread(target=server)
    t0.start()
    for _ in range(200):                                     # a stranger, as soon as the port listens
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=1.0) as c:
                c.sendall(b"GET / HTTP/1.0\r\n\r\n" + b"x" * 8)
            break
        except ConnectionRefusedError:
            import time
            time.sleep(0.02)
    got[1] = rccl_direct.exchange_unique_id(b"", 1, 3, "127.0.0.1", port, timeout=30)
    with socket.create_connection(("127.0.0.1", port), timeout=5.0) as c:      # rank 1 again: answered again, rank 2's place stays free
        c.sendall(rccl_direct._HELLO_MAGIC + rccl_direct._job_nonce() + (1).to_bytes(4, "little"))
        c.settimeout(5.0)
        assert rccl_direct._recv_exact(c, 128) == payload
    got[2] = rccl_direct.exchange_unique_id(b"", 2, 3, "127.0.0.1", port, timeout=30)
    t0.join(40)
    assert got[0] == got[1] == got[2] == payload


_ONE_RANK = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
from recsys2019_deeplearning_evaluation_amd import _native as N, rccl_direct
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
comm = rccl_direct.RcclCommunicator(0, 1, "127.0.0.1", %d)
assert comm.count() == 1 and comm.user_rank() == 0
a, b = N.DeviceArray(1000), N.DeviceArray(1000)
src = np.arange(1000, dtype=np.int32)
N.check(N.load().mi355rec_device_memcpy(a.ptr, N.ptr(src), 4000, 1))
comm.all_gather_words(a.address(), b.address(), 1000)
assert (b.to_host() == src).all()
print("RCCL single-rank all-gather ran: ncclCommCount = %%d" %% comm.count())
# the gather of the sharded similarity build: one group of ncclSend / ncclRecv, the root's own piece as a send to itself
c = N.DeviceArray(1000)
comm.gather_words_async(a.address(250), c.address(), 750, 0)
comm.synchronize()
assert (c.to_host()[:750] == src[250:]).all()
print("RCCL single-rank gather (send / recv group) ran")
comm.close()
"""


@pytest.mark.gpu
def test_single_rank_rccl_communicator_all_gather(gpu):
    """One rank: ncclCommInitRank / ncclCommCount / ncclAllGather on raw device buffers of the library (the code path of every rank
    of an N-GPU run), in a process of its own -- one process per GPU is how the library is run, and RCCL's bootstrap inside a
    process that has already driven the device through a whole test suite is what failed in round 3 (64 s, "remote process exited
    or there was a network error"; scripts/rccl_probe.py: in a fresh process both ncclCommInitRank and ncclCommInitAll work on the
    1-GPU box).  A green line here means ncclAllGather moved bytes; a box on which RCCL cannot initialise at all reports XFAIL
    with RCCL's own message, never a pass."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NCCL_DEBUG="WARN")
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    r = subprocess.run([sys.executable, "-c", _ONE_RANK % (root, _free_port())], capture_output=True, text=True, timeout=180, env=env)
    if r.returncode != 0 and "ncclCommInitRank failed" in (r.stdout + r.stderr):
        pytest.xfail("RCCL cannot initialise on this box (single-rank ncclCommInitRank): %s" % " | ".join(
            ln[-160:] for ln in (r.stdout + r.stderr).strip().splitlines()[-6:]))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "RCCL single-rank all-gather ran: ncclCommCount = 1" in r.stdout
    assert "RCCL single-rank gather (send / recv group) ran" in r.stdout


@pytest.mark.gpu
def test_device_buffers_and_sharded_build_object_at_world_1(gpu):
    """Raw device buffers of the library round-trip; ShardedSimilarityBuild at world = 1 equals compute_slabs()."""
    from recsys2019_deeplearning_evaluation_amd import Compute_Similarity_MI355X, _native as N
    from recsys2019_deeplearning_evaluation_amd.sharding import ShardedSimilarityBuild
    from recsys2019_deeplearning_evaluation_amd.synthetic import named_urm
    a = N.DeviceArray(64)
    src = np.arange(64, dtype=np.int32)
    N.check(N.load().mi355rec_device_memcpy(a.ptr, N.ptr(src), 256, 1))
    np.testing.assert_array_equal(a.to_host(), src)
    X = named_urm("ml1m", "binary", scale=0.2)
    sim = Compute_Similarity_MI355X(X, topK=25, shrink=1)
    idx, val, _ = sim.compute_slabs()
    job = ShardedSimilarityBuild(sim)
    job.build()
    i2, v2 = job.download()
    np.testing.assert_array_equal(i2, idx)
    np.testing.assert_array_equal(v2, val)
    job.close()
