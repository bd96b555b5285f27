"""RCCL through ctypes: the collective of the sharded similarity build without PyTorch.

One process per GPU.  Rank 0 creates the `ncclUniqueId` and hands it to the other ranks over a plain TCP socket
(MASTER_ADDR : MI355REC_RCCL_PORT, or MASTER_PORT + 1 -- the launcher's own store keeps MASTER_PORT; clients identify
themselves with a job nonce and their rank, and rank 0 answers each rank once); every rank then calls
`ncclCommInitRank`.  `all_gather` runs on the null stream and returns after the device has finished (the callers are
blocking library calls anyway).  Only the three entry points the build needs are bound.
"""
import ctypes as C
import os
import socket
import time

from . import _native as N

NCCL_UNIQUE_ID_BYTES = 128
NCCL_INT32 = 2          # ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ...


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def _unique_id_bytes(uid):
    return C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES)


def _load_rccl():
    last = None
    for name in (os.environ.get("MI355REC_RCCL_LIBRARY"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"):
        if not name:
            continue
        try:
            lib = C.CDLL(name)
            break
        except OSError as exc:
            last = exc
    else:
        raise N.NativeLibraryError("librccl not found: %s" % last)
    lib.ncclGetUniqueId.restype = C.c_int
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.restype = C.c_int
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllGather.restype = C.c_int
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    for name in ("ncclSend", "ncclRecv"):           # (buffer, count, datatype, peer, comm, stream)
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for name in ("ncclGroupStart", "ncclGroupEnd"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = []
    lib.ncclCommCount.restype = C.c_int
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ncclCommUserRank.restype = C.c_int
    lib.ncclCommUserRank.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ncclCommDestroy.restype = C.c_int
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    return lib


_HELLO_MAGIC = b"MI355RCL"


def _job_nonce():
    """8 bytes every rank of ONE launch agrees on (torchrun exports TORCHELASTIC_RUN_ID; MI355REC_RCCL_NONCE overrides)."""
    import hashlib
    tag = os.environ.get("MI355REC_RCCL_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or "mi355rec"
    tag += ":" + os.environ.get("MASTER_ADDR", "") + ":" + os.environ.get("MASTER_PORT", "")
    return hashlib.sha256(tag.encode()).digest()[:8]


def _recv_exact(conn, need):
    chunks = []
    while need > 0:
        part = conn.recv(need)
        if not part:
            raise ConnectionError("peer closed the connection early")
        chunks.append(part)
        need -= len(part)
    return b"".join(chunks)


def exchange_unique_id(payload, rank, world, address, port, timeout=120.0):
    """Rank 0 serves `payload` (bytes) to the world - 1 other ranks; they return what they received.

    Every client opens with a 20-byte hello (magic, job nonce, rank): rank 0 answers EVERY valid hello of this job with the payload
    and stops once world - 1 distinct ranks have been answered.  A stray connection (another job on the same port, a port scanner)
    uses up no place; a rank that gave up on its first connection (rank 0 was busy with a slow peer) and comes back is simply
    answered again -- it cannot be locked out by its own retry.  Rank 0 waits 2 s for a hello, the clients 30 s for the answer."""
    if world == 1:
        return payload
    nonce = _job_nonce()
    if rank == 0:
        served = set()
        deadline = time.time() + timeout
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((address, port))
            srv.listen(max(world, 8))
            while len(served) < world - 1:
                srv.settimeout(max(0.05, deadline - time.time()))
                conn, _ = srv.accept()          # socket.timeout propagates once the deadline has passed
                with conn:
                    try:
                        conn.settimeout(2.0)
                        hello = _recv_exact(conn, 20)
                        peer = int.from_bytes(hello[16:20], "little")
                        if hello[:8] != _HELLO_MAGIC or hello[8:16] != nonce or not (0 < peer < world):
                            continue
                        conn.sendall(payload)
                        served.add(peer)
                    except (ConnectionError, socket.timeout, OSError):
                        continue
        return payload
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((address, port), timeout=30.0) as conn:
                conn.sendall(_HELLO_MAGIC + nonce + int(rank).to_bytes(4, "little"))
                return _recv_exact(conn, len(payload) if payload else NCCL_UNIQUE_ID_BYTES)
        except (ConnectionRefusedError, socket.timeout, ConnectionError):
            if time.time() > deadline:
                raise
            time.sleep(0.05)


class RcclCommunicator:
    def __init__(self, rank=None, world=None, address=None, port=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        address = address or os.environ.get("MASTER_ADDR", "127.0.0.1")
        # the rendezvous port: explicit argument, MI355REC_RCCL_PORT, or MASTER_PORT + 1 (the launcher's store keeps MASTER_PORT)
        if port is None:
            port = os.environ.get("MI355REC_RCCL_PORT") or int(os.environ.get("MASTER_PORT", "29500")) + 1
        port = int(port)
        self._lib = _load_rccl()
        N.check(N.load().mi355rec_device_synchronize())          # binds this process to its device (set_device) first
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self._lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        # (all 128 bytes: reading the c_char array field itself would stop at the first NUL of the socket address inside)
        raw = exchange_unique_id(_unique_id_bytes(uid) if self.rank == 0 else b"", self.rank, self.world, address, port)
        if len(raw) != NCCL_UNIQUE_ID_BYTES:
            raise N.NativeLibraryError("RCCL rendezvous: got %d bytes of unique id, expected %d" % (len(raw), NCCL_UNIQUE_ID_BYTES))
        C.memmove(C.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self._comm = C.c_void_p()
        self._check(self._lib.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def count(self):
        """ncclCommCount: the number of ranks RCCL itself sees in this communicator (must equal WORLD_SIZE)."""
        n = C.c_int(-1)
        self._check(self._lib.ncclCommCount(self._comm, C.byref(n)), "ncclCommCount")
        return n.value

    def user_rank(self):
        r = C.c_int(-1)
        self._check(self._lib.ncclCommUserRank(self._comm, C.byref(r)), "ncclCommUserRank")
        return r.value

    def _check(self, rc, what):
        if rc != 0:
            raise N.NativeLibraryError("%s failed: %s" % (what, self._lib.ncclGetErrorString(rc).decode()))

    def all_gather_words(self, send_address, recv_address, n_words):
        """recv[r * n_words : (r + 1) * n_words] = rank r's send[0 : n_words] (int32 words), on every rank.  Blocking."""
        self._check(self._lib.ncclAllGather(C.c_void_p(send_address), C.c_void_p(recv_address), n_words, NCCL_INT32, self._comm, None),
                    "ncclAllGather")
        N.check(N.load().mi355rec_device_synchronize())

    def all_gather_words_async(self, send_address, recv_address, n_words):
        """The same, enqueued only (null stream): the library's own non-blocking streams keep running; synchronize() before the
        receive buffer is read."""
        self._check(self._lib.ncclAllGather(C.c_void_p(send_address), C.c_void_p(recv_address), n_words, NCCL_INT32, self._comm, None),
                    "ncclAllGather")

    def gather_words_async(self, send_address, recv_address, n_words, root):
        """recv[r * n_words : (r + 1) * n_words] on rank `root` = rank r's send[0 : n_words]; nothing is written elsewhere (recv_address is
        ignored there).  One group of point-to-point transfers: every rank sends to the root over its own xGMI link, the root posts
        `world` receives -- its own piece included (a send to oneself inside the group of its receive, the pattern of an all-to-all):
        nothing here blocks the host.  Enqueued only (null stream), like all_gather_words_async."""
        self._check(self._lib.ncclGroupStart(), "ncclGroupStart")
        try:
            if self.rank == root:
                for r in range(self.world):
                    self._check(self._lib.ncclRecv(C.c_void_p(recv_address + 4 * r * n_words), n_words, NCCL_INT32, r, self._comm, None), "ncclRecv")
            self._check(self._lib.ncclSend(C.c_void_p(send_address), n_words, NCCL_INT32, root, self._comm, None), "ncclSend")
        finally:
            self._check(self._lib.ncclGroupEnd(), "ncclGroupEnd")

    def synchronize(self):
        N.check(N.load().mi355rec_device_synchronize())

    def close(self):
        if getattr(self, "_comm", None) is not None and self._comm.value:
            self._lib.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
