"""ctypes binding of libmi355rec.so (C ABI declared in include/mi355rec.h).

There is NO CPU fallback: if the shared library is missing, or no gfx950 device is visible when a handle is
created, the call raises.  The library is only dlopen()ed on first use and the HIP context is only created
by the first *_create call, so importing this package before a fork (the reference's hyper-parameter search
forks one worker per configuration, ParameterTuning/run_parameter_search.py:498) is safe.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MI355REC_LIB: another build of the same library, for measurements of compile-time variants -- scripts/sim_depth_sweep.sh)
LIB_PATH = os.environ.get("MI355REC_LIB") or os.path.join(_HERE, "libmi355rec.so")

E_INVALID, E_HIP, E_NO_DEVICE, E_UNSUPPORTED, E_NUMERIC = -1, -2, -3, -4, -5

SIMILARITY_CODES = {"cosine": 0, "adjusted": 1, "asymmetric": 2, "pearson": 3, "jaccard": 4, "tanimoto": 4,
                    "dice": 5, "tversky": 6, "euclidean": 7}
EUCLIDEAN_MODE_CODES = {"lin": 0, "log": 1, "exp": 2}
FEATURE_WEIGHTING_CODES = {"none": 0, "BM25": 1, "TF-IDF": 2}
SGD_MODE_CODES = {"sgd": 0, "adagrad": 1, "rmsprop": 2, "adam": 3}
ALGORITHM_CODES = {"MF_BPR": 0, "FUNK_SVD": 1, "ASY_SVD": 2}
PRECISION_CODES = {"fp32": 0, "fp64": 1}


class NativeLibraryError(RuntimeError):
    """libmi355rec.so is missing, or the HIP runtime / device failed."""


class Stats(C.Structure):
    _fields_ = [("call_ms", C.c_double), ("kernel_ms", C.c_double), ("n_launches", C.c_int64), ("n_timed", C.c_int64),
                ("n_units", C.c_int64), ("algorithmic_bytes", C.c_double), ("algorithmic_flops", C.c_double),
                ("loss", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class SimConfig(C.Structure):
    _fields_ = [("topK", C.c_int32), ("shrink", C.c_int32), ("normalize", C.c_int32), ("similarity", C.c_int32),
                ("asymmetric_alpha", C.c_float), ("tversky_alpha", C.c_float), ("tversky_beta", C.c_float),
                ("unit_column_side", C.c_int32), ("normalize_avg_row", C.c_int32), ("euclidean_mode", C.c_int32),
                ("feature_weighting", C.c_int32), ("weighting_documents", C.c_int32), ("bm25_k1", C.c_float), ("bm25_b", C.c_float),
                ("norm_sum_order", C.c_int32), ("reserved", C.c_int32)]


class MFConfig(C.Structure):
    _fields_ = [("algorithm", C.c_int32), ("n_factors", C.c_int32), ("batch_size", C.c_int32),
                ("use_bias", C.c_int32), ("sgd_mode", C.c_int32), ("learning_rate", C.c_double),
                ("user_reg", C.c_double), ("item_reg", C.c_double), ("bias_reg", C.c_double),
                ("positive_reg", C.c_double), ("negative_reg", C.c_double),
                ("negative_interactions_quota", C.c_double),
                ("gamma", C.c_double), ("beta_1", C.c_double), ("beta_2", C.c_double),
                ("random_seed", C.c_uint64), ("precision", C.c_int32), ("reserved", C.c_int32)]


class SlimConfig(C.Structure):
    _fields_ = [("symmetric", C.c_int32), ("sgd_mode", C.c_int32), ("learning_rate", C.c_double),
                ("li_reg", C.c_double), ("lj_reg", C.c_double),
                ("gamma", C.c_double), ("beta_1", C.c_double), ("beta_2", C.c_double), ("random_seed", C.c_uint64),
                ("precision", C.c_int32), ("train_with_sparse_weights", C.c_int32), ("topK", C.c_int32), ("reserved", C.c_int32)]


_vp = C.c_void_p
_i32, _i64, _f32, _f64 = C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); must list every symbol declared in include/mi355rec.h
SIGNATURES = {
    "mi355rec_last_error": (C.c_char_p, []),
    "mi355rec_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mi355rec_set_device": (C.c_int, [C.c_int]),
    "mi355rec_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "mi355rec_device_malloc": (C.c_int, [C.POINTER(_vp), C.c_uint64]),
    "mi355rec_device_free": (C.c_int, [_vp]),
    "mi355rec_device_memcpy": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int]),
    "mi355rec_device_synchronize": (C.c_int, []),
    "mi355rec_device_trim": (C.c_int, [C.POINTER(C.c_uint64)]),
    "mi355rec_sim_create": (C.c_int, [C.POINTER(_vp), C.POINTER(SimConfig), _i32, _i32, _vp, _vp, _vp, _vp]),
    "mi355rec_sim_create_resident": (C.c_int, [C.POINTER(_vp), C.POINTER(SimConfig), _i32, _i32, _vp, _vp, _vp, _vp]),
    "mi355rec_sim_get_weighted_values": (C.c_int, [_vp, _vp]),
    "mi355rec_sim_compute_part_device": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "mi355rec_sim_compute_part_chunk_device": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mi355rec_sim_pack_slab_device": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "mi355rec_sim_unpack_slab_device": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "mi355rec_sim_part_columns": (C.c_int, [_vp, _i32, _i32, _vp, C.POINTER(_i32)]),
    "mi355rec_sim_compute": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "mi355rec_sim_compute_device": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "mi355rec_sim_compute_dense": (C.c_int, [_vp, _i32, _i32, _vp, _i64]),
    "mi355rec_sim_compute_csr": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "mi355rec_sim_column_costs": (C.c_int, [_vp, _vp]),
    "mi355rec_sim_schedule_info": (C.c_int, [_vp, _vp, _vp, _vp]),
    "mi355rec_sim_accumulator_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_f64)]),
    "mi355rec_sim_selection_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "mi355rec_lds_atomic_rate": (C.c_int, [C.POINTER(_f64)]),
    "mi355rec_sim_sync": (C.c_int, [_vp]),
    "mi355rec_sim_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mi355rec_sim_destroy": (None, [_vp]),
    "mi355rec_mf_create": (C.c_int, [C.POINTER(_vp), C.POINTER(MFConfig), _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mi355rec_mf_run_epochs": (C.c_int, [_vp, _i32]),
    "mi355rec_mf_run_samples": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "mi355rec_mf_shard_begin_epoch": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(_i32)]),
    "mi355rec_mf_shard_batch": (C.c_int, [_vp, _i32]),
    "mi355rec_mf_shard_merge": (C.c_int, [_vp, _i32]),
    "mi355rec_mf_shard_end_epoch": (C.c_int, [_vp]),
    "mi355rec_mf_get_factors": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mi355rec_mf_get_factors_f64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mi355rec_mf_get_last_samples": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "mi355rec_mf_set_profiling": (C.c_int, [_vp, _i32]),
    "mi355rec_mf_get_phase_ticks": (C.c_int, [_vp, _vp, _i64, C.POINTER(_i64)]),
    "mi355rec_mf_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mi355rec_mf_destroy": (None, [_vp]),
    "mi355rec_mf_group_create": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), _i32]),
    "mi355rec_mf_group_run_epochs": (C.c_int, [_vp, _i32]),
    "mi355rec_mf_group_set_profiling": (C.c_int, [_vp, _i32]),
    "mi355rec_mf_group_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mi355rec_mf_group_destroy": (None, [_vp]),
    "mi355rec_slim_create": (C.c_int, [C.POINTER(_vp), C.POINTER(SlimConfig), _i32, _i32, _vp, _vp]),
    "mi355rec_slim_run_epochs": (C.c_int, [_vp, _i32]),
    "mi355rec_slim_run_samples": (C.c_int, [_vp, _vp, _vp, _vp, _i64]),
    "mi355rec_slim_get_last_samples": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "mi355rec_slim_get_S_topk": (C.c_int, [_vp, _i32, _vp, _vp]),
    "mi355rec_slim_get_W_csr": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int64)]),
    "mi355rec_slim_get_S_sparse": (C.c_int, [_vp, _vp, _vp]),
    "mi355rec_slim_get_S_dense": (C.c_int, [_vp, _vp]),
    "mi355rec_slim_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mi355rec_slim_schedule_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "mi355rec_slim_destroy": (None, [_vp]),
    "mi355rec_ials_create": (C.c_int, [C.POINTER(_vp), _i32, _i32, _i32, _f64, _vp, _vp, _vp, _vp, _vp]),
    "mi355rec_ials_run_epochs": (C.c_int, [_vp, _i32]),
    "mi355rec_ials_user_half": (C.c_int, [_vp, _i32, _i32]),
    "mi355rec_ials_item_half": (C.c_int, [_vp, _i32, _i32]),
    "mi355rec_ials_device_factors": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "mi355rec_ials_sync": (C.c_int, [_vp]),
    "mi355rec_ials_get_factors": (C.c_int, [_vp, _vp, _vp]),
    "mi355rec_ials_schedule_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "mi355rec_ials_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mi355rec_ials_destroy": (None, [_vp]),
    "mi355rec_scorer_create": (C.c_int, [C.POINTER(_vp), _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _f32, _vp, _vp]),
    "mi355rec_scorer_update": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32]),
    "mi355rec_scorer_recommend": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mi355rec_scorer_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mi355rec_scorer_destroy": (None, [_vp]),
    "mi355rec_spscorer_create": (C.c_int, [C.POINTER(_vp), _i32, _i32, _i32] + [_vp] * 8),
    "mi355rec_spscorer_recommend": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mi355rec_spscorer_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "mi355rec_spscorer_destroy": (None, [_vp]),
}

_lib = None


def load():
    """dlopen libmi355rec.so (built by __graft_entry__.build() / `make -C csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header and library out of sync
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc):
    """Map a C-ABI return code to the exception the reference would raise at the same point."""
    if rc == 0:
        return
    msg = load().mi355rec_last_error().decode("utf-8", "replace")
    if rc == E_INVALID:
        raise ValueError(msg)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == E_NUMERIC:
        raise FloatingPointError(msg)
    raise NativeLibraryError(msg)


def ptr(array):
    """Host pointer of a C-contiguous ndarray (or None)."""
    if array is None:
        return None
    assert array.flags["C_CONTIGUOUS"]
    return array.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    check(load().mi355rec_device_count(C.byref(n)))
    return n.value


def set_device(index):
    check(load().mi355rec_set_device(int(index)))


def device_name():
    buf = C.create_string_buffer(256)
    check(load().mi355rec_device_name(buf, 256))
    return buf.value.decode()


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def lds_atomic_rate():
    """ds_add_u32 lane-adds per second of the whole device on uniformly random LDS cells (the similarity build's roofline peak)."""
    rate = C.c_double(0.0)
    check(load().mi355rec_lds_atomic_rate(C.byref(rate)))
    return float(rate.value)


def trim_device_cache():
    """Give the library's cache of released device blocks back to the driver (a process that shares the GPU with PyTorch / RCCL calls
    this between phases of fits); returns the bytes freed."""
    freed = C.c_uint64(0)
    check(load().mi355rec_device_trim(C.byref(freed)))
    return int(freed.value)


class DeviceArray:
    """A raw device allocation of the calling process's GPU (int32 words), released with the object."""

    def __init__(self, n_words):
        self.n_words = int(n_words)
        self.ptr = C.c_void_p()
        check(load().mi355rec_device_malloc(C.byref(self.ptr), 4 * self.n_words))

    def address(self, word_offset=0):
        return (self.ptr.value or 0) + 4 * int(word_offset)

    def to_host(self, out=None):
        out = np.empty(self.n_words, np.int32) if out is None else out
        check(load().mi355rec_device_memcpy(ptr(out), self.ptr, 4 * self.n_words, 0))
        return out

    def copy_from_device(self, other, n_words, word_offset=0):
        """Blocking device-to-device copy of other[0:n_words] into self[word_offset:...]."""
        check(load().mi355rec_device_memcpy(C.c_void_p(self.address(word_offset)), other.ptr, 4 * int(n_words), 2))

    def close(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            load().mi355rec_device_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentURM:
    """A CSR matrix (int32 structure, float32 values, sorted indices) uploaded ONCE to the calling process's GPU.  Similarity
    builds started from it (`Compute_Similarity_MI355X(..., resident=...)`, `ItemKNNCFRecommender.fit(..., resident_urm=...)`)
    copy the arrays at HBM speed instead of paying the PCIe upload in every fit -- the reference's hyper-parameter search runs
    hundreds of fits on the same URM_train (ParameterTuning/SearchAbstractClass.py:253-262)."""

    def __init__(self, matrix):
        import scipy.sparse as sps
        csr = sps.csr_matrix(matrix, dtype=np.float32)
        if not csr.has_sorted_indices:
            csr = csr.sorted_indices()
        self.shape = csr.shape
        self.nnz = int(csr.nnz)
        self._fingerprint = self.fingerprint_of(csr)
        self._buffers = self._buffers_of(csr)
        self._host = csr                         # (keeps the buffers alive: their addresses identify the matrix)
        self._full = None
        self.indptr, self.indices, self.data = DeviceArray(len(csr.indptr)), DeviceArray(max(1, self.nnz)), DeviceArray(max(1, self.nnz))
        lib = load()
        for dev, host in ((self.indptr, as_i32(csr.indptr)), (self.indices, as_i32(csr.indices)), (self.data, as_f32(csr.data))):
            if len(host):
                check(lib.mi355rec_device_memcpy(dev.ptr, ptr(host), 4 * len(host), 1))

    @staticmethod
    def fingerprint_of(csr):
        """Shape, nnz and a checksum of a sample of the row pointers (every 64th: each one counts ALL the entries before it), indices and values (every nnz / 8192-th entry:
        a strided sample is one cache miss per entry, 65 536 of them per array cost 0.5 ms of every constructor)."""
        import zlib
        return (csr.shape, int(csr.nnz), zlib.crc32(np.ascontiguousarray(csr.indptr, np.int32)[::64].tobytes()),
                zlib.crc32(np.ascontiguousarray(csr.data, np.float32)[:: max(1, csr.nnz // 8192)].tobytes()),
                zlib.crc32(np.ascontiguousarray(csr.indices, np.int32)[:: max(1, csr.nnz // 8192)].tobytes()))

    @staticmethod
    def _buffers_of(csr):
        return tuple(a.__array_interface__["data"][0] for a in (csr.indptr, csr.indices, csr.data))

    @staticmethod
    def _full_checksum(csr):
        """Digest of EVERY index and value.  xxh3 where the module is there (7 ms per 80 MB array; zlib.crc32 over a copy of the bytes took 150):
        a search that hands every fit its own copy of URM_train pays this once per fit, so it has to stay near the 3 ms upload it saves."""
        arrays = (np.ascontiguousarray(csr.indices, np.int32), np.ascontiguousarray(csr.data, np.float32))
        try:
            import xxhash
            return tuple(xxhash.xxh3_128_digest(memoryview(a)) for a in arrays)
        except ImportError:
            import zlib
            return tuple(zlib.crc32(memoryview(a)) for a in arrays)

    def _rotating_sample_equal(self, csr):
        """A strided sample of indices and values of `csr` against the same positions of the uploaded matrix; the offset moves on
        with every call, so that repeated fits look at different entries (8192 per array and call)."""
        stride = max(1, self.nnz // 8192)
        self._calls = getattr(self, "_calls", 0) + 1
        at = (self._calls * 2654435761) % stride
        mine = self._host
        return (np.array_equal(np.asarray(csr.indices)[at::stride], mine.indices[at::stride]) and
                np.array_equal(np.asarray(csr.data, dtype=np.float32)[at::stride], mine.data[at::stride]))

    def matches(self, csr, thorough=None):
        """Is `csr` the uploaded matrix?  Always compared: shape, nnz, every 64th row pointer, a fixed sample of indices and values and
        a second sample whose offset changes from call to call -- a fraction of a millisecond.  The FIRST time a matrix (identified
        by the addresses of its three buffers) is presented, a checksum of EVERY index and value is compared as well (~15 ms at
        ML-20M size) and the verdict is remembered for those buffers: the hundreds of fits of a search on the same URM_train copy
        (the recommenders copy it in their constructor, BaseRecommender.py:29, so it is never the uploaded object itself) pay it
        once.  thorough=True (or MI355REC_RESIDENT_VERIFY=full) compares the checksum on every call; thorough=False (or
        MI355REC_RESIDENT_VERIFY=sample) never does (the caller vouches for the copy)."""
        import os
        if self.fingerprint_of(csr) != self._fingerprint or not self._rotating_sample_equal(csr):
            return False
        key = self._buffers_of(csr)
        if key == self._buffers:
            return True
        if thorough is None:
            mode = os.environ.get("MI355REC_RESIDENT_VERIFY", "")
            thorough = True if mode == "full" else (False if mode == "sample" else None)
        if thorough is False:
            return True
        seen = self.__dict__.setdefault("_verified", {})
        if thorough is None and key in seen:
            return seen[key]
        if self._full is None:
            self._full = self._full_checksum(self._host)
        verdict = self._full_checksum(csr) == self._full
        if len(seen) >= 64:
            seen.clear()
        seen[key] = verdict
        return verdict

    def close(self):
        for a in (self.indptr, self.indices, self.data):
            a.close()
