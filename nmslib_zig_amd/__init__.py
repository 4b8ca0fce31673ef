"""nmslib_zig_amd -- MI355X-native k-NN engine behind the NMSLIB-ZIG C ABI.

This package is a thin host-side mirror of the reference's Zig binding (lib.zig:495-1270:
Index.init / addDenseBatch / addUInt8Batch / buildIndex / knnQuery / knnQueryBatch / ...)
written against the *same* C ABI (include/nmslib_c.h) that lib.zig binds with @cImport.  All
k-NN work happens inside libnmslib_c.so (HIP kernels for gfx950); nothing here computes
distances, and importing the package fails loudly if the library is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnmslib_c.so")

# nmslib_error_t (nmslib_c.h:21-37) -> exception names follow lib.zig:56-73
ERRORS = {
    1: "NullPointer", 2: "InvalidArgument", 3: "OutOfMemory", 4: "BufferTooSmall",
    5: "SpaceIncompatible", 6: "QueryTooLarge", 7: "InvalidSparseElement", 8: "IndexBuildFailed",
    9: "QueryExecutionFailed", 10: "DataIOFailed", 11: "PluginRegistrationFailed", 12: "Internal",
    13: "Runtime", 14: "IndexNotBuilt",
}
DATATYPE = {"DenseVector": 0, "SparseVector": 1, "DenseUInt8Vector": 2, "ObjectAsString": 3}
DISTTYPE = {"Float": 0, "Int": 1}


class NmslibError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        self.name = ERRORS.get(code, "Runtime")
        super().__init__(f"{self.name} ({code}): {detail}")


class Allocator(C.Structure):
    _fields_ = [("alloc", C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)),
                ("free", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)),
                ("ctx", C.c_void_p)]


class Result(C.Structure):
    _fields_ = [("ids", C.POINTER(C.c_int32)), ("distances", C.POINTER(C.c_float)),
                ("size", C.c_size_t), ("capacity", C.c_size_t)]


class ErrorDetail(C.Structure):
    _fields_ = [("code", C.c_int), ("message", C.c_void_p), ("file", C.c_void_p), ("line", C.c_int)]


class GpuStats(C.Structure):
    _fields_ = [("upload_seconds", C.c_double), ("build_seconds", C.c_double),
                ("hbm_bytes", C.c_size_t), ("rows", C.c_size_t), ("dim", C.c_size_t), ("shards", C.c_size_t),
                ("last_path", C.c_size_t), ("fast_tiles", C.c_size_t), ("fast_tiles_precise", C.c_size_t),
                ("fast_tiles_fallback", C.c_size_t), ("hnsw_redone", C.c_size_t)]


def build_library(force=False):
    """Compile nmslib_zig_amd/libnmslib_c.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_alloc_live = {}


def lib():
    """Load the C ABI.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C nmslib_zig_amd/csrc` (the engine has no CPU fallback)")
    # One HIP runtime per process: PyTorch's wheels request "libamdhip64.so" (their bundled copy)
    # while this library is linked against the SONAME "libamdhip64.so.7".  Opening the runtime
    # by its unversioned NAME first makes both requests resolve to the same loaded object,
    # whichever of torch / this library is imported first (two runtimes in one process cannot
    # both see the GPU).
    try:
        C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    except OSError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i32p, f32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    AP = C.POINTER(Allocator)
    sigs = {
        "nmslib_init": (None, []),
        "nmslib_index_create": (C.c_int, [C.c_char_p, vp, C.c_char_p, C.c_int, C.c_int, AP, C.POINTER(vp)]),
        "nmslib_index_destroy": (None, [vp]),
        "nmslib_create_index": (C.c_int, [vp, vp, C.c_int]),
        "nmslib_reset_index": (C.c_int, [vp]),
        "nmslib_create_params": (vp, [AP]),
        "nmslib_add_param": (C.c_int, [vp, C.c_char_p, C.c_int, vp]),
        "nmslib_free_params": (None, [vp]),
        "nmslib_get_space_type": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), AP]),
        "nmslib_get_method": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), AP]),
        "nmslib_free_string": (None, [vp, AP]),
        "nmslib_get_last_error_detail": (C.c_int, [C.POINTER(ErrorDetail), AP]),
        "nmslib_add_data_point": (C.c_int, [vp, vp, sz, C.c_int32]),
        "nmslib_add_data_point_batch": (C.c_int, [vp, vp, sz, sz, vp, vp]),
        "nmslib_add_data_point_batch_uint8": (C.c_int, [vp, vp, sz, sz, vp]),
        "nmslib_add_data_point_batch_string": (C.c_int, [vp, vp, sz, vp]),
        "nmslib_add_data_point_batch_pointers": (C.c_int, [vp, C.c_int, vp, sz, sz, vp, vp]),
        "nmslib_knn_query_get_size": (C.c_int, [vp, vp, sz, sz, C.POINTER(sz), sz]),
        "nmslib_knn_query_fill": (C.c_int, [vp, vp, sz, sz, C.POINTER(Result), sz]),
        "nmslib_knn_query_batch": (C.c_int, [vp, vp, sz, sz, sz, C.POINTER(Result), vp, sz]),
        "nmslib_range_query_get_size": (C.c_int, [vp, vp, sz, C.c_double, C.POINTER(sz), sz]),
        "nmslib_range_query_fill": (C.c_int, [vp, vp, sz, C.c_double, C.POINTER(Result), sz]),
        "nmslib_get_distance": (C.c_int, [vp, sz, sz, C.POINTER(C.c_float)]),
        "nmslib_get_data_point_size": (C.c_int, [vp, sz, C.POINTER(sz)]),
        "nmslib_get_data_point_fill": (C.c_int, [vp, sz, vp, sz]),
        "nmslib_get_data_point_string": (C.c_int, [vp, sz, C.POINTER(vp), C.POINTER(sz), AP]),
        "nmslib_borrow_data_dense": (C.c_int, [vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)]),
        "nmslib_borrow_data_sparse": (C.c_int, [vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)]),
        "nmslib_save_index": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "nmslib_load_index": (C.c_int, [C.c_char_p, C.c_int, C.c_int, AP, C.c_int, C.POINTER(vp)]),
        "nmslib_set_query_time_params": (C.c_int, [vp, vp]),
        "nmslib_set_thread_pool_size": (C.c_int, [vp, sz]),
        "nmslib_get_thread_pool_size": (sz, [vp]),
        "nmslib_data_qty": (sz, [vp]),
        "nmslib_index_memory_usage": (sz, [vp]),
        "nmslib_initialize_pool": (None, [vp]),
        "nmslib_free_result": (None, [C.POINTER(Result), AP]),
        # device-resident extensions (include/nmslib_gpu.h)
        "nmslib_gpu_device_count": (C.c_int, []),
        "nmslib_gpu_finalize": (C.c_int, [vp]),
        "nmslib_gpu_knn_query_batch_device": (C.c_int, [vp, vp, sz, sz, sz, vp, vp, vp, vp]),
        "nmslib_gpu_last_batch_counters": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "nmslib_gpu_merge_topk": (C.c_int, [vp, vp, sz, sz, sz, vp, vp, vp]),
        "nmslib_gpu_merge_topk_strided": (C.c_int, [vp, vp, sz, sz, sz, sz, vp, vp, vp]),
        "nmslib_gpu_get_stats": (C.c_int, [vp, C.POINTER(GpuStats)]),
        "nmslib_gpu_kernel_timing": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)   # AttributeError here = a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


ABI_SYMBOLS_C = [  # the 37 symbols of the reference boundary
    "nmslib_init", "nmslib_index_create", "nmslib_index_destroy", "nmslib_create_index",
    "nmslib_reset_index", "nmslib_create_params", "nmslib_add_param", "nmslib_free_params",
    "nmslib_get_space_type", "nmslib_get_method", "nmslib_free_string",
    "nmslib_get_last_error_detail", "nmslib_add_data_point", "nmslib_add_data_point_batch",
    "nmslib_add_data_point_batch_uint8", "nmslib_add_data_point_batch_string",
    "nmslib_knn_query_get_size", "nmslib_knn_query_fill", "nmslib_knn_query_batch",
    "nmslib_range_query_get_size", "nmslib_range_query_fill", "nmslib_get_distance",
    "nmslib_get_data_point_size", "nmslib_get_data_point_fill", "nmslib_get_data_point_string",
    "nmslib_borrow_data_dense", "nmslib_borrow_data_sparse", "nmslib_save_index",
    "nmslib_load_index", "nmslib_set_query_time_params", "nmslib_set_thread_pool_size",
    "nmslib_get_thread_pool_size", "nmslib_data_qty", "nmslib_index_memory_usage",
    "nmslib_add_data_point_batch_pointers", "nmslib_initialize_pool", "nmslib_free_result",
]
ABI_SYMBOLS_GPU = ["nmslib_gpu_device_count", "nmslib_gpu_finalize",
                   "nmslib_gpu_knn_query_batch_device", "nmslib_gpu_last_batch_counters",
                   "nmslib_gpu_merge_topk", "nmslib_gpu_merge_topk_strided", "nmslib_gpu_get_stats", "nmslib_gpu_kernel_timing"]


class TrackingAllocator:
    """The caller-side allocator the ABI requires (lib.zig:192-257): counts live blocks so tests
    can assert that the library returns everything it took."""

    def __init__(self):
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        libc.free.argtypes = [C.c_void_p]
        self.live = {}
        self._libc = libc

        def _alloc(n, ctx):
            p = libc.malloc(max(n, 1))
            if p:
                self.live[p] = n
            return p

        def _free(p, ctx):
            if p:
                self.live.pop(p, None)
                libc.free(p)

        self._a = Allocator._fields_[0][1](_alloc)
        self._f = Allocator._fields_[1][1](_free)
        self.c = Allocator(self._a, self._f, None)

    def ref(self):
        return C.byref(self.c)


def last_error_detail(alloc):
    """nmslib_get_last_error_detail -> message; the returned strings are allocator-owned and are
    released with nmslib_free_string, as lib.zig does (lib.zig:564-571)."""
    d = ErrorDetail()
    L = lib()
    if L.nmslib_get_last_error_detail(C.byref(d), alloc.ref()) != 0:
        return ""
    msg = C.string_at(d.message).decode() if d.message else ""
    if d.message:
        L.nmslib_free_string(d.message, alloc.ref())
    if d.file:
        L.nmslib_free_string(d.file, alloc.ref())
    return msg


def _check(rc, alloc=None):
    if rc != 0:
        raise NmslibError(rc, last_error_detail(alloc) if alloc is not None else "")


class Params:
    """lib.zig:260-348 Params: typed name/value pairs behind nmslib_create_params / nmslib_add_param."""

    def __init__(self, alloc, **kw):
        self.alloc = alloc
        self.h = C.c_void_p(lib().nmslib_create_params(alloc.ref()))
        if not self.h:
            raise NmslibError(3, "nmslib_create_params")
        for k, v in kw.items():
            self.add(k, v)

    def add(self, key, value):
        L = lib()
        if isinstance(value, bool):
            value = int(value)
        if isinstance(value, int):
            v = C.c_int(value)
            _check(L.nmslib_add_param(self.h, key.encode(), 0, C.byref(v)), self.alloc)
        elif isinstance(value, float):
            v = C.c_double(value)
            _check(L.nmslib_add_param(self.h, key.encode(), 1, C.byref(v)), self.alloc)
        else:
            s = C.create_string_buffer(str(value).encode())
            _check(L.nmslib_add_param(self.h, key.encode(), 2, s), self.alloc)

    def free(self):
        if self.h:
            lib().nmslib_free_params(self.h)
            self.h = None


class Index:
    """Mirror of lib.zig's Index (lib.zig:495-1270) over the same C ABI.

    Call order is the reference's own *data-first* order (add -> nmslib_create_index -> queries),
    the order BASELINE.md prescribes; lib.zig's create-then-add order is also accepted by the
    library (nmslib_initialize_pool finalises a dirty index once)."""

    def __init__(self, space, method="hnsw", data_type="DenseVector", dist_type="Float", space_params=None):
        L = lib()
        L.nmslib_init()
        self.alloc = TrackingAllocator()
        if space == "cosine":        # lib.zig:530-533 canonicalisation
            space = "cosinesimil"
        self.data_type = data_type
        self.h = C.c_void_p()
        sp = Params(self.alloc, **space_params) if space_params else None
        rc = L.nmslib_index_create(space.encode(), sp.h if sp else None, method.encode(), DATATYPE[data_type],
                                   DISTTYPE[dist_type], self.alloc.ref(), C.byref(self.h))
        if sp:
            sp.free()
        _check(rc, self.alloc)
        self.built = False

    # -- data ------------------------------------------------------------------------------
    def addDenseBatch(self, data, ids=None):
        data = np.ascontiguousarray(data, np.float32)
        idp = None if ids is None else np.ascontiguousarray(ids, np.int32)
        _check(lib().nmslib_add_data_point_batch(self.h, data.ctypes.data, data.shape[0], data.shape[1],
                                                 None if idp is None else idp.ctypes.data, None), self.alloc)

    def addUInt8Batch(self, data, ids=None):
        data = np.ascontiguousarray(data, np.uint8)
        idp = None if ids is None else np.ascontiguousarray(ids, np.int32)
        _check(lib().nmslib_add_data_point_batch_uint8(self.h, data.ctypes.data, data.shape[0], data.shape[1],
                                                       None if idp is None else idp.ctypes.data), self.alloc)

    def buildIndex(self, **index_params):
        p = Params(self.alloc, **index_params) if index_params else None
        try:
            _check(lib().nmslib_create_index(self.h, p.h if p else None, 0), self.alloc)
        finally:
            if p:
                p.free()
        self.built = True

    def setQueryTimeParams(self, **params):
        p = Params(self.alloc, **params)
        try:
            _check(lib().nmslib_set_query_time_params(self.h, p.h), self.alloc)
        finally:
            p.free()

    # -- queries (host buffers: the reference's own entry points) -----------------------------
    def knnQuery(self, query, k):
        L = lib()
        q = np.ascontiguousarray(query, np.uint8 if self.data_type == "DenseUInt8Vector" else np.float32)
        L.nmslib_initialize_pool(self.h)      # lib.zig:802
        cap = C.c_size_t()
        _check(L.nmslib_knn_query_get_size(self.h, q.ctypes.data, q.shape[0], k, C.byref(cap), 0), self.alloc)
        ids = np.empty(cap.value, np.int32)
        ds = np.empty(cap.value, np.float32)
        r = Result(ids.ctypes.data_as(C.POINTER(C.c_int32)), ds.ctypes.data_as(C.POINTER(C.c_float)), 0, cap.value)
        _check(L.nmslib_knn_query_fill(self.h, q.ctypes.data, q.shape[0], k, C.byref(r), 0), self.alloc)
        return ids[:r.size].copy(), ds[:r.size].copy()

    def rangeQuery(self, query, radius):
        """lib.zig:933-965: get_size (an estimate, 128) sizes the buffers, fill writes the first `capacity`
        objects within the radius, in insertion order.  HNSW -> NmslibError(SPACE_INCOMPATIBLE)."""
        L = lib()
        q = np.ascontiguousarray(query, np.uint8 if self.data_type == "DenseUInt8Vector" else np.float32)
        L.nmslib_initialize_pool(self.h)
        cap = C.c_size_t()
        _check(L.nmslib_range_query_get_size(self.h, q.ctypes.data, q.shape[0], float(radius), C.byref(cap), 0), self.alloc)
        return self.rangeQueryFill(q, radius, cap.value)

    def rangeQueryFill(self, query, radius, capacity):
        L = lib()
        q = np.ascontiguousarray(query, np.uint8 if self.data_type == "DenseUInt8Vector" else np.float32)
        ids = np.empty(capacity, np.int32)
        ds = np.empty(capacity, np.float32)
        r = Result(ids.ctypes.data_as(C.POINTER(C.c_int32)), ds.ctypes.data_as(C.POINTER(C.c_float)), 0, capacity)
        _check(L.nmslib_range_query_fill(self.h, q.ctypes.data, q.shape[0], float(radius), C.byref(r), 0), self.alloc)
        return ids[:r.size].copy(), ds[:r.size].copy()

    def knnQueryBatch(self, queries, k):
        """One call of nmslib_knn_query_batch: a single GPU batch.  -> ids [Q,k], dists [Q,k], counts [Q]"""
        L = lib()
        q = np.ascontiguousarray(queries, np.uint8 if self.data_type == "DenseUInt8Vector" else np.float32)
        nq = q.shape[0]
        ids = np.full((nq, k), -1, np.int32)
        ds = np.full((nq, k), np.inf, np.float32)
        res = (Result * nq)()
        for i in range(nq):
            res[i] = Result(ids[i].ctypes.data_as(C.POINTER(C.c_int32)),
                            ds[i].ctypes.data_as(C.POINTER(C.c_float)), 0, k)
        _check(L.nmslib_knn_query_batch(self.h, q.ctypes.data, nq, q.shape[1], k, res, None, 0), self.alloc)
        cnt = np.array([res[i].size for i in range(nq)], np.int32)
        return ids, ds, cnt

    # -- device-resident entry (include/nmslib_gpu.h) -------------------------------------------
    def finalize(self):
        _check(lib().nmslib_gpu_finalize(self.h), self.alloc)

    def knn_device(self, d_queries_ptr, nq, dim, k, d_ids_ptr, d_dists_ptr, d_cnt_ptr=None, stream=None):
        _check(lib().nmslib_gpu_knn_query_batch_device(self.h, d_queries_ptr, nq, dim, k, d_ids_ptr, d_dists_ptr,
                                                       d_cnt_ptr, stream), self.alloc)

    def last_counters(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().nmslib_gpu_last_batch_counters(self.h, C.byref(a), C.byref(b), C.byref(c)), self.alloc)
        return a.value, b.value, c.value

    def read_counters(self, nq):
        """Copy the per-query work counters of the last HNSW batch to the host:
        -> (ndc, hops, hops_up) int32 arrays, or None after a brute-force batch."""
        pn, ph, pu = self.last_counters()
        if not pn:
            return None
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipDeviceSynchronize()
        out = []
        for p in (pn, ph, pu):
            a = np.empty(nq, np.int32)
            rc = hip.hipMemcpy(a.ctypes.data, p, 4 * nq, 2)   # hipMemcpyDeviceToHost
            if rc != 0:
                raise RuntimeError(f"hipMemcpy failed: {rc}")
            out.append(a)
        return tuple(out)

    def kernel_timing(self, enable=True, collect=False):
        """HIP-event timing of the dominant kernel: -> (total_ms, launches) when collect."""
        ms, n = C.c_double(), C.c_uint64()
        _check(lib().nmslib_gpu_kernel_timing(self.h, int(enable), C.byref(ms) if collect else None,
                                              C.byref(n) if collect else None), self.alloc)
        return (ms.value, n.value) if collect else None

    def stats(self):
        s = GpuStats()
        _check(lib().nmslib_gpu_get_stats(self.h, C.byref(s)), self.alloc)
        return {f[0]: getattr(s, f[0]) for f in GpuStats._fields_}

    # -- metadata / stored data -------------------------------------------------------------------
    def getDistance(self, a, b):
        v = C.c_float()
        _check(lib().nmslib_get_distance(self.h, a, b, C.byref(v)), self.alloc)
        return v.value

    def dataQty(self):
        return lib().nmslib_data_qty(self.h)

    def _string(self, fn):
        p, n = C.c_void_p(), C.c_size_t()
        _check(fn(self.h, C.byref(p), C.byref(n), self.alloc.ref()), self.alloc)
        s = C.string_at(p, n.value).decode()
        lib().nmslib_free_string(p, self.alloc.ref())
        return s

    def getSpaceType(self):
        s = self._string(lib().nmslib_get_space_type)
        return "cosine" if s == "cosinesimil" else s   # lib.zig:1224-1240 round-trips the alias

    def getMethod(self):
        return self._string(lib().nmslib_get_method)

    def getDataPoint(self, pos):
        n = C.c_size_t()
        _check(lib().nmslib_get_data_point_size(self.h, pos, C.byref(n)), self.alloc)
        buf = np.empty(n.value, np.uint8)
        _check(lib().nmslib_get_data_point_fill(self.h, pos, buf.ctypes.data, n.value), self.alloc)
        if self.data_type == "DenseUInt8Vector":
            return buf[:128].copy()
        return buf.view(np.float32).copy()

    def save(self, path, save_data=True):
        _check(lib().nmslib_save_index(self.h, path.encode(), int(save_data)), self.alloc)

    @classmethod
    def load(cls, path, data_type="DenseVector", dist_type="Float", load_data=True):
        self = cls.__new__(cls)
        self.alloc = TrackingAllocator()
        self.data_type = data_type
        self.h = C.c_void_p()
        _check(lib().nmslib_load_index(path.encode(), DATATYPE[data_type], DISTTYPE[dist_type], self.alloc.ref(),
                                       int(load_data), C.byref(self.h)), self.alloc)
        self.built = True
        return self

    def setThreadPoolSize(self, n):
        _check(lib().nmslib_set_thread_pool_size(self.h, n), self.alloc)

    def getThreadPoolSize(self):
        return lib().nmslib_get_thread_pool_size(self.h)

    def reset(self):
        _check(lib().nmslib_reset_index(self.h), self.alloc)
        self.built = False

    def close(self):
        if getattr(self, "h", None):
            lib().nmslib_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
