"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REF_LIB = os.path.join(REF_DIR, "libnmslib_ref.so")
REF_DRIVER = os.path.join(REF_DIR, "ref_driver")

SPACES = {"l2": 0, "l1": 1, "linf": 2, "cosinesimil": 3, "angulardist": 4, "negdotprod": 5,
          "l2sqr_sift": 6}

_lib = None


def build():
    """Compile oracle/liboracle.so if missing or stale (gcc, plain C)."""
    src = os.path.join(ORACLE_DIR, "knn_oracle.c")
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"],
                              stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    f32p, i32p, i64p, vp = (C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                            C.c_void_p)
    for name in ("orc_l2sqr_simd", "orc_l2_simd", "orc_l1_simd", "orc_linf_simd", "orc_dot_simd",
                 "orc_normdot_simd", "orc_cosine", "orc_angular", "orc_l2sqr16_avx",
                 "orc_l2sqr_avx", "orc_dot_avx"):
        fn = getattr(L, name)
        fn.restype = C.c_float
        fn.argtypes = [vp, vp, C.c_size_t]
    L.orc_space_distance.restype = C.c_double
    L.orc_space_distance.argtypes = [C.c_int, vp, vp, C.c_size_t]
    L.orc_hnsw_opt_distance.restype = C.c_double
    L.orc_hnsw_opt_distance.argtypes = [C.c_int, vp, vp, C.c_size_t]
    L.orc_seq_search.restype = None
    L.orc_seq_search.argtypes = [C.c_int, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t,
                                 vp, vp, vp]
    L.orc_hnsw_build.restype = vp
    L.orc_hnsw_build.argtypes = [C.c_int, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_uint32, C.c_int]
    L.orc_hnsw_from_arrays.restype = vp
    L.orc_hnsw_from_arrays.argtypes = [C.c_int, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                       C.c_int, C.c_int, vp, vp, vp, vp]
    L.orc_hnsw_free.restype = None
    L.orc_hnsw_free.argtypes = [vp]
    L.orc_hnsw_maxlevel.restype = C.c_int
    L.orc_hnsw_maxlevel.argtypes = [vp]
    L.orc_hnsw_enterpoint.restype = C.c_int
    L.orc_hnsw_enterpoint.argtypes = [vp]
    L.orc_hnsw_levels.restype = None
    L.orc_hnsw_levels.argtypes = [vp, vp]
    L.orc_hnsw_links0.restype = None
    L.orc_hnsw_links0.argtypes = [vp, vp]
    L.orc_hnsw_links_up.restype = None
    L.orc_hnsw_links_up.argtypes = [vp, C.c_int, C.c_int, vp]
    L.orc_hnsw_search.restype = None
    L.orc_hnsw_search.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t,
                                  vp, vp, vp, vp, vp]
    L.orc_random_levels.restype = None
    L.orc_random_levels.argtypes = [C.c_uint32, C.c_double, C.c_int, C.c_size_t, vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _rows(space, x):
    dt = np.uint8 if space == "l2sqr_sift" else np.float32
    return np.ascontiguousarray(x, dtype=dt)


def space_distance(space, a, b):
    a, b = _rows(space, a), _rows(space, b)
    return lib().orc_space_distance(SPACES[space], _ptr(a), _ptr(b), a.shape[-1])


def hnsw_opt_distance(space, q, b):
    q, b = _rows(space, q), _rows(space, b)
    return lib().orc_hnsw_opt_distance(SPACES[space], _ptr(q), _ptr(b), q.shape[-1])


def seq_search(space, base, queries, k):
    """-> (positions [Q,k] int32, distances [Q,k] f32, counts [Q])"""
    base, queries = _rows(space, base), _rows(space, queries)
    nq = queries.shape[0]
    pos = np.full((nq, k), -1, np.int32)
    dist = np.full((nq, k), np.inf, np.float32)
    cnt = np.zeros(nq, np.int32)
    lib().orc_seq_search(SPACES[space], _ptr(base), base.shape[0], base.shape[1], _ptr(queries),
                         nq, k, _ptr(pos), _ptr(dist), _ptr(cnt))
    return pos, dist, cnt


class HnswGraph:
    """Owns an orc_hnsw_t (and keeps the base rows alive)."""

    def __init__(self, handle, space, base, maxM, maxM0):
        self.h, self.space, self.base, self.maxM, self.maxM0 = handle, space, base, maxM, maxM0
        self.n = base.shape[0]

    @classmethod
    def build(cls, space, base, M=16, efConstruction=200, maxM=None, maxM0=None,
              delaunay_type=2, seed=0, log_variant=0):
        base = _rows(space, base)
        maxM = M if maxM is None else maxM
        maxM0 = 2 * M if maxM0 is None else maxM0
        h = lib().orc_hnsw_build(SPACES[space], _ptr(base), base.shape[0], base.shape[1], M, maxM,
                                 maxM0, efConstruction, delaunay_type, seed, log_variant)
        return cls(h, space, base, maxM, maxM0)

    @classmethod
    def from_arrays(cls, space, base, maxM, maxM0, maxlevel, enterpoint, levels, links0, up_off,
                    up_links):
        base = _rows(space, base)
        levels = np.ascontiguousarray(levels, np.int32)
        links0 = np.ascontiguousarray(links0, np.int32)
        up_off = np.ascontiguousarray(up_off, np.int64)
        up_links = np.ascontiguousarray(up_links if len(up_links) else np.zeros(1), np.int32)
        h = lib().orc_hnsw_from_arrays(SPACES[space], _ptr(base), base.shape[0], base.shape[1],
                                       maxM, maxM0, maxlevel, enterpoint, _ptr(levels),
                                       _ptr(links0), _ptr(up_off), _ptr(up_links))
        return cls(h, space, base, maxM, maxM0)

    @property
    def maxlevel(self):
        return lib().orc_hnsw_maxlevel(self.h)

    @property
    def enterpoint(self):
        return lib().orc_hnsw_enterpoint(self.h)

    def levels(self):
        out = np.zeros(self.n, np.int32)
        lib().orc_hnsw_levels(self.h, _ptr(out))
        return out

    def links0(self):
        out = np.zeros((self.n, self.maxM0 + 1), np.int32)
        lib().orc_hnsw_links0(self.h, _ptr(out))
        return out

    def links_up(self, i, level):
        out = np.zeros(self.maxM + 1, np.int32)
        lib().orc_hnsw_links_up(self.h, int(i), int(level), _ptr(out))
        return out

    def flat_upper(self):
        """-> (up_off [n] int64 (-1 = none), up_links int32) in the layout of orc_hnsw_from_arrays."""
        lv = self.levels()
        off = np.full(self.n, -1, np.int64)
        chunks, cur = [], 0
        for i in np.nonzero(lv > 0)[0]:
            off[i] = cur
            for l in range(1, lv[i] + 1):
                chunks.append(self.links_up(i, l))
                cur += self.maxM + 1
        return off, (np.concatenate(chunks) if chunks else np.zeros(0, np.int32))

    def search(self, queries, k, ef, optimized=True, algo="v1merge"):
        """-> pos, dist, cnt, ndc, hops"""
        queries = _rows(self.space, queries)
        nq = queries.shape[0]
        pos = np.full((nq, k), -1, np.int32)
        dist = np.full((nq, k), np.inf, np.float32)
        cnt = np.zeros(nq, np.int32)
        ndc = np.zeros(nq, np.int64)
        hops = np.zeros(nq, np.int64)
        lib().orc_hnsw_search(self.h, int(bool(optimized)), 0 if algo == "v1merge" else 1,
                              _ptr(queries), nq, k, ef, _ptr(pos), _ptr(dist), _ptr(cnt),
                              _ptr(ndc), _ptr(hops))
        return pos, dist, cnt, ndc, hops

    def __del__(self):
        try:
            if self.h:
                lib().orc_hnsw_free(self.h)
                self.h = None
        except Exception:
            pass


def random_levels(n, M=16, seed=0, log_variant=0):
    out = np.zeros(n, np.int32)
    lib().orc_random_levels(seed, 1.0 / np.log(1.0 * M), log_variant, n, _ptr(out))
    return out
