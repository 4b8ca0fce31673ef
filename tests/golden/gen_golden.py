#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref, built by oracle/Makefile
from /root/reference).  Run in the dev container only:

    make -C oracle ref && python3 tests/golden/gen_golden.py

The fixtures are data: seeded inputs and the reference's outputs for them.  The reference has
no golden vectors of its own for this path, so these pin both the CPU oracle
(tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_*.py).
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import orc, refio  # noqa: E402

assert refio.HAVE_REF, "build oracle/_ref first: make -C oracle ref"


class RefCABI:
    """The reference's own C shim (nmslib_c.h) through ctypes, data-first call order."""

    class Alloc(C.Structure):
        _fields_ = [("alloc", C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)),
                    ("free", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)), ("ctx", C.c_void_p)]

    class Result(C.Structure):
        _fields_ = [("ids", C.POINTER(C.c_int32)), ("distances", C.POINTER(C.c_float)),
                    ("size", C.c_size_t), ("capacity", C.c_size_t)]

    def __init__(self):
        self.L = C.CDLL(orc.REF_LIB)
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        libc.free.argtypes = [C.c_void_p]
        self._a = self.Alloc._fields_[0][1](lambda n, ctx: libc.malloc(n))
        self._f = self.Alloc._fields_[1][1](lambda p, ctx: libc.free(p))
        self.alloc = self.Alloc(self._a, self._f, None)
        self.L.nmslib_init()

    def index(self, space, method, data_type, dist_type):
        h = C.c_void_p()
        rc = self.L.nmslib_index_create(space.encode(), None, method.encode(), data_type,
                                        dist_type, C.byref(self.alloc), C.byref(h))
        assert rc == 0, rc
        return h

    def params(self, **kw):
        self.L.nmslib_create_params.restype = C.c_void_p
        p = C.c_void_p(self.L.nmslib_create_params(C.byref(self.alloc)))
        for k, v in kw.items():
            iv = C.c_int(int(v))
            assert self.L.nmslib_add_param(p, k.encode(), 0, C.byref(iv)) == 0
        return p

    def knn(self, h, q, k):
        ids = (C.c_int32 * k)()
        ds = (C.c_float * k)()
        r = self.Result(ids, ds, 0, k)
        q = np.ascontiguousarray(q)
        rc = self.L.nmslib_knn_query_fill(h, q.ctypes.data_as(C.c_void_p), C.c_size_t(q.shape[0]),
                                          C.c_size_t(k), C.byref(r), C.c_size_t(0))
        return rc, np.array(ids[:r.size], np.int32), np.array(ds[:r.size], np.float32)


def main():
    out = {}
    cabi = RefCABI()
    L = cabi.L

    # ---- (1) sequential search, float spaces: D in {128 (16-unrolled), 100 (4-tail), 21 (scalar tail)}
    for D in (128, 100, 21):
        base = refio.s_lowrank(300, D, seed=100 + D)
        base[7] = 0.0                      # zero vector: cosine zero-norm rule (distcomp_scalar.cc:154-160)
        base[11] = base[3]                 # exact duplicate: tie ordering by position
        qs = refio.s_lowrank(8, D, seed=200 + D)
        qs[5] = base[3]                    # a query that ties two rows at distance 0
        out[f"f32_D{D}_base"] = base
        out[f"f32_D{D}_queries"] = qs
        for space in ("l2", "l1", "linf", "cosinesimil", "angulardist", "negdotprod"):
            ids, d, cnt, _, _ = refio.run_ref_driver(space, "seq_search", base, qs, 10)
            out[f"seq_{space}_D{D}_ids"] = ids
            out[f"seq_{space}_D{D}_dists"] = d
            # pairwise nmslib_get_distance through the reference's own C shim
            h = cabi.index(space, "seq_search", 0, 0)
            assert L.nmslib_add_data_point_batch(h, base.ctypes.data_as(C.c_void_p),
                                                 C.c_size_t(300), C.c_size_t(D), None, None) == 0
            pairs = np.array([(i, (i * 7 + 3) % 300) for i in range(0, 32)], np.int32)
            pd = np.zeros(len(pairs), np.float32)
            for j, (a, b) in enumerate(pairs):
                v = C.c_float()
                assert L.nmslib_get_distance(h, C.c_size_t(int(a)), C.c_size_t(int(b)), C.byref(v)) == 0
                pd[j] = v.value
            out[f"pair_{space}_D{D}_idx"] = pairs
            out[f"pair_{space}_D{D}_dists"] = pd
            L.nmslib_index_destroy(h)

    # ---- (2) uint8 SIFT with deliberate ties: small alphabet + planted duplicates
    rng = np.random.default_rng(44)
    u8 = refio.s_sift_like(1000, seed=44)
    u8[500:700] = (rng.integers(0, 4, (200, 128)) * 60).astype(np.uint8)   # 4-symbol alphabet
    u8[700:720] = u8[10]                                                     # 20 copies of one row
    q8 = refio.s_sift_like(6, seed=45)
    q8[1] = u8[10]
    q8[2] = (rng.integers(0, 4, 128) * 60).astype(np.uint8)
    out["u8_base"], out["u8_queries"] = u8, q8
    ids, d, cnt, _, _ = refio.run_ref_driver("l2sqr_sift", "seq_search", u8, q8, 100)
    out["seq_l2sqr_sift_ids"], out["seq_l2sqr_sift_dists"] = ids, d
    ids, d, cnt, ndc, _ = refio.run_ref_driver("l2sqr_sift", "hnsw", u8, q8, 100,
                                               "M=8,efConstruction=50,indexThreadQty=1", "efSearch=150")
    out["hnsw_l2sqr_sift_ids"], out["hnsw_l2sqr_sift_dists"], out["hnsw_l2sqr_sift_ndc"] = ids, d, ndc

    # ---- (3) HNSW: deterministic single-thread build (seed 0), parsed adjacency, searches
    tmp = tempfile.mkdtemp(prefix="golden_")
    for space, D in (("l2", 128), ("cosinesimil", 100), ("negdotprod", 21), ("l1", 21)):
        base = out[f"f32_D{D}_base"]
        qs = out[f"f32_D{D}_queries"]
        path = os.path.join(tmp, f"{space}.idx")
        refio.run_ref_driver(space, "hnsw", base, qs, 10, "M=8,efConstruction=50,indexThreadQty=1",
                             "efSearch=20", save=path)
        P = refio.parse_optimized_index(path)
        for key in ("links0", "levels", "up_off", "up_links"):
            out[f"hnsw_{space}_{key}"] = P[key]
        out[f"hnsw_{space}_meta"] = np.array([P["maxlevel"], P["enterpoint"], P["maxM"], P["maxM0"],
                                              P["mem_per_obj"], P["off_level0"], P["dist_func"],
                                              P["search_method"]], np.int64)
        for ef in (5, 20, 200):
            for algo in ("v1merge", "old"):
                ids, d, cnt, _, _ = refio.run_ref_driver(space, "hnsw", base, qs, 10, "",
                                                         f"efSearch={ef},algoType={algo}", load=path)
                out[f"hnsw_{space}_ef{ef}_{algo}_ids"] = ids
                out[f"hnsw_{space}_ef{ef}_{algo}_dists"] = d
        if space == "l2":   # raw index file bytes for the loader test (N1); 300 nodes ~ 110 KB
            out["hnsw_l2_index_file"] = np.fromfile(path, np.uint8)
    # generic-path HNSW (no optimized index: angulardist), incl. distance computation counts
    base, qs = out["f32_D21_base"], out["f32_D21_queries"]
    ids, d, cnt, ndc, _ = refio.run_ref_driver("angulardist", "hnsw", base, qs, 10,
                                               "M=8,efConstruction=50,indexThreadQty=1", "efSearch=20")
    out["hnsw_angulardist_ids"], out["hnsw_angulardist_dists"], out["hnsw_angulardist_ndc"] = ids, d, ndc

    # ---- (4) the C shim's observable quirks (SURVEY.md 8a "behavioural facts")
    base, qs = out["f32_D128_base"], out["f32_D128_queries"]
    h = cabi.index("l2", "hnsw", 0, 0)
    ids_in = (np.arange(300, dtype=np.int32) * 3 + 5)
    assert L.nmslib_add_data_point_batch(h, base.ctypes.data_as(C.c_void_p), C.c_size_t(300),
                                         C.c_size_t(128), ids_in.ctypes.data_as(C.c_void_p), None) == 0
    p = cabi.params(M=8, efConstruction=50, indexThreadQty=1)
    assert L.nmslib_create_index(h, p, 0) == 0
    rows_i, rows_d = [], []
    for q in qs:
        rc, i, d = cabi.knn(h, q, 10)
        assert rc == 0
        rows_i.append(i)
        rows_d.append(d)
    out["cabi_hnsw_l2_ids"] = np.stack(rows_i)          # external ids, efSearch forced to 200
    out["cabi_hnsw_l2_dists"] = np.stack(rows_d)        # squared L2 (optimized index)
    out["cabi_ids_in"] = ids_in
    rc, i, d = cabi.knn(h, qs[0], 10)
    small = cabi.Result((C.c_int32 * 4)(), (C.c_float * 4)(), 99, 4)   # capacity < k
    rc2 = L.nmslib_knn_query_fill(h, qs[0].ctypes.data_as(C.c_void_p), C.c_size_t(128), C.c_size_t(10),
                                  C.byref(small), C.c_size_t(0))
    out["cabi_small_buffer"] = np.array([rc2, small.size], np.int64)    # (SUCCESS, 0): nmslib_c.cpp:307-312
    L.nmslib_index_destroy(h)

    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
