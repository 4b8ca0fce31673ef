#!/usr/bin/env python3
"""Second fixture file from the REAL reference (oracle/_ref): cases added after golden_v1.

    make -C oracle ref && python3 tests/golden/gen_golden_v2.py

  * 768-dimensional rows (D in {128, 100, 768}): sequential search for l2 / cosinesimil and a
    deterministic single-thread cosine HNSW (adjacency + SearchV1Merge results);
  * wide level-0 lists (M = 32 -> maxM0 = 64): single-thread l2 HNSW, adjacency + searches;
  * range queries through the reference's own C shim (nmslib_range_query_fill) on seq_search.
Large inputs are NOT stored: they are regenerated from their seeds with refio.s_gauss (PCG64 + ziggurat, no BLAS:
bit-identical on every machine) and pinned by a SHA-256 of their bytes.
"""
import ctypes as C
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import refio  # noqa: E402


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def inputs_768():
    return refio.s_gauss(2000, 768, seed=768), refio.s_gauss(16, 768, seed=769)


def inputs_hnsw768():
    return refio.s_gauss(5000, 768, seed=770), refio.s_gauss(32, 768, seed=771)


def inputs_wide():
    return refio.s_gauss(3000, 32, seed=772), refio.s_gauss(32, 32, seed=773)


def main():
    assert refio.HAVE_REF, "build oracle/_ref first: make -C oracle ref"
    from tests.golden.gen_golden import RefCABI
    out = {}
    tmp = tempfile.mkdtemp(prefix="golden2_")

    base, qs = inputs_768()
    out["d768_base_sha"], out["d768_queries_sha"] = sha(base), sha(qs)
    for space in ("l2", "cosinesimil"):
        ids, d, _, _, _ = refio.run_ref_driver(space, "seq_search", base, qs, 10)
        out[f"seq_{space}_D768_ids"], out[f"seq_{space}_D768_dists"] = ids, d

    base, qs = inputs_hnsw768()
    out["hnsw768_base_sha"], out["hnsw768_queries_sha"] = sha(base), sha(qs)
    path = os.path.join(tmp, "cos768.idx")
    refio.run_ref_driver("cosinesimil", "hnsw", base, qs, 10, "M=16,efConstruction=100,indexThreadQty=1",
                         "efSearch=10", save=path)
    P = refio.parse_optimized_index(path)
    for key in ("links0", "levels", "up_links"):
        out[f"hnsw768_{key}"] = P[key]
    out["hnsw768_meta"] = np.array([P["maxlevel"], P["enterpoint"], P["maxM"], P["maxM0"]], np.int64)
    for ef in (10, 128, 200):
        ids, d, _, ndc, _ = refio.run_ref_driver("cosinesimil", "hnsw", base, qs, 10, "", f"efSearch={ef}", load=path)
        out[f"hnsw768_ef{ef}_ids"], out[f"hnsw768_ef{ef}_dists"], out[f"hnsw768_ef{ef}_ndc"] = ids, d, ndc

    base, qs = inputs_wide()
    out["wide_base_sha"], out["wide_queries_sha"] = sha(base), sha(qs)
    path = os.path.join(tmp, "wide.idx")
    refio.run_ref_driver("l2", "hnsw", base, qs, 10, "M=32,efConstruction=120,indexThreadQty=1", "efSearch=40",
                         save=path)
    P = refio.parse_optimized_index(path)
    for key in ("links0", "levels", "up_links"):
        out[f"wide_{key}"] = P[key]
    out["wide_meta"] = np.array([P["maxlevel"], P["enterpoint"], P["maxM"], P["maxM0"]], np.int64)
    for ef in (40, 150):
        ids, d, _, ndc, _ = refio.run_ref_driver("l2", "hnsw", base, qs, 10, "", f"efSearch={ef}", load=path)
        out[f"wide_ef{ef}_ids"], out[f"wide_ef{ef}_dists"], out[f"wide_ef{ef}_ndc"] = ids, d, ndc

    # range queries through the reference's C shim: insertion order, capacity cut, distances by IndexTimeDistance
    g1 = np.load(os.path.join(HERE, "golden_v1.npz"))
    cabi = RefCABI()
    L = cabi.L
    for space, D in (("l2", 128), ("cosinesimil", 100), ("l1", 21)):
        base, qs = g1[f"f32_D{D}_base"], g1[f"f32_D{D}_queries"]
        h = cabi.index(space, "seq_search", 0, 0)
        ext = (np.arange(300, dtype=np.int32) * 2 + 9)
        assert L.nmslib_add_data_point_batch(h, base.ctypes.data_as(C.c_void_p), C.c_size_t(300), C.c_size_t(D),
                                             ext.ctypes.data_as(C.c_void_p), None) == 0
        assert L.nmslib_create_index(h, None, 0) == 0
        seq_d = g1[f"seq_{space}_D{D}_dists"]
        for qi in (0, 3):
            radius = float(seq_d[qi, 9]) * 1.5 + 1e-3          # well past the 10th neighbour: dozens of matches
            for cap in (128, 7):
                ids = (C.c_int32 * cap)()
                ds = (C.c_float * cap)()
                r = cabi.Result(ids, ds, 0, cap)
                q = np.ascontiguousarray(qs[qi])
                rc = L.nmslib_range_query_fill(h, q.ctypes.data_as(C.c_void_p), C.c_size_t(D), C.c_double(radius),
                                               C.byref(r), C.c_size_t(0))
                assert rc == 0, rc
                out[f"range_{space}_q{qi}_cap{cap}_ids"] = np.array(ids[:r.size], np.int32)
                out[f"range_{space}_q{qi}_cap{cap}_dists"] = np.array(ds[:r.size], np.float32)
            out[f"range_{space}_q{qi}_radius"] = np.array([radius], np.float64)
        L.nmslib_index_destroy(h)
    out["range_ext_ids"] = (np.arange(300, dtype=np.int32) * 2 + 9)

    path = os.path.join(HERE, "golden_v2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
