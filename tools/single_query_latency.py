#!/usr/bin/env python3
"""Latency of ONE query per call through the reference's per-query entry (nmslib_knn_query_fill, what
lib.zig's knnQuery issues), 1M x 128 l2: brute force and HNSW.  Prints median / p95 in microseconds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmslib_zig_amd as nz
from tests import refio

X, Q = refio.s_lowrank(1_000_000, 128, 42), refio.s_lowrank(200, 128, 43)
for method, params in (("seq_search", {}), ("hnsw", dict(M=16, efConstruction=200))):
    idx = nz.Index("l2", method)
    idx.addDenseBatch(X)
    idx.buildIndex(**params)
    if method == "hnsw":
        idx.setQueryTimeParams(efSearch=128)
    for q in Q[:20]:
        idx.knnQuery(q, 10)
    t = []
    for q in Q:
        t0 = time.perf_counter()
        idx.knnQuery(q, 10)
        t.append(time.perf_counter() - t0)
    idx.kernel_timing(enable=True)
    for q in Q[:50]:
        idx.knnQuery(q, 10)
    kms, kn = idx.kernel_timing(enable=False, collect=True)
    t = np.array(t) * 1e6
    print(f"{method:10s} dominant kernel alone: {kms / max(1, kn) * 1e3:8.1f} us per call")
    print(f"{method:10s} one query per call: median {np.median(t):8.1f} us   p95 {np.percentile(t, 95):8.1f} us")
    idx.close()
