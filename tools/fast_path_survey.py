#!/usr/bin/env python3
"""Which scan serves which data: exact scan of 1M x D float rows, batch 1024, over spaces / k / data shapes.
Prints per case: ms per batch (host entry, median of 5), query tiles, split-product tiles, fallback tiles.
    python3 tools/fast_path_survey.py [--n 1000000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nmslib_zig_amd as nz  # noqa: E402
from tests import refio  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--only", default="", help="comma-separated data set names")
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    datasets = {
        "gauss128": lambda n: refio.s_gauss(n, 128, 11),
        "lowrank128": lambda n: refio.s_lowrank(n, 128, 12),
        "gauss32": lambda n: refio.s_gauss(n, 32, 13),
        "siftlike128f": lambda n: refio.s_sift_like(n, 14).astype(np.float32),
        "offset128": lambda n: (refio.s_gauss(n, 128, 15) + np.float32(30.0)),
        "lowrank768": lambda n: refio.s_lowrank(n, 768, 16),
        "gauss256": lambda n: refio.s_gauss(n, 256, 17),
    }
    if a.only:
        datasets = {k: v for k, v in datasets.items() if k in a.only.split(",")}
    for dname, gen in datasets.items():
        X = gen(a.n)
        Q = gen(a.nq + 7)[7:]
        for space in ("l2", "cosinesimil", "negdotprod"):
            idx = nz.Index(space, "seq_search")
            idx.addDenseBatch(X)
            idx.buildIndex()
            for k in (1, 10, 100):
                idx.knnQueryBatch(Q, k)
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    idx.knnQueryBatch(Q, k)
                    ts.append(time.perf_counter() - t0)
                st = idx.stats()
                print(f"{dname:14s} {space:12s} k={k:<4d} {np.median(ts) * 1e3:7.3f} ms  path {st['last_path']} tiles {st['fast_tiles']} "
                      f"split-product {st['fast_tiles_precise']} fallback {st['fast_tiles_fallback']}", flush=True)
            idx.close()


if __name__ == "__main__":
    main()
