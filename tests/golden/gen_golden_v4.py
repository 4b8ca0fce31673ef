#!/usr/bin/env python3
"""Fourth fixture file from the REAL reference (oracle/_ref): HNSW construction options added in round 3.

    make -C oracle ref && python3 tests/golden/gen_golden_v4.py

  * delaunay_type 0..3 x post 0..2 (hnsw.cc:251-330, hnsw.h:82-256), single thread: the flattened graph of the
    reference's saved index (levels, level-0 lists, upper lists, maxM0 -- post=1 widens it) and SearchV1Merge results;
  * M = 64 (maxM0 = 128: lists longer than two words per lane), single thread: graph + results for both algorithms.
The rows are regenerated from their seeds (refio.s_gauss: bit-identical on every machine) and pinned by SHA-256.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import refio  # noqa: E402

COMBOS = [(d, p) for d in (0, 1, 2, 3) for p in (0, 1, 2) if not (p == 0 and d in (0, 2))]


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def inputs_opts():
    return refio.s_gauss(1200, 16, seed=881), refio.s_gauss(24, 16, seed=882)


def inputs_m64():
    return refio.s_gauss(1500, 12, seed=883), refio.s_gauss(24, 12, seed=884)


def store(out, tag, path, ids, d):
    P = refio.parse_optimized_index(path)
    out[f"{tag}_meta"] = np.array([P["maxlevel"], P["enterpoint"], P["maxM"], P["maxM0"]], np.int64)
    for key in ("levels", "links0", "up_off", "up_links"):
        out[f"{tag}_{key}"] = P[key]
    out[f"{tag}_ids"], out[f"{tag}_dists"] = ids, d


def main():
    assert refio.HAVE_REF, "build oracle/_ref first: make -C oracle ref"
    out = {}
    tmp = tempfile.mkdtemp(prefix="golden4_")
    base, qs = inputs_opts()
    out["opts_base_sha"], out["opts_queries_sha"] = sha(base), sha(qs)
    for dl, post in COMBOS:
        path = os.path.join(tmp, f"d{dl}p{post}.idx")
        ids, d, _, _, _ = refio.run_ref_driver("l2", "hnsw", base, qs, 10,
                                               f"M=6,efConstruction=40,indexThreadQty=1,delaunay_type={dl},post={post}",
                                               "efSearch=40", save=path)
        store(out, f"d{dl}p{post}", path, ids, d)
        print(dl, post, "maxM0", int(out[f"d{dl}p{post}_meta"][3]), flush=True)
    base, qs = inputs_m64()
    out["m64_base_sha"], out["m64_queries_sha"] = sha(base), sha(qs)
    path = os.path.join(tmp, "m64.idx")
    ids, d, _, _, _ = refio.run_ref_driver("l2", "hnsw", base, qs, 10, "M=64,efConstruction=150,indexThreadQty=1",
                                           "efSearch=60", save=path)
    store(out, "m64", path, ids, d)
    ids, d, _, _, _ = refio.run_ref_driver("l2", "hnsw", base, qs, 10, "M=64,efConstruction=150,indexThreadQty=1",
                                           "efSearch=60,algoType=old")
    out["m64_old_ids"], out["m64_old_dists"] = ids, d
    np.savez_compressed(os.path.join(HERE, "golden_v4.npz"), **out)
    print("wrote golden_v4.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
