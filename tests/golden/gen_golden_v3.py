#!/usr/bin/env python3
"""Third fixture file from the REAL reference (oracle/_ref): full-size parity with INDEPENDENT queries.

    make -C oracle ref && python3 tests/golden/gen_golden_v3.py

The reference's own sequential scan (seq_search) of the BASELINE data sets at full size, for queries that are NOT
base rows (the bench's sets: S-lowrank seeds 42/43; S-sift-like seeds 44/45):
  * C2: 1M x 128 f32 l2, k = 10, the first 64 of the 1024 queries;
  * C4: 1M x 128 u8 l2sqr_sift, k = 100, the first 32 of the 4096 queries.
Only ids and distances are stored (inputs are regenerated from their seeds and pinned by SHA-256)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import refio  # noqa: E402


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def main():
    assert refio.HAVE_REF, "build oracle/_ref first: make -C oracle ref"
    out = {}
    X, Q = refio.s_lowrank(1_000_000, 128, 42), refio.s_lowrank(1024, 128, 43)
    out["c2_base_sha"], out["c2_queries_sha"] = sha(X), sha(Q)
    ids, d, _, _, _ = refio.run_ref_driver("l2", "seq_search", X, Q[:64], 10, threads=8)
    out["c2_ids"], out["c2_dists"] = ids, d
    U, UQ = refio.s_sift_like(1_000_000, 44), refio.s_sift_like(4096, 45)
    out["c4_base_sha"], out["c4_queries_sha"] = sha(U), sha(UQ)
    ids, d, _, _, _ = refio.run_ref_driver("l2sqr_sift", "seq_search", U, UQ[:32], 100, threads=8)
    out["c4_ids"], out["c4_dists"] = ids, d
    np.savez_compressed(os.path.join(HERE, "golden_v3.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
