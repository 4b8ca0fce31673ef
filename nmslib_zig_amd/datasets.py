"""Synthetic data sets and NMSLIB's recall measure: what bench.py measures on and what the tests seed
their cases with.  Pure numpy; nothing here touches the oracle or the GPU."""
import numpy as np


def s_lowrank(n, dim, seed, rank=16, noise=0.1, basis_seed=42):
    """x = A z + noise*eps with A fixed by basis_seed: low intrinsic dimension, recall-friendly."""
    A = np.random.default_rng(basis_seed).standard_normal((dim, rank)).astype(np.float32)
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, rank)).astype(np.float32)
    e = rng.standard_normal((n, dim)).astype(np.float32)
    return (z @ A.T + noise * e).astype(np.float32)


def s_gauss(n, dim, seed):
    return np.random.default_rng(seed).standard_normal((n, dim)).astype(np.float32)


def s_sift_like(n, seed, dim=128):
    """uint8 descriptors, |N(0,1)|*40 rescaled to ||x||~512: many integer-distance ties."""
    rng = np.random.default_rng(seed)
    x = np.abs(rng.standard_normal((n, dim))) * 40.0
    x *= 512.0 / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def approx_equal_ulps(a, b, ulps=4):
    """utils.h:195 ApproxEqual: within 4 ULPs (floats), exact for ints."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib) <= ulps


def recall_nmslib(approx_ids, exact_ids, exact_dists, k, integer=False):
    """NMSLIB's recall (eval_results.h:122-130,176-183; eval_metrics.h:113-127): the exact set is
    the true top-k extended by everything ApproxEqual to the k-th distance.  exact_* must hold
    MORE than k entries per query (ascending) so the tie extension is visible."""
    tot = 0.0
    nq = len(approx_ids)
    for q in range(nq):
        ed = np.asarray(exact_dists[q])
        kk = min(k, len(ed))
        kth = ed[kk - 1]
        if integer:
            ext = ed == kth
        else:
            ext = approx_equal_ulps(ed, np.full_like(ed, kth))
        m = kk
        while m < len(ed) and ext[m]:
            m += 1
        exact = set(int(x) for x in exact_ids[q][:m])
        got = set(int(x) for x in approx_ids[q][:k] if x >= 0)
        tot += len(got & exact) / float(min(k, len(exact)))
    return tot / nq
