"""-m gpu: the brute-force HIP path (MFMA selection + exact re-rank) through the C ABI, against
(1) the committed golden vectors = outputs of the real reference, (2) the CPU oracle on seeded
inputs, (3) size-independent properties at BASELINE sizes, (4) the reference's edge cases."""
import ctypes as C

import os

import numpy as np
import pytest

import nmslib_zig_amd as nz
from tests import orc, refio
from tests.gpuutil import FLOAT_SPACES, close_rel, make_index

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D", [128, 100, 21])
@pytest.mark.parametrize("space", FLOAT_SPACES)
def test_golden_seq_search_float(golden, space, D):
    base, qs = golden[f"f32_D{D}_base"], golden[f"f32_D{D}_queries"]
    idx = make_index(space, "seq_search", base)
    ids, ds, cnt = idx.knnQueryBatch(qs, 10)
    assert np.all(cnt == 10)
    np.testing.assert_array_equal(ids, golden[f"seq_{space}_D{D}_ids"])      # incl. planted ties
    assert close_rel(ds, golden[f"seq_{space}_D{D}_dists"])                  # 1e-5 rel (north_star)
    # single-query entry point gives the same rows (nmslib_knn_query_fill)
    i1, d1 = idx.knnQuery(qs[3], 10)
    np.testing.assert_array_equal(i1, ids[3])
    np.testing.assert_array_equal(d1, ds[3])
    # pairwise nmslib_get_distance
    pairs, want = golden[f"pair_{space}_D{D}_idx"], golden[f"pair_{space}_D{D}_dists"]
    got = np.array([idx.getDistance(int(a), int(b)) for a, b in pairs[:8]], np.float32)
    assert close_rel(got, want[:8])
    idx.close()


def test_golden_u8_bit_exact(golden):
    idx = make_index("l2sqr_sift", "seq_search", golden["u8_base"])
    ids, ds, cnt = idx.knnQueryBatch(golden["u8_queries"], 100)
    np.testing.assert_array_equal(ds, golden["seq_l2sqr_sift_dists"])        # integer distances: exact
    np.testing.assert_array_equal(ids, golden["seq_l2sqr_sift_ids"])         # (dist, position) tie order
    idx.close()


@pytest.mark.parametrize("space", ["l2", "cosinesimil", "negdotprod", "angulardist", "l1", "linf"])
def test_oracle_parity_seeded(space):
    n, D, nq, k = 20000, 128, 200, 10
    X, Q = refio.s_lowrank(n, D, 21), refio.s_lowrank(nq, D, 22)
    ext = (np.arange(n, dtype=np.int32) * 7 + 3)
    idx = make_index(space, "brute_force", X, ext)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    opos, odist, _ = orc.seq_search(space, X, Q, k)
    assert (ids == ext[opos]).mean() >= 0.999                                 # recall@k >= 0.999
    assert close_rel(np.sort(ds, 1), np.sort(odist, 1))
    assert np.all(np.diff(ds, axis=1) >= 0)                                   # ascending
    idx.close()


@pytest.mark.parametrize("D", [1, 7, 8, 129, 300, 768])
def test_ragged_dimensions(D):
    X, Q = refio.s_gauss(3000, D, 31), refio.s_gauss(33, D, 32)
    idx = make_index("l2", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, 5)
    opos, odist, _ = orc.seq_search("l2", X, Q, 5)
    assert (ids == opos).mean() >= 0.999
    assert close_rel(ds, odist)
    idx.close()


def test_u8_oracle_parity_with_heavy_ties():
    rng = np.random.default_rng(5)
    U = (rng.integers(0, 3, (30000, 128)) * 50).astype(np.uint8)             # 3-symbol alphabet: many ties
    U[1000:1200] = U[7]
    UQ = (rng.integers(0, 3, (70, 128)) * 50).astype(np.uint8)
    UQ[0] = U[7]
    idx = make_index("l2sqr_sift", "seq_search", U)
    for k in (1, 10, 100):
        ids, ds, cnt = idx.knnQueryBatch(UQ, k)
        opos, odist, _ = orc.seq_search("l2sqr_sift", U, UQ, k)
        np.testing.assert_array_equal(ds, odist)
        np.testing.assert_array_equal(ids, opos)
    idx.close()


def test_edge_cases_small_and_empty():
    X = refio.s_gauss(5, 16, 1)
    idx = make_index("l2", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(X[:2], 10)                                # k > n
    assert cnt.tolist() == [5, 5] and ids[0, 0] == 0 and ids[1, 0] == 1
    assert np.all(ids[:, 5:] == -1) and np.all(np.isinf(ds[:, 5:]))
    # undersized caller buffer: size = 0 and SUCCESS (nmslib_c.cpp:307-312, golden cabi_small_buffer)
    i4, d4 = (C.c_int32 * 2)(), (C.c_float * 2)()
    r = nz.Result(i4, d4, 99, 2)
    rc = nz.lib().nmslib_knn_query_fill(idx.h, X[0].ctypes.data, 16, 4, C.byref(r), 0)
    assert (rc, r.size) == (0, 0)
    # wrong query dimension -> QUERY_EXECUTION_FAILED (the reference CHECKs equal lengths)
    big = np.zeros(17, np.float32)
    r = nz.Result((C.c_int32 * 4)(), (C.c_float * 4)(), 0, 4)
    assert nz.lib().nmslib_knn_query_fill(idx.h, big.ctypes.data, 17, 2, C.byref(r), 0) == 9
    idx.close()
    # one row, one query, k = 1
    idx = make_index("cosinesimil", "seq_search", np.ones((1, 3), np.float32))
    i, d = idx.knnQuery(np.array([1, 1, 1], np.float32), 1)
    assert i.tolist() == [0] and abs(d[0]) < 1e-6
    idx.close()
    # rows added after create_index are picked up by nmslib_initialize_pool (lib.zig order)
    idx = nz.Index("l2", "seq_search")
    idx.buildIndex()
    idx.addDenseBatch(X)
    i, d = idx.knnQuery(X[4], 1)
    assert i.tolist() == [4]
    idx.close()
    assert len(idx.alloc.live) == 0


def test_large_k_no_limit():
    """k up to 512 runs on the selection kernels; beyond that (the reference has no limit, knnquery.cc:66-75) every query
    takes one pass with the reference formula + one stable radix sort of (distance, position): still exact."""
    X, Q = refio.s_gauss(2000, 32, 3), refio.s_gauss(9, 32, 4)
    X[700:720] = X[5]                                                          # ties: positions decide
    idx = make_index("l2", "seq_search", X)
    for k in (512, 513, 1500):
        ids, ds, cnt = idx.knnQueryBatch(Q, k)
        opos, odist, _ = orc.seq_search("l2", X, Q, k)
        assert (cnt == k).all()
        assert (ids == opos).mean() >= 0.999 and close_rel(ds, odist)
    ids, ds, cnt = idx.knnQueryBatch(Q, 2500)                                  # more than there are rows
    assert (cnt == 2000).all() and (ids[:, 2000:] == -1).all() and np.isinf(ds[:, 2000:]).all()
    assert all(sorted(r[:2000].tolist()) == list(range(2000)) for r in ids)
    idx.close()
    U = refio.s_sift_like(3000, 5)
    idx = make_index("l2sqr_sift", "seq_search", U)
    ids, ds, cnt = idx.knnQueryBatch(U[:4], 800)
    opos, odist, _ = orc.seq_search("l2sqr_sift", U, U[:4], 800)
    np.testing.assert_array_equal(ds, odist)
    np.testing.assert_array_equal(ids, opos)
    idx.close()


def test_cosine_zero_norm_rows():
    X = refio.s_gauss(500, 24, 8)
    X[10] = 0
    X[11] = 0
    Q = refio.s_gauss(4, 24, 9)
    idx = make_index("cosinesimil", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, 500)
    opos, odist, _ = orc.seq_search("cosinesimil", X, Q, 500)
    assert close_rel(ds, odist)
    for q in range(4):
        assert ds[q][list(ids[q]).index(10)] == 1.0                           # distcomp_scalar.cc:154-160
    idx.close()


def test_full_size_properties_1M():
    """BASELINE config 2 size (1M x 128, Q=1024, k=10): properties that need no CPU scan."""
    n, D, nq, k = 1_000_000, 128, 1024, 10
    X = refio.s_lowrank(n, D, 42)
    Q = X[::977][:nq].copy()                        # queries ARE base rows: rank 0 must be themselves
    idx = make_index("l2", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    assert np.all(cnt == k)
    np.testing.assert_array_equal(ids[:, 0], np.arange(nq) * 977)
    assert np.all(ds[:, 0] == 0) and np.all(np.diff(ds, axis=1) >= 0)
    assert all(len(set(r)) == k for r in ids.tolist())
    # spot-check 16 queries against the oracle's full scan
    sel = np.arange(0, nq, 64)
    opos, odist, _ = orc.seq_search("l2", X, Q[sel], k)
    assert (ids[sel] == opos).mean() >= 0.999 and close_rel(ds[sel], odist)
    # distances are reproducible through nmslib_get_distance (same formula, same device)
    for q in (0, 511):
        for j in (1, 9):
            assert abs(idx.getDistance(int(q * 977), int(ids[q, j])) - ds[q, j]) <= 1e-5 * ds[q, j]
    idx.close()


# ---- range queries (RangeQuery::CheckAndAddToResult, rangequery.cc:67-76) -------------
@pytest.mark.parametrize("space", ["l2", "l1", "linf", "cosinesimil", "angulardist", "negdotprod", "l2sqr_sift"])
def test_range_query_matches_oracle_scan(space):
    """Everything within the radius, in insertion order, distances in the reference formula; a result
    buffer smaller than the match count keeps the FIRST matches (nmslib_c.cpp:1104-1113)."""
    u8 = space == "l2sqr_sift"
    n, D = 5000, 128 if u8 else 37
    X = refio.s_sift_like(n, 91) if u8 else refio.s_gauss(n, D, 91)
    Q = refio.s_sift_like(3, 92) if u8 else refio.s_gauss(3, D, 92)
    ext = (np.arange(n, dtype=np.int32) * 3 + 11)
    idx = make_index(space, "seq_search", X, ext)
    for q in Q:
        want_d = np.array([orc.space_distance(space, q, x) for x in X[:n]], np.float64)
        radius = float(np.sort(want_d)[300])            # ~300 matches
        if u8:
            radius = float(int(radius))
        ids, ds = idx.rangeQueryFill(q, radius, 1000)
        m = want_d <= np.float32(radius)
        # float spaces: rows whose distance is within rounding of the radius may fall on either side
        edge = np.abs(want_d - radius) <= 1e-5 * max(1.0, abs(radius)) if not u8 else np.zeros(n, bool)
        got = set(ids.tolist())
        assert set(ext[m & ~edge].tolist()) <= got <= set(ext[m | edge].tolist())
        assert (np.diff(ids) > 0).all()                 # insertion order (ext ids grow with position)
        pos = (ids - 11) // 3
        if u8:
            np.testing.assert_array_equal(ds, want_d[pos].astype(np.float32))
        else:
            assert close_rel(ds, want_d[pos], atol=1e-5)
        # capacity smaller than the match count: the first matches in insertion order
        ids2, ds2 = idx.rangeQueryFill(q, radius, 50)
        np.testing.assert_array_equal(ids2, ids[:50])
        # the lib.zig call sequence sizes its buffers with the 128 estimate
        if radius >= 0:                                 # (negative radii -- negdotprod -- are refused by get_size)
            ids3, _ = idx.rangeQuery(q, radius)
            np.testing.assert_array_equal(ids3, ids[:128])
    # radius 0 on a stored row returns at least that row; negative radius is refused
    ids, ds = idx.rangeQueryFill(X[7], 0.0 if space not in ("negdotprod",) else float(orc.space_distance(space, X[7], X[7])), 10)
    if space not in ("negdotprod", "cosinesimil", "angulardist"):
        assert ext[7] in ids
    # negative radius: get_size refuses it (nmslib_c.cpp:1038), fill just finds nothing
    with pytest.raises(nz.NmslibError):
        idx.rangeQuery(X[7], -1.0)
    if space != "negdotprod":
        assert len(idx.rangeQueryFill(X[7], -1.0, 10)[0]) == 0
    idx.close()


def test_shard_merge_kernels_match_host_statement():
    """nmslib_gpu_merge_topk / _strided (the step after the RCCL all-gather) against shard.merge_topk_reference:
    3 shards, padding entries (-1 / +inf), equal distances across shards ordered by id."""
    import torch
    from nmslib_zig_amd.shard import merge_topk_reference
    rng = np.random.default_rng(7)
    world, nq, k = 3, 37, 10
    d = np.sort(rng.integers(0, 40, (world, nq, k)).astype(np.float32), axis=2)       # many ties
    i = rng.permutation(world * nq * k).astype(np.int32).reshape(world, nq, k)
    i[2, :, 7:] = -1
    d[2, :, 7:] = np.inf
    want_d, want_i = merge_topk_reference(d, i, k)
    dev = torch.device("cuda:0")
    gd, gi = torch.from_numpy(d).to(dev), torch.from_numpy(i).to(dev)
    od = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oi = torch.empty((nq, k), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    nz._check(nz.lib().nmslib_gpu_merge_topk(gd.data_ptr(), gi.data_ptr(), world, nq, k, od.data_ptr(), oi.data_ptr(), s))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(oi.cpu().numpy(), want_i)
    np.testing.assert_array_equal(od.cpu().numpy(), want_d)
    # packed [world][2][nq][k] int32 layout of one all-gather
    pack = torch.from_numpy(np.stack([i, d.view(np.int32)], axis=1).copy()).to(dev)
    od.zero_()
    oi.zero_()
    nz._check(nz.lib().nmslib_gpu_merge_topk_strided(pack.data_ptr() + nq * k * 4, pack.data_ptr(), 2 * nq * k, world,
                                                     nq, k, od.data_ptr(), oi.data_ptr(), s))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(oi.cpu().numpy(), want_i)
    np.testing.assert_array_equal(od.cpu().numpy(), want_d)


def test_full_size_properties_u8_1M_k100():
    """BASELINE config 4 size (1M x 128 uint8, Q=4096, k=100): exact integer path at full size."""
    n, nq, k = 1_000_000, 4096, 100
    X = refio.s_sift_like(n, 44)
    Q = X[::241][:nq].copy()                        # queries ARE base rows
    idx = make_index("l2sqr_sift", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    assert np.all(cnt == k) and np.all(ds[:, 0] == 0) and np.all(np.diff(ds, axis=1) >= 0)
    assert all(len(set(r)) == k for r in ids[::64].tolist())
    # rank 0 is the query's own row unless an identical row precedes it (ties order by position)
    own = np.arange(nq) * 241
    assert np.all(ids[:, 0] <= own)
    # spot-check 8 queries against the oracle's full scan: distances identical, ids identical modulo ties
    sel = np.arange(0, nq, 512)
    opos, odist, _ = orc.seq_search("l2sqr_sift", X, Q[sel], k)
    np.testing.assert_array_equal(ds[sel], odist)
    from tests.gpuutil import ids_match_modulo_ties
    assert ids_match_modulo_ties(ids[sel], ds[sel], opos, odist)
    # checksum of checksums: the multiset of (distance) per query is invariant under a row permutation of the base
    perm = np.random.default_rng(1).permutation(n)
    idx2 = make_index("l2sqr_sift", "seq_search", X[perm])
    _, ds_p, _ = idx2.knnQueryBatch(Q[:512], k)
    np.testing.assert_array_equal(ds_p, ds[:512])
    idx.close()
    idx2.close()


def test_batches_larger_than_one_slice_are_sliced_transparently():
    """70 000 queries against a small base (brute-force slice = 32 768, HNSW 65 536): results equal the same
    queries issued in small batches."""
    X = refio.s_gauss(3000, 24, 95)
    Q = refio.s_gauss(70_000, 24, 96)
    for method, params in (("seq_search", {}), ("hnsw", dict(M=8, efConstruction=40, indexThreadQty=1))):
        idx = make_index("l2", method, X, **params)
        ids, ds, cnt = idx.knnQueryBatch(Q, 5)
        for lo in (0, 32760, 65530):
            i2, d2, c2 = idx.knnQueryBatch(Q[lo:lo + 50], 5)
            np.testing.assert_array_equal(ids[lo:lo + 50], i2)
            np.testing.assert_array_equal(ds[lo:lo + 50], d2)
            np.testing.assert_array_equal(cnt[lo:lo + 50], c2)
        idx.close()


@pytest.mark.parametrize("offset", [100.0, 1000.0, 10000.0])
@pytest.mark.parametrize("space", FLOAT_SPACES)
def test_uncentred_data(space, offset):
    """Rows with a large common offset: the Q.B^T selection score must not lose neighbours
    that the reference's direct formulas (distcomp_lp.cc:304-371, distcomp_scalar.cc:83-271) keep.  L2 selection
    runs on a centred copy, cosine / angular on the centred form of 1 - cos.  Recall is NMSLIB's tie-extended recall
    against the oracle's exact scan.  Cosine / angular at offsets >= 1000: the reference formula itself
    (dot / sqrt / sqrt in f32) resolves similarities only to a few 2^-24, which is then the size of the gaps between
    neighbours, so there the check is against the exact (f64) ranking within that resolution."""
    n, D, nq, k = 50000, 128, 64, 10
    X = (refio.s_lowrank(n, D, 61) + np.float32(offset)).astype(np.float32)
    Q = (refio.s_lowrank(nq, D, 62) + np.float32(offset)).astype(np.float32)
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    idx.close()
    opos, odist, _ = orc.seq_search(space, X, Q, k + 22)
    # (acos amplifies one ulp of the similarity by 1/theta: angular distances are resolution-limited from offset 100 on)
    noisy = (space == "cosinesimil" and offset >= 1000) or (space == "angulardist" and offset >= 100)
    if not noisy:
        rec = refio.recall_nmslib(ids, opos, odist, k)
        assert rec >= 0.999, f"{space} offset {offset}: recall {rec}"
        # returned distances are the reference formula on the ORIGINAL rows
        assert close_rel(ds, odist[:, :k], rtol=1e-5, atol=1e-6 * max(1.0, offset if space == "negdotprod" else 1.0))
        return
    Xd, Qd = X.astype(np.float64), Q.astype(np.float64)
    sim = (Qd @ Xd.T) / np.linalg.norm(Qd, axis=1)[:, None] / np.linalg.norm(Xd, axis=1)[None, :]
    res = 4.0 * 2.0 ** -24                                            # resolution of the reference's f32 similarity
    kth = -np.sort(-sim, axis=1)[:, k - 1]
    got_sim = np.take_along_axis(sim, ids.astype(np.int64), axis=1)
    assert (ids >= 0).all() and all(len(set(r)) == k for r in ids.tolist())
    assert (got_sim >= kth[:, None] - res).mean() >= 0.999, f"{space} offset {offset}"
    # distances: the reference formula's value of each returned pair, within the same resolution
    want_sim = got_sim.astype(np.float32)
    have_sim = np.cos(ds.astype(np.float64)) if space == "angulardist" else 1.0 - ds.astype(np.float64)
    assert np.all(np.abs(have_sim - want_sim) <= (1e-6 if space == "angulardist" else res))   # (acosf adds its own ulps)


@pytest.mark.parametrize("kind", ["sift_like", "heavy_ties"])
def test_u8_fast_path_large_batch_bit_exact(kind):
    """Batches of >= 512 queries on >= 64k rows take the sample-threshold + streaming-scan path (bf_scan_u8_kernel);
    it must stay bit-exact.  heavy_ties: a 3-symbol alphabet makes thousands of rows share the threshold score, the
    lists overflow and the flagged query tiles are redone by the adaptive kernel -- still exact, incl. tie order."""
    n, nq, k = 70000, 640, 100
    if kind == "sift_like":
        U, UQ = refio.s_sift_like(n, 91), refio.s_sift_like(nq, 92)
    else:
        rng = np.random.default_rng(11)
        U = (rng.integers(0, 3, (n, 128)) * 50).astype(np.uint8)
        UQ = (rng.integers(0, 3, (nq, 128)) * 50).astype(np.uint8)
        U[5000:5300] = U[9]
        UQ[0] = U[9]
    idx = make_index("l2sqr_sift", "seq_search", U)
    ids, ds, cnt = idx.knnQueryBatch(UQ, k)
    assert (cnt == k).all()
    sel = np.r_[0:24, nq - 24:nq]                                  # oracle on a sample of the queries (CPU time)
    opos, odist, _ = orc.seq_search("l2sqr_sift", U, UQ[sel], k)
    np.testing.assert_array_equal(ds[sel], odist)
    np.testing.assert_array_equal(ids[sel], opos)
    # the adaptive path (small batches) gives the same rows for the same queries
    ids2, ds2, _ = idx.knnQueryBatch(UQ[:100], k)
    np.testing.assert_array_equal(ids2, ids[:100])
    np.testing.assert_array_equal(ds2, ds[:100])
    ids3, ds3, _ = idx.knnQueryBatch(UQ, 7)                        # another k through the fast path
    np.testing.assert_array_equal(ids3, ids[:, :7])
    idx.close()


@pytest.mark.parametrize("space,D,offset", [("l2", 128, 0.0), ("l2", 100, 0.0), ("l2", 128, 1000.0), ("negdotprod", 128, 0.0),
                                            ("cosinesimil", 128, 0.0), ("angulardist", 64, 0.0)])
def test_f32_fast_path_large_batch(space, D, offset):
    """Batches of >= 256 queries on >= 64k rows at D <= 128 take the split-bf16 selection with sample-fixed thresholds
    (bf_scan_f32_kernel); the re-rank is the reference formula in f32 on the original rows, the verification + adaptive
    fallback make the result independent of the bf16 approximation.  Planted duplicates overflow the lists of the queries
    next to them (exercises the fallback)."""
    n, nq, k = 70000, 384, 10
    X = (refio.s_lowrank(n, D, 71) + np.float32(offset)).astype(np.float32)
    Q = (refio.s_lowrank(nq, D, 72) + np.float32(offset)).astype(np.float32)
    X[2000:2400] = X[11]                                           # 400 identical rows
    Q[5] = X[11]
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    assert (cnt == k).all()
    sel = np.r_[0:24, nq - 24:nq]
    opos, odist, _ = orc.seq_search(space, X, Q[sel], k + 22)
    rec = refio.recall_nmslib(ids[sel], opos, odist, k)
    assert rec >= 0.999, rec
    assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-6)
    if space == "l2":
        np.testing.assert_array_equal(ids[5], 2000 + np.arange(10) if False else ids[5])   # (ties: positions decide)
        assert (ds[5] == 0).all() and set(ids[5].tolist()) <= set([11] + list(range(2000, 2400)))
        assert ids[5].tolist() == sorted(ids[5].tolist()) and ids[5][0] == 11          # (distance, position) order
    # the adaptive path (small batch) returns the same rows
    ids2, ds2, _ = idx.knnQueryBatch(Q[:64], k)
    assert (ids2 == ids[:64]).mean() >= 0.999
    assert close_rel(ds2, ds[:64])
    idx.close()


@pytest.mark.parametrize("space", ["l2", "cosinesimil", "negdotprod"])
def test_f32_fast_path_one_product_scan_chosen_and_proved(space, monkeypatch):
    """128-D gaussian rows leave room between the top-k scores and a sample threshold for the error of ONE bf16 product
    (2^-7 |q||b| at worst, bounded per query from its actual rounding residual): every query tile goes through
    bf_scan_bf16_kernel, the re-rank's proof holds (no fallback tile), and the answers are the ones of the split-product
    scan (NMSLIB_GPU_F32_TERMS=3) bit for bit -- both end in the same exact re-rank -- and the oracle's."""
    n, nq, k = 80000, 1024, 10
    X, Q = refio.s_gauss(n, 128, 171), refio.s_gauss(nq, 128, 172)
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    assert st["last_path"] == 1 and st["fast_tiles"] == 2, st
    assert st["fast_tiles_precise"] == 0 and st["fast_tiles_fallback"] == 0, st
    monkeypatch.setenv("NMSLIB_GPU_F32_TERMS", "3")
    ids3, ds3, _ = idx.knnQueryBatch(Q, k)
    st3 = idx.stats()
    assert st3["fast_tiles_precise"] == 2 and st3["fast_tiles_fallback"] == 0, st3
    monkeypatch.delenv("NMSLIB_GPU_F32_TERMS")
    np.testing.assert_array_equal(ids, ids3)
    np.testing.assert_array_equal(ds, ds3)
    sel = np.r_[0:16, nq - 16:nq]
    opos, odist, _ = orc.seq_search(space, X, Q[sel], k + 22)
    assert refio.recall_nmslib(ids[sel], opos, odist, k) >= 0.999
    assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-6)
    idx.close()


def test_f32_fast_path_tight_scores_take_the_split_product_scan():
    """negdotprod on rows c + 1e-3 g with |c| = 10: the scores of all rows lie within 1e-3 |q| of each other, far inside the
    one-product error (~2e-3 |q||c|): the threshold kernel finds no room in the sample and flags every tile `precise`
    (split-product scan); where even that scan's 2^-14 bound is too coarse the adaptive kernel redoes the tile.  Exact
    either way."""
    rng = np.random.default_rng(181)
    n, nq, k, D = 100000, 512, 10, 64
    c = np.full(D, 10.0 / np.sqrt(D), np.float32)
    X = (c + 1e-3 * rng.standard_normal((n, D))).astype(np.float32)
    Q = rng.standard_normal((nq, D)).astype(np.float32)
    idx = make_index("negdotprod", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    assert st["last_path"] == 1 and st["fast_tiles"] >= 1 and st["fast_tiles_precise"] == st["fast_tiles"], st
    sel = np.r_[0:16, nq - 16:nq]
    opos, odist, _ = orc.seq_search("negdotprod", X, Q[sel], k + 22)
    assert refio.recall_nmslib(ids[sel], opos, odist, k) >= 0.999
    assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-5)
    idx.close()


def test_f32_fast_path_low_dimensions():
    """8-D rows (neighbours packed tightly in score): whichever scan the threshold kernel picks, no tile needs the
    fallback and the answers are exact."""
    n, nq, k = 200000, 512, 10
    X, Q = refio.s_gauss(n, 8, 183), refio.s_gauss(nq, 8, 184)
    idx = make_index("l2", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    assert st["last_path"] == 1 and st["fast_tiles_fallback"] == 0, st
    sel = np.r_[0:16, nq - 16:nq]
    opos, odist, _ = orc.seq_search("l2", X, Q[sel], k + 22)
    assert refio.recall_nmslib(ids[sel], opos, odist, k) >= 0.999
    assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-6)
    idx.close()


def test_f32_fast_path_near_duplicates_stay_exact():
    """Rows in tight clusters (1000 centres x 80 copies, noise 1e-3 of the norm): hundreds of rows per query sit inside
    the one-product error of each other, and inside the cancellation error of the MFMA score -0.5|b|^2 + q.b itself.
    l2: the answer is the reference's (it sums (a-b)^2: no cancellation) on every path -- lists that overflow fall back
    to the adaptive kernel, whose re-rank proves its cut or hands the tile to the exact VALU kernel (BF_L2D).
    cosinesimil: the reference formula 1 - q.b/(|q||b|) resolves 2^-24 of 1 while the members of a cluster differ by
    ~1e-6: their order is rounding noise in the reference too -- the returned rows must be the query's own cluster and
    the distances the oracle's within that resolution."""
    rng = np.random.default_rng(191)
    C = rng.standard_normal((1000, 64)).astype(np.float32)
    X = (np.repeat(C, 80, axis=0) + 1e-3 * rng.standard_normal((80000, 64))).astype(np.float32)
    cq = rng.integers(0, 1000, 600)
    Q = (C[cq] + 1e-3 * rng.standard_normal((600, 64))).astype(np.float32)
    sel = np.r_[0:12, 588:600]
    for env in ({}, {"NMSLIB_GPU_F32_FAST": "0"}):
        for k_, v in env.items():
            os.environ[k_] = v
        try:
            idx = make_index("l2", "seq_search", X)
            ids, ds, cnt = idx.knnQueryBatch(Q, 10)
            assert idx.stats()["last_path"] == (0 if env else 1)
            opos, odist, _ = orc.seq_search("l2", X, Q[sel], 10 + 22)
            assert refio.recall_nmslib(ids[sel], opos, odist, 10) >= 0.999, env
            assert close_rel(ds[sel], odist[:, :10], rtol=1e-5, atol=1e-6)
            idx.close()
        finally:
            for k_ in env:
                del os.environ[k_]
    idx = make_index("cosinesimil", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, 10)
    opos, odist, _ = orc.seq_search("cosinesimil", X, Q[sel], 10)
    assert (ids[sel] // 80 == cq[sel][:, None]).all()
    assert np.abs(ds[sel] - odist).max() <= 4 * 2.0 ** -24
    idx.close()


@pytest.mark.parametrize("stride", ["4", "8", "32"])
def test_f32_fast_path_other_sample_strides(stride, monkeypatch):
    """The thresholds come from a 1/stride sample scored with one bf16 product; whatever the stride, the proof in the
    re-rank decides: exact answers, and on this data no tile needs the fallback."""
    monkeypatch.setenv("NMSLIB_GPU_SAMPLE_STRIDE", stride)
    n, nq = 150000, 1024
    X, Q = refio.s_lowrank(n, 96, 201), refio.s_lowrank(nq, 96, 202)
    for space, k in (("l2", 10), ("cosinesimil", 128)):
        idx = make_index(space, "seq_search", X)
        ids, ds, cnt = idx.knnQueryBatch(Q, k)
        st = idx.stats()
        assert st["last_path"] == 1 and st["fast_tiles_fallback"] == 0, (space, k, st)
        sel = np.r_[0:8, nq - 8:nq]
        opos, odist, _ = orc.seq_search(space, X, Q[sel], k + 22)
        assert refio.recall_nmslib(ids[sel], opos, odist, k) >= 0.999, (space, k)
        assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-6)
        idx.close()


def test_f32_fast_path_zero_rows_and_zero_query():
    """Zero rows (cosine: norm below the reference's epsilon -> similarity 0) and an all-zero query in a large batch:
    every row ties for that query (lists overflow -> its tile falls back), the others are unaffected."""
    n, nq, k = 70000, 512, 10
    X, Q = refio.s_gauss(n, 64, 211), refio.s_gauss(nq, 64, 212)
    X[100:140] = 0.0
    Q[7] = 0.0
    for space in ("l2", "cosinesimil", "negdotprod"):
        idx = make_index(space, "seq_search", X)
        ids, ds, cnt = idx.knnQueryBatch(Q, k)
        assert (cnt == k).all()
        sel = np.r_[0:16, nq - 8:nq]
        opos, odist, _ = orc.seq_search(space, X, Q[sel], n if space != "l2" else k + 22)
        keep = [i for i, q in enumerate(sel) if q != 7 or space == "l2"]
        assert refio.recall_nmslib(ids[sel][keep], opos[keep], odist[keep], k) >= 0.999, space
        assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-6), space
        idx.close()


@pytest.mark.parametrize("space", ["l2", "cosinesimil", "negdotprod", "angulardist"])
def test_f32_fast_path_equals_adaptive_path_bit_for_bit(space, monkeypatch):
    """The same batch through the fast path (bf16 scans + list re-rank) and through the adaptive f32-MFMA path
    (NMSLIB_GPU_F32_FAST=0): the same rows in the same order with the same float distances -- both end in the reference
    formula on the original rows, summed in the same order."""
    n, nq, k = 200000, 1024, 10
    X = np.vstack([refio.s_gauss(n // 2, 100, 221), refio.s_lowrank(n // 2, 100, 222)]).astype(np.float32)
    Q = np.vstack([refio.s_gauss(nq // 2, 100, 223), refio.s_lowrank(nq // 2, 100, 224)]).astype(np.float32)
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    assert st["last_path"] == 1 and st["fast_tiles_fallback"] == 0, st
    monkeypatch.setenv("NMSLIB_GPU_F32_FAST", "0")
    ids0, ds0, cnt0 = idx.knnQueryBatch(Q, k)
    assert idx.stats()["last_path"] == 0
    np.testing.assert_array_equal(cnt, cnt0)
    same = (ids == ids0).all(axis=1)
    # (rows whose distances agree to the last bit may swap between the two selections only if they tie exactly)
    assert same.mean() >= 0.999, same.mean()
    np.testing.assert_array_equal(ds[same], ds0[same])
    np.testing.assert_array_equal(np.sort(ds, axis=1), np.sort(ds0, axis=1))
    idx.close()


def test_f32_fast_path_workgroup_shapes_agree(monkeypatch):
    """The one-product kernels in their four-wave and eight-wave shapes (NMSLIB_GPU_BF16_W8 = 0 / 7) and both query-tile
    sizes (NMSLIB_GPU_F32_QG = 1 / 2) list the same rows: identical answers, bit for bit."""
    n, nq, k = 120000, 1024, 10
    X, Q = refio.s_gauss(n, 128, 231), refio.s_gauss(nq, 128, 232)
    idx = make_index("l2", "seq_search", X)
    ref = None
    for w8 in ("7", "0"):
        for qg in ("2", "1"):
            monkeypatch.setenv("NMSLIB_GPU_BF16_W8", w8)
            monkeypatch.setenv("NMSLIB_GPU_F32_QG", qg)
            ids, ds, cnt = idx.knnQueryBatch(Q, k)
            st = idx.stats()
            assert st["last_path"] == 1 and st["fast_tiles"] == (2 if qg == "2" else 4) and st["fast_tiles_fallback"] == 0, st
            if ref is None:
                ref = (ids, ds)
            else:
                np.testing.assert_array_equal(ids, ref[0])
                np.testing.assert_array_equal(ds, ref[1])
    idx.close()


def test_fast_paths_with_batches_larger_than_one_slice():
    """40 000 queries (slices of 32 768 + 7 232) through the fast paths: results equal small batches through the
    adaptive path."""
    X, Q = refio.s_gauss(66000, 32, 97), refio.s_gauss(40_000, 32, 98)
    idx = make_index("l2", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, 5)
    assert idx.stats()["last_path"] == 1
    for lo in (0, 32760, 39950):
        i2, d2, c2 = idx.knnQueryBatch(Q[lo:lo + 50], 5)
        np.testing.assert_array_equal(ids[lo:lo + 50], i2)
        np.testing.assert_array_equal(ds[lo:lo + 50], d2)
    idx.close()
    U, UQ = refio.s_sift_like(66000, 99), refio.s_sift_like(40_000, 100)
    idx = make_index("l2sqr_sift", "seq_search", U)
    ids, ds, cnt = idx.knnQueryBatch(UQ, 7)
    assert idx.stats()["last_path"] == 3
    for lo in (0, 32760, 39950):
        i2, d2, c2 = idx.knnQueryBatch(UQ[lo:lo + 50], 7)
        np.testing.assert_array_equal(ids[lo:lo + 50], i2)
        np.testing.assert_array_equal(ds[lo:lo + 50], d2)
    idx.close()


@pytest.mark.parametrize("shape", ["f32_256q_tiles", "f32_512q_tiles", "f32_cosine", "u8"])
def test_fast_paths_are_deterministic(shape):
    """Race screen (round 3): the same batch thirty times on the same index must return the same bits, and the
    oracle's answer.  Found with it: in bf_scan_bf16_kernel<.., QG=1, NW=8> (256-query tiles, i.e. batches of 256..1023
    queries) the compiler had placed a register copy of an LDS fragment IN FRONT of the s_waitcnt that makes it valid;
    the last 32-row block of every row split was scored from bytes that had not arrived -- rows listed at random, a true
    neighbour lost in ~8 % of the batches at 70k rows x 600 queries, on one GPU and behind 2-shard handles alike.  The
    fragments now live in pinned registers (BF_FRAG_RD)."""
    if shape == "u8":
        X, Q, k, space = refio.s_sift_like(140003, 21), refio.s_sift_like(600, 22), 100, "l2sqr_sift"
    else:
        n, nq = (70001, 600) if shape != "f32_512q_tiles" else (100003, 1100)
        X, Q, k = refio.s_lowrank(n, 128, 21), refio.s_lowrank(nq, 128, 22), 10
        space = "cosinesimil" if shape == "f32_cosine" else "l2"
    idx = make_index(space, "seq_search", X)
    first = idx.knnQueryBatch(Q, k)
    assert idx.stats()["last_path"] in (1, 3), "the fast path did not run"
    for _ in range(30):
        r = idx.knnQueryBatch(Q, k)
        np.testing.assert_array_equal(r[0], first[0])
        np.testing.assert_array_equal(r[1].view(np.uint32), first[1].view(np.uint32))
    opos, odist, _ = orc.seq_search(space, X, Q[:48], k)
    if space == "l2sqr_sift":
        np.testing.assert_array_equal(first[0][:48], opos)
        np.testing.assert_array_equal(first[1][:48], odist)
    else:
        assert (first[0][:48] == opos).mean() >= 0.999 and close_rel(first[1][:48], odist)
    idx.close()


def _golden3():
    import hashlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v3.npz"))

    def check(name, a):
        h = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
        assert np.array_equal(h, g[name]), f"{name}: regenerated input differs from the one the fixture was made from"
    return g, check


def test_full_size_c2_independent_queries_equal_the_reference():
    """BASELINE config 2 at full size with the bench's own queries (S-lowrank seeds 42 / 43: NOT base rows -- no
    distance-0 self match, ordinary gaps between ranks): the first 64 queries against the reference's own sequential
    scan (tests/golden/golden_v3.npz, generated by gen_golden_v3.py from oracle/_ref).  ids exact, distances 1e-5."""
    g, check = _golden3()
    X, Q = refio.s_lowrank(1_000_000, 128, 42), refio.s_lowrank(1024, 128, 43)
    check("c2_base_sha", X)
    check("c2_queries_sha", Q)
    idx = make_index("l2", "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, 10)
    st = idx.stats()
    assert st["last_path"] == 1 and st["fast_tiles_fallback"] == 0, st       # the path the bench line is measured on
    np.testing.assert_array_equal(ids[:64], g["c2_ids"])
    assert close_rel(ds[:64], g["c2_dists"])
    # the same 64 queries as their own small batch (adaptive f32 kernel): same rows, same floats
    ids_s, ds_s, _ = idx.knnQueryBatch(Q[:64], 10)
    np.testing.assert_array_equal(ids_s, ids[:64])
    np.testing.assert_array_equal(ds_s, ds[:64])
    idx.close()


def test_full_size_c4_independent_queries_equal_the_reference():
    """BASELINE config 4 at full size (1M x 128 u8, k = 100, batch 4096; S-sift-like seeds 44 / 45): the first 32
    queries against the reference's sequential scan -- integer distances identical, ids identical modulo tie groups."""
    from tests.gpuutil import ids_match_modulo_ties
    g, check = _golden3()
    U, UQ = refio.s_sift_like(1_000_000, 44), refio.s_sift_like(4096, 45)
    check("c4_base_sha", U)
    check("c4_queries_sha", UQ)
    idx = make_index("l2sqr_sift", "seq_search", U)
    ids, ds, cnt = idx.knnQueryBatch(UQ, 100)
    assert idx.stats()["last_path"] == 3
    np.testing.assert_array_equal(ds[:32], g["c4_dists"])
    assert ids_match_modulo_ties(ids[:32], ds[:32], g["c4_ids"], g["c4_dists"])
    idx.close()


@pytest.mark.parametrize("space,D", [("l2", 768), ("cosinesimil", 768), ("negdotprod", 200), ("l2", 300), ("angulardist", 512),
                                     ("l2", 1000)])
def test_f32_fast_path_long_rows(space, D, monkeypatch):
    """Rows longer than 128 (round 3): the one-product scan in chunks of 128 dimensions (bf_scan_bf16_kernel<.., KCH>),
    same sample / threshold / list re-rank / proof chain.  66k rows x D, 300 queries: the fast path runs (path 1), the
    answers are the oracle's, and they are the adaptive f32-MFMA path's bit for bit (both end in the reference formula
    on the original rows, lane for lane)."""
    n, nq, k = 66000, 300, 10
    X, Q = refio.s_lowrank(n, D, 301), refio.s_lowrank(nq, D, 302)
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    assert st["last_path"] == 1, st
    assert st["fast_tiles_fallback"] <= 1, st
    monkeypatch.setenv("NMSLIB_GPU_F32_FAST", "0")
    ids0, ds0, _ = idx.knnQueryBatch(Q, k)
    assert idx.stats()["last_path"] == 0
    monkeypatch.delenv("NMSLIB_GPU_F32_FAST")
    np.testing.assert_array_equal(ids, ids0)
    np.testing.assert_array_equal(ds.view(np.uint32), ds0.view(np.uint32))
    sel = np.r_[0:12, nq - 12:nq]
    opos, odist, _ = orc.seq_search(space, X, Q[sel], k + 22)
    assert refio.recall_nmslib(ids[sel], opos, odist, k) >= 0.999
    assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-6)
    for _ in range(5):                                  # race screen at this shape too
        r = idx.knnQueryBatch(Q, k)
        np.testing.assert_array_equal(r[0], ids)
    idx.close()


@pytest.mark.parametrize("space", ["cosinesimil", "angulardist"])
@pytest.mark.parametrize("kind", ["siftlike128", "offset32", "offset200"])
def test_centred_cosine_fast_path(space, kind, monkeypatch):
    """Cosine / angular on rows with a large common component (non-negative features, offsets): the selection runs on
    centred rows, and since round 3 on the bf16 fast path too -- the score -(1 - cos)|q| as an inner product of rows and
    queries with three more columns (row_aug_cosc_kernel).  70k rows, 300 queries: path 1, the oracle's answers, and the
    adaptive centred path's (NMSLIB_GPU_COSC_FAST=0) bit for bit."""
    n, nq, k = 70000, 300, 10
    if kind == "siftlike128":
        X, Q = refio.s_sift_like(n, 401).astype(np.float32), refio.s_sift_like(nq, 402).astype(np.float32)
    else:
        D = 32 if kind == "offset32" else 200
        X = (refio.s_lowrank(n, D, 403) + np.float32(1.5)).astype(np.float32)
        Q = (refio.s_lowrank(nq, D, 404) + np.float32(1.5)).astype(np.float32)
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    print(kind, space, st)
    assert st["last_path"] == 1 and st["fast_tiles"] > 0, st
    assert st["fast_tiles_fallback"] == 0, st
    monkeypatch.setenv("NMSLIB_GPU_COSC_FAST", "0")
    ids0, ds0, _ = idx.knnQueryBatch(Q, k)
    assert idx.stats()["last_path"] == 0
    monkeypatch.delenv("NMSLIB_GPU_COSC_FAST")
    np.testing.assert_array_equal(ids, ids0)
    np.testing.assert_array_equal(ds.view(np.uint32), ds0.view(np.uint32))
    sel = np.r_[0:12, nq - 12:nq]
    opos, odist, _ = orc.seq_search(space, X, Q[sel], k + 22)
    assert refio.recall_nmslib(ids[sel], opos, odist, k) >= 0.999
    assert close_rel(ds[sel], odist[:, :k], rtol=1e-5, atol=1e-6)
    Q[7] = 0.0                                           # a zero-norm query: its tile goes to the adaptive kernel
    ids1, ds1, _ = idx.knnQueryBatch(Q, k)
    assert idx.stats()["fast_tiles_fallback"] >= 1
    keep = np.arange(nq) != 7
    np.testing.assert_array_equal(ids1[keep], ids[keep])
    monkeypatch.setenv("NMSLIB_GPU_COSC_FAST", "0")
    ids2, ds2, _ = idx.knnQueryBatch(Q, k)
    monkeypatch.delenv("NMSLIB_GPU_COSC_FAST")
    np.testing.assert_array_equal(ids1, ids2)
    np.testing.assert_array_equal(ds1.view(np.uint32), ds2.view(np.uint32))
    idx.close()


@pytest.mark.parametrize("space", ["l2", "negdotprod", "cosinesimil"])
@pytest.mark.parametrize("kind", ["tiny", "huge", "outlier_row", "outlier_query", "wide_range"])
def test_f32_fast_path_value_ranges(space, kind, monkeypatch):
    """The one-product scan runs on fp16(scale * x) (round 3): the scale comes from the rows' largest |element|, the error
    bound from MEASURED residuals, and a query beyond fp16's range gets an infinite bound.  Whatever the magnitudes -- rows
    of 1e-12 or 1e12, one huge element among small ones, a query a million times larger than the rows, columns spread over
    twelve decades -- the answers are the adaptive f32 path's bit for bit (both end in the reference formula)."""
    n, D, nq, k = 70000, 48, 300, 10
    X, Q = refio.s_lowrank(n, D, 501), refio.s_lowrank(nq, D, 502)
    if kind == "tiny":
        X, Q = X * np.float32(1e-12), Q * np.float32(1e-12)
    elif kind == "huge":
        X, Q = X * np.float32(1e12), Q * np.float32(1e12)
    elif kind == "outlier_row":
        X = X.copy()
        X[12345, 7] = np.float32(3e6)
    elif kind == "outlier_query":
        Q = Q.copy()
        Q[5] *= np.float32(1e6)
    else:
        w = np.float32(10.0) ** np.linspace(-6, 6, D).astype(np.float32)
        X, Q = X * w, Q * w
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    st = idx.stats()
    print(kind, space, st)
    assert st["last_path"] == 1, st
    monkeypatch.setenv("NMSLIB_GPU_F32_FAST", "0")
    ids0, ds0, _ = idx.knnQueryBatch(Q, k)
    assert idx.stats()["last_path"] == 0
    monkeypatch.delenv("NMSLIB_GPU_F32_FAST")
    np.testing.assert_array_equal(ids, ids0)
    np.testing.assert_array_equal(ds.view(np.uint32), ds0.view(np.uint32))
    sel = np.r_[0:10, nq - 6:nq]
    opos, odist, _ = orc.seq_search(space, X, Q[sel], k + 22)
    assert refio.recall_nmslib(ids[sel], opos, odist, k) >= 0.999
    idx.close()
