"""-m gpu: row shards behind ONE index handle (index parameter gpu_shards, SURVEY.md 8e / VERDICT r01 #7).
On a one-GPU box the shards share the device; the code path (per-shard engines, streams, peer copies, merge by
(distance, global position), id map) is the one an 8-GPU node runs.  The exact scan must reproduce the unsharded
index bit for bit."""
import numpy as np
import pytest

import nmslib_zig_amd as nz
from tests import refio
from tests.gpuutil import make_index

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("space,shards", [("l2", 2), ("l2", 3), ("cosinesimil", 2), ("l2sqr_sift", 2), ("l1", 2)])
def test_sharded_scan_equals_unsharded_bit_for_bit(space, shards):
    n, nq = 30011, 97
    if space == "l2sqr_sift":
        rng = np.random.default_rng(3)
        X = (rng.integers(0, 3, (n, 128)) * 60).astype(np.uint8)            # heavy ties: positions decide
        Q = (rng.integers(0, 3, (nq, 128)) * 60).astype(np.uint8)
        k = 50
    else:
        X, Q = refio.s_lowrank(n, 96, 5), refio.s_lowrank(nq, 96, 6)
        X[2000:2040] = X[17]                                                # planted duplicates across a shard border
        X[n // shards - 3:n // shards + 3] = X[17]
        k = 10
    ext = (np.random.default_rng(1).permutation(n).astype(np.int32) * 3 + 1)   # external ids unrelated to positions
    one = make_index(space, "seq_search", X, ext, gpu_shards=1)
    many = make_index(space, "seq_search", X, ext, gpu_shards=shards)
    assert many.stats()["shards"] == shards and one.stats()["shards"] == 1
    a = one.knnQueryBatch(Q, k)
    b = many.knnQueryBatch(Q, k)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    # single query entry + range query
    i1, d1 = many.knnQuery(Q[5], k)
    np.testing.assert_array_equal(i1, a[0][5])
    np.testing.assert_array_equal(d1, a[1][5])
    r = float(a[1][3, k - 1])
    ra, rb = one.rangeQueryFill(Q[3], r, 64), many.rangeQueryFill(Q[3], r, 64)
    np.testing.assert_array_equal(ra[0], rb[0])
    np.testing.assert_array_equal(ra[1], rb[1])
    one.close()
    many.close()


def test_sharded_more_shards_than_useful_and_k_beyond_a_shard():
    X, Q = refio.s_gauss(50, 16, 1), refio.s_gauss(7, 16, 2)
    one = make_index("l2", "seq_search", X, gpu_shards=1)
    many = make_index("l2", "seq_search", X, gpu_shards=4)                   # 12-13 rows per shard, k = 20
    a, b = one.knnQueryBatch(Q, 20), many.knnQueryBatch(Q, 20)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    huge = make_index("l2", "seq_search", X[:3], gpu_shards=8)               # clamped to the number of rows
    ids, ds, cnt = huge.knnQueryBatch(Q, 5)
    assert (cnt == 3).all() and (ids[:, 3:] == -1).all()
    for i in (one, many, huge):
        i.close()


def test_sharded_hnsw_recall_and_save_is_refused(tmp_path):
    n, nq, k = 40000, 300, 10
    X, Q = refio.s_lowrank(n, 64, 7), refio.s_lowrank(nq, 64, 8)
    bf = make_index("l2", "seq_search", X)
    gi, gd, _ = bf.knnQueryBatch(Q, k + 22)
    bf.close()
    rec = {}
    for sh in (1, 2):
        idx = make_index("l2", "hnsw", X, M=16, efConstruction=100, gpu_shards=sh)
        idx.setQueryTimeParams(efSearch=64)
        ids, ds, cnt = idx.knnQueryBatch(Q, k)
        rec[sh] = refio.recall_nmslib(ids, gi, gd ** 2, k)
        assert (cnt == k).all() and np.all(np.diff(ds, axis=1) >= 0)
        if sh == 2:
            with pytest.raises(nz.NmslibError):
                idx.save(str(tmp_path / "sharded.idx"), save_data=False)
        idx.close()
    assert rec[2] >= rec[1] - 0.002, rec        # each shard returns its local top-k: never worse than one graph


@pytest.mark.parametrize("space", ["l2", "l2sqr_sift"])
def test_sharded_fast_paths_equal_unsharded(space):
    """Shards big enough (>= 64k rows each, >= 512 queries) for the large-batch fast paths: per-shard sample thresholds,
    scans and list re-ranks, then the merge -- still the unsharded result bit for bit."""
    n, nq = 140003, 600
    if space == "l2sqr_sift":
        X, Q, k = refio.s_sift_like(n, 21), refio.s_sift_like(nq, 22), 100
    else:
        X, Q, k = refio.s_lowrank(n, 128, 21), refio.s_lowrank(nq, 128, 22), 10
        X[69990:70020] = X[3]                       # duplicates across the shard border
        Q[0] = X[3]
    one = make_index(space, "seq_search", X, gpu_shards=1)
    two = make_index(space, "seq_search", X, gpu_shards=2)
    a, b = one.knnQueryBatch(Q, k), two.knnQueryBatch(Q, k)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    small = one.knnQueryBatch(Q[:40], k)            # adaptive path on the same index: same rows
    np.testing.assert_array_equal(small[0], a[0][:40])
    np.testing.assert_array_equal(small[1], a[1][:40])
    one.close()
    two.close()


def test_sharded_k_beyond_the_lds_merge():
    """nshards * k keys that do not fit the LDS sort of merge_topk: the HBM rank-by-bisection merge takes
    over; brute force k = 3000 on 3 shards and HNSW algoType=old with ef = k = 2500 on 2 shards, against the
    unsharded index (exact scan: bit for bit; HNSW: each shard is its own graph, so the exact scan is the yardstick)."""
    n, nq = 20011, 9
    X, Q = refio.s_lowrank(n, 48, 31), refio.s_lowrank(nq, 48, 32)
    X[500:520] = X[7]                                  # ties that straddle the shard border
    X[n // 3 - 2:n // 3 + 2] = X[7]
    ext = np.random.default_rng(2).permutation(n).astype(np.int32) + 5
    one = make_index("l2", "seq_search", X, ext, gpu_shards=1)
    many = make_index("l2", "seq_search", X, ext, gpu_shards=3)
    a, b = one.knnQueryBatch(Q, 3000), many.knnQueryBatch(Q, 3000)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    many.close()
    h = make_index("l2", "hnsw", X, ext, M=12, efConstruction=60, gpu_shards=2)
    h.setQueryTimeParams(efSearch=2500, algoType="old")
    ids, ds, cnt = h.knnQueryBatch(Q, 2500)
    assert (cnt == 2500).all() and np.all(np.diff(ds, axis=1) >= 0)
    gi, gd, _ = one.knnQueryBatch(Q, 2500)
    rec = np.mean([len(set(x) & set(y)) / 2500 for x, y in zip(ids.tolist(), gi.tolist())])
    assert rec >= 0.98, rec
    h.close()
    one.close()


def test_hnsw_is_not_auto_sharded(monkeypatch):
    """without gpu_shards / NMSLIB_GPU_SHARDS an HNSW index is ONE graph whatever the number of visible
    GPUs (ids, recall and saveIndex must not depend on the machine); the exact scan may shard by itself."""
    for v in ("NMSLIB_GPU_SHARDS", "NMSLIB_GPU_DEVICE", "LOCAL_RANK"):
        monkeypatch.delenv(v, raising=False)
    X = refio.s_lowrank(3000, 32, 3)
    idx = make_index("l2", "hnsw", X, M=8, efConstruction=40)
    assert idx.stats()["shards"] == 1
    idx.close()


def test_shards_on_distinct_devices_when_the_node_has_them():
    """The peer-to-peer leg of Engine::knn_sharded (hipMemcpyPeerAsync of queries and per-shard lists, events across
    devices) only runs with >= 2 visible GPUs: skipped on a one-GPU box, exercised the first time hardware allows."""
    ndev = nz.lib().nmslib_gpu_device_count()
    if ndev < 2:
        pytest.skip("one visible GPU: all shards share it (covered by the tests above)")
    shards = min(ndev, 4)
    n, nq, k = 60000, 300, 10
    X, Q = refio.s_lowrank(n, 64, 41), refio.s_lowrank(nq, 64, 42)
    one = make_index("l2", "seq_search", X, gpu_shards=1)
    many = make_index("l2", "seq_search", X, gpu_shards=shards)
    assert many.stats()["shards"] == shards
    a, b = one.knnQueryBatch(Q, k), many.knnQueryBatch(Q, k)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    U, UQ = refio.s_sift_like(n, 43), refio.s_sift_like(nq, 44)
    one8 = make_index("l2sqr_sift", "seq_search", U, gpu_shards=1)
    many8 = make_index("l2sqr_sift", "seq_search", U, gpu_shards=shards)
    a, b = one8.knnQueryBatch(UQ, 50), many8.knnQueryBatch(UQ, 50)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    h = make_index("l2", "hnsw", X, M=16, efConstruction=100, gpu_shards=shards)
    h.setQueryTimeParams(efSearch=64)
    ids, ds, cnt = h.knnQueryBatch(Q, k)
    gi = one.knnQueryBatch(Q, k)[0]
    assert np.mean([len(set(x) & set(y)) / k for x, y in zip(ids.tolist(), gi.tolist())]) >= 0.95
    for i in (one, many, one8, many8, h):
        i.close()
