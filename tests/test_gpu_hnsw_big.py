"""-m gpu: SearchV1Merge with a sorted array beyond the LDS kernels' 1024 items (hnsw_search_big_kernel): reachable
with an explicit algoType=v1merge and efSearch > 1024, or k > 1024 below efSearch 1000.  The reference sizes SortArrBI to
max(ef, k) without a cap (hnsw_distfunc_opt.cc:152-167); results, distances and work counters must equal the oracle's on
the same graph."""
import numpy as np
import pytest

from tests import orc, refio
from tests.gpuutil import close_rel, make_index

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("space,M", [("l2", 12), ("cosinesimil", 8), ("l2", 40)])
def test_v1merge_beyond_1024_items_equals_the_oracle(space, M):
    n, D, nq = 15000, 40, 24
    X, Q = refio.s_lowrank(n, D, 91), refio.s_lowrank(nq, D, 92)
    idx = make_index(space, "hnsw", X, M=M, efConstruction=60, indexThreadQty=1)
    g = orc.HnswGraph.build(space, X, M, 60)
    for ef, k in ((1500, 10), (1100, 1100), (50, 1300), (2100, 300)):
        idx.setQueryTimeParams(efSearch=ef, algoType="v1merge")
        ids, ds, cnt = idx.knnQueryBatch(Q, k)
        opos, odist, ocnt, ondc, ohops = g.search(Q, k, ef)
        np.testing.assert_array_equal(cnt, ocnt)
        assert (ids == opos).mean() >= 0.999
        assert close_rel(ds[opos >= 0], odist[opos >= 0])
        ndc, hops, hops_up = (x.astype(np.int64) for x in idx.read_counters(nq))
        assert np.mean(ndc == ondc) >= 0.95 and abs(ndc.mean() / ondc.mean() - 1) < 0.01
        assert np.mean(hops == ohops) >= 0.95
    idx.close()


def test_v1merge_big_with_ties_is_exact():
    rng = np.random.default_rng(6)
    base = rng.integers(0, 3, size=(2500, 20)).astype(np.float32)
    X = np.concatenate([base, base])[rng.permutation(5000)]
    Q = rng.integers(0, 3, size=(16, 20)).astype(np.float32)
    idx = make_index("l2", "hnsw", X, M=8, efConstruction=40, indexThreadQty=1)
    g = orc.HnswGraph.build("l2", X, 8, 40)
    idx.setQueryTimeParams(efSearch=1200, algoType="v1merge")
    ids, ds, cnt = idx.knnQueryBatch(Q, 50)
    opos, odist, ocnt, ondc, ohops = g.search(Q, 50, 1200)
    np.testing.assert_array_equal(ds, odist)
    np.testing.assert_array_equal(ids, opos)
    ndc, hops, _ = (x.astype(np.int64) for x in idx.read_counters(len(Q)))
    np.testing.assert_array_equal(ndc, ondc)
    np.testing.assert_array_equal(hops, ohops)
    idx.close()


@pytest.mark.parametrize("space,M,maxM,maxM0", [("l2", 64, None, None), ("cosinesimil", 70, None, None), ("l2", 20, 90, 200),
                                                 ("l2", 20, 70, 300)])
def test_any_M_same_graph_same_walk(space, M, maxM, maxM0, tmp_path):
    """M / maxM > 62 or maxM0 > 126 (hnsw.cc:189-208 takes any M): built on the host in the reference's order, searched by
    the kernels that walk adjacency lists in chunks of 64: the LDS SearchV1Merge kernel up to maxM0 = 254 (frontier arrays of
    256 entries), the HBM-array one beyond (the last case), SearchOld always.  Same graph as the oracle's build, and on it the
    same ids, distances and work counters for SearchV1Merge and SearchOld."""
    from tests.test_gpu_hnsw_build import graph_of
    n, D, nq, k = 4000, 24, 96, 10
    X, Q = refio.s_lowrank(n, D, 811), refio.s_lowrank(nq, D, 812)
    extra = {}
    if maxM is not None:
        extra = dict(maxM=maxM, maxM0=maxM0)
    idx = make_index(space, "hnsw", X, M=M, efConstruction=120, indexThreadQty=1, **extra)
    g = orc.HnswGraph.build(space, X, M, 120, maxM=maxM, maxM0=maxM0)
    got = graph_of(idx, tmp_path, "wide.idx")
    np.testing.assert_array_equal(got["links0"], g.links0())
    assert got["links0"][:, 0].max() > 63                     # (lists really are longer than one word per lane)
    for algo in ("v1merge", "old"):
        for ef in (40, 150):
            idx.setQueryTimeParams(efSearch=ef, algoType=algo)
            ids, ds, cnt = idx.knnQueryBatch(Q, k)
            opos, odist, ocnt, ondc, ohops = g.search(Q, k, ef, algo=algo)
            np.testing.assert_array_equal(ids, opos)
            assert close_rel(ds, odist)
            ndc, hops, hops_up = (x.astype(np.int64) for x in idx.read_counters(nq))
            if algo == "v1merge":
                np.testing.assert_array_equal(ndc, ondc)
                np.testing.assert_array_equal(hops, ohops)
    idx.close()
