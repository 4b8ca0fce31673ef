"""-m gpu: the multi-wave HNSW search (one workgroup per query: a control wave with the sorted array in registers
plus four gather waves; hnsw_mw_kernels.hip) against the one-wave kernel and the oracle.  Both kernels restate
Hnsw::SearchV1Merge (hnsw_distfunc_opt.cc:152-283) decision for decision, so ids, distances and the work counters
must be the same BITS whichever kernel served the batch."""
import os

import numpy as np
import pytest

import nmslib_zig_amd as nz
from tests import orc, refio
from tests.gpuutil import close_rel, make_index

pytestmark = pytest.mark.gpu


def _search(idx, Q, k, mode):
    old = os.environ.get("NMSLIB_HNSW_MW")
    os.environ["NMSLIB_HNSW_MW"] = mode
    try:
        ids, ds, cnt = idx.knnQueryBatch(Q, k)
        ctr = [x.copy() for x in idx.read_counters(len(Q))]
        redone = idx.stats()["hnsw_redone"]
    finally:
        if old is None:
            del os.environ["NMSLIB_HNSW_MW"]
        else:
            os.environ["NMSLIB_HNSW_MW"] = old
    return ids, ds, cnt, ctr, redone


@pytest.mark.parametrize("space,D", [("l2", 128), ("cosinesimil", 100), ("negdotprod", 48), ("l1", 21), ("linf", 21),
                                     ("angulardist", 36), ("l2", 200)])
def test_multiwave_equals_onewave_bit_for_bit(space, D):
    n, nq = 12000, 192
    X, Q = refio.s_lowrank(n, D, 71), refio.s_lowrank(nq, D, 72)
    idx = make_index(space, "hnsw", X, M=12, efConstruction=60, indexThreadQty=1)
    for ef, k in ((8, 10), (50, 10), (128, 10), (200, 100), (256, 7)):
        idx.setQueryTimeParams(efSearch=ef)
        a = _search(idx, Q, k, "0")
        b = _search(idx, Q, k, "2")
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        np.testing.assert_array_equal(a[2], b[2])
        for x, y in zip(a[3], b[3]):
            np.testing.assert_array_equal(x, y)
        assert b[4] == 0, "the multi-wave kernel sent queries to the bitset kernel"
    idx.close()


def test_multiwave_ties_follow_the_reference_placement():
    """Rows on a coarse integer grid, every row stored three times: equal keys everywhere, so the insertions go
    through the replayed exponential probe of SortArrBI::push_or_replace_non_empty_exp (sort_arr_bi.h:172-186).
    Oracle on the same graph: same ids, same distances, same counters."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 3, size=(3000, 24)).astype(np.float32)
    X = np.concatenate([base, base, base])[rng.permutation(9000)]
    Q = rng.integers(0, 3, size=(128, 24)).astype(np.float32)
    idx = make_index("l2", "hnsw", X, M=8, efConstruction=40, indexThreadQty=1)
    g = orc.HnswGraph.build("l2", X, 8, 40)
    for ef, k in ((10, 10), (64, 10), (128, 50)):
        idx.setQueryTimeParams(efSearch=ef)
        opos, odist, ocnt, ondc, ohops = g.search(Q, k, ef)
        for mode in ("2", "0"):
            r = _search(idx, Q, k, mode)
            np.testing.assert_array_equal(r[0], opos)
            np.testing.assert_array_equal(r[1], odist)          # small integers: exact in f32
            np.testing.assert_array_equal(r[3][0].astype(np.int64), ondc)
            np.testing.assert_array_equal(r[3][1].astype(np.int64), ohops)
            assert r[4] == 0
    idx.close()


def test_multiwave_oracle_parity_1024_queries_counters():
    """The C3 shape at 50k rows: batch 1024, ef 128, k 10 through the default dispatch (multi-wave)."""
    n, D, nq = 50000, 128, 1024
    X, Q = refio.s_lowrank(n, D, 81), refio.s_lowrank(nq, D, 82)
    idx = make_index("l2", "hnsw", X, M=16, efConstruction=100, indexThreadQty=1)
    g = orc.HnswGraph.build("l2", X, 16, 100)
    idx.setQueryTimeParams(efSearch=128)
    ids, ds, cnt = idx.knnQueryBatch(Q, 10)
    ndc, hops, hops_up = (x.astype(np.int64) for x in idx.read_counters(nq))
    assert idx.stats()["hnsw_redone"] == 0
    opos, odist, ocnt, ondc, ohops = g.search(Q, 10, 128)
    assert (ids == opos).mean() >= 0.999 and close_rel(ds, odist)
    assert np.mean(ndc == ondc) >= 0.98 and abs(ndc.mean() / ondc.mean() - 1) < 0.01
    assert np.mean(hops == ohops) >= 0.98
    idx.close()


def test_multiwave_table_overflow_goes_to_the_bitset_kernel():
    n, D, nq = 30000, 32, 40
    X, Q = refio.s_gauss(n, D, 61), refio.s_gauss(nq, D, 62)
    idx = make_index("l2", "hnsw", X, M=6, efConstruction=40, indexThreadQty=1)
    g = orc.HnswGraph.build("l2", X, 6, 40)
    idx.setQueryTimeParams(efSearch=250, algoType="v1merge")
    a = _search(idx, Q, 10, "0")
    b = _search(idx, Q, 10, "2")
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    opos, odist, _, _, _ = g.search(Q, 10, 250)
    assert (b[0] == opos).mean() >= 0.999 and close_rel(b[1], odist)
    idx.close()
