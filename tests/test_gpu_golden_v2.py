"""-m gpu: the HIP path against golden_v2 (reference outputs for 768-D rows, wide level-0 lists, range queries)."""
import numpy as np
import pytest

from tests.gpuutil import ULP1, close_rel, ids_match_modulo_near_ties, make_index, sim_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("space", ["l2", "cosinesimil"])
def test_seq_search_768(golden2, space):
    base, qs = golden2["d768"]()
    idx = make_index(space, "seq_search", base)
    ids, ds, cnt = idx.knnQueryBatch(qs, 10)
    np.testing.assert_array_equal(ids, golden2[f"seq_{space}_D768_ids"])
    assert close_rel(ds, golden2[f"seq_{space}_D768_dists"])
    idx.close()


@pytest.mark.parametrize("case,space,M,efc,efs", [("hnsw768", "cosinesimil", 16, 100, (10, 128, 200)),
                                                  ("wide", "l2", 32, 120, (40, 150))])
def test_hnsw_same_graph_as_reference(golden2, case, space, M, efc, efs):
    """indexThreadQty=1 -> the reference's graph (test_oracle_golden_v2.py proves it); the search kernel must
    then return the reference's SearchV1Merge results."""
    base, qs = golden2[case]()
    idx = make_index(space, "hnsw", base, M=M, efConstruction=efc, indexThreadQty=1)
    for ef in efs:
        idx.setQueryTimeParams(efSearch=ef)
        ids, ds, cnt = idx.knnQueryBatch(qs, 10)
        want_i, want_d = golden2[f"{case}_ef{ef}_ids"], golden2[f"{case}_ef{ef}_dists"]
        if space == "cosinesimil":
            # d = 1 - s: the similarity s is what both sides compute to a few ulp; 1e-5 relative on d itself wherever
            # the subtraction does not cancel (d >= 0.05 leaves 4 ulp of s below 1e-5 d)
            assert sim_close(1.0 - np.sort(ds, axis=1), 1.0 - np.sort(want_d, axis=1), ulps=4)
            big = want_d >= 0.05
            assert close_rel(ds[big], want_d[big], rtol=1e-5, atol=0.0)
            bad = ids_match_modulo_near_ties(ids, want_i, want_d, lambda v: 4 * ULP1, ds)
        else:
            assert close_rel(ds, want_d, rtol=1e-5, atol=1e-6)
            bad = ids_match_modulo_near_ties(ids, want_i, want_d, lambda v: 1e-5 * v, ds)
        assert not bad, bad[:5]                                # ids exact outside runs the reference cannot resolve
    idx.close()


@pytest.mark.parametrize("space,D", [("l2", 128), ("cosinesimil", 100), ("l1", 21)])
def test_range_query_equals_reference_shim(golden, golden2, space, D):
    base, qs = golden[f"f32_D{D}_base"], golden[f"f32_D{D}_queries"]
    idx = make_index(space, "seq_search", base, golden2["range_ext_ids"])
    for qi in (0, 3):
        radius = float(golden2[f"range_{space}_q{qi}_radius"][0])
        for cap in (128, 7):
            ids, ds = idx.rangeQueryFill(qs[qi], radius, cap)
            np.testing.assert_array_equal(ids, golden2[f"range_{space}_q{qi}_cap{cap}_ids"])
            assert close_rel(ds, golden2[f"range_{space}_q{qi}_cap{cap}_dists"])
    idx.close()
