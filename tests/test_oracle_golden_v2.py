"""CPU: the oracle and the library's host logic against golden_v2 (outputs of the real reference for 768-D
rows, wide level-0 lists and range queries; tests/golden/gen_golden_v2.py)."""
import numpy as np
import pytest

import nmslib_zig_amd as nz
from tests import orc, refio

RTOL = 1e-5


def close(a, b, atol=1e-6):
    return bool(np.all(np.abs(np.asarray(a, np.float64) - b) <= RTOL * np.abs(b) + atol))


@pytest.mark.parametrize("space", ["l2", "cosinesimil"])
def test_oracle_seq_search_768(golden2, space):
    base, qs = golden2["d768"]()
    pos, dist, cnt = orc.seq_search(space, base, qs, 10)
    np.testing.assert_array_equal(pos, golden2[f"seq_{space}_D768_ids"])
    assert close(dist, golden2[f"seq_{space}_D768_dists"])


@pytest.mark.parametrize("case,space,M,efc,efs", [("hnsw768", "cosinesimil", 16, 100, (10, 128, 200)),
                                                  ("wide", "l2", 32, 120, (40, 150))])
def test_oracle_build_and_search_equal_reference(golden2, case, space, M, efc, efs):
    base, qs = golden2[case]()
    g = orc.HnswGraph.build(space, base, M, efc)
    mx, ep, maxM, maxM0 = (int(v) for v in golden2[f"{case}_meta"])
    assert (g.maxlevel, g.enterpoint, g.maxM, g.maxM0) == (mx, ep, maxM, maxM0)
    np.testing.assert_array_equal(g.levels(), golden2[f"{case}_levels"])
    np.testing.assert_array_equal(g.links0(), golden2[f"{case}_links0"])
    np.testing.assert_array_equal(g.flat_upper()[1], golden2[f"{case}_up_links"])
    for ef in efs:
        pos, dist, cnt, ndc, hops = g.search(qs, 10, ef)
        np.testing.assert_array_equal(pos, golden2[f"{case}_ef{ef}_ids"])
        assert close(dist, golden2[f"{case}_ef{ef}_dists"])
        # (the reference's optimized-index search does not count distance computations: no ndc to pin here)


@pytest.mark.parametrize("case,space,M,efc", [("hnsw768", "cosinesimil", 16, 100), ("wide", "l2", 32, 120)])
def test_library_host_builder_equals_reference_graph(golden2, tmp_path, case, space, M, efc):
    base, _ = golden2[case]()
    idx = nz.Index(space, "hnsw")
    idx.addDenseBatch(base)
    idx.buildIndex(M=M, efConstruction=efc, indexThreadQty=1, gpu_defer=1)
    path = str(tmp_path / "idx")
    idx.save(path, False)
    P = refio.parse_optimized_index(path)
    mx, ep, maxM, maxM0 = (int(v) for v in golden2[f"{case}_meta"])
    assert (P["maxlevel"], P["enterpoint"], P["maxM"], P["maxM0"]) == (mx, ep, maxM, maxM0)
    np.testing.assert_array_equal(P["levels"], golden2[f"{case}_levels"])
    np.testing.assert_array_equal(P["links0"], golden2[f"{case}_links0"])
    np.testing.assert_array_equal(P["up_links"], golden2[f"{case}_up_links"])
    idx.close()


@pytest.mark.parametrize("space,D", [("l2", 128), ("cosinesimil", 100), ("l1", 21)])
def test_oracle_range_scan_equals_reference_shim(golden, golden2, space, D):
    """RangeQuery through the reference's C shim: matches in insertion order, capacity cut keeps the first."""
    base, qs = golden[f"f32_D{D}_base"], golden[f"f32_D{D}_queries"]
    ext = golden2["range_ext_ids"]
    for qi in (0, 3):
        radius = float(golden2[f"range_{space}_q{qi}_radius"][0])
        d = np.array([orc.space_distance(space, qs[qi], x) for x in base], np.float32)
        m = np.nonzero(d <= np.float32(radius))[0]
        for cap in (128, 7):
            want_i, want_d = golden2[f"range_{space}_q{qi}_cap{cap}_ids"], golden2[f"range_{space}_q{qi}_cap{cap}_dists"]
            np.testing.assert_array_equal(ext[m[:cap]], want_i)
            assert close(d[m[:cap]], want_d)
