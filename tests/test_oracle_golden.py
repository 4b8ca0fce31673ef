"""The CPU oracle (oracle/knn_oracle.c) against the committed golden vectors, which are outputs of
the real reference (tests/golden/gen_golden.py).  CPU only.

Bars: integer paths bit-exact; float ids exact and distances within 1e-5 relative (the oracle keeps
the reference's lane structure, so in practice they agree to the last bit except where the
reference calls libm through a different overload)."""
import numpy as np
import pytest

from tests import orc, refio

FLOAT_SPACES = ("l2", "l1", "linf", "cosinesimil", "angulardist", "negdotprod")
RTOL = 1e-5


def close(a, b, atol=1e-6):
    return np.all(np.abs(a - b) <= RTOL * np.abs(b) + atol)


@pytest.mark.parametrize("D", [128, 100, 21])
@pytest.mark.parametrize("space", FLOAT_SPACES)
def test_seq_search_float(golden, space, D):
    base, qs = golden[f"f32_D{D}_base"], golden[f"f32_D{D}_queries"]
    pos, dist, cnt = orc.seq_search(space, base, qs, 10)
    assert np.all(cnt == 10)
    np.testing.assert_array_equal(pos, golden[f"seq_{space}_D{D}_ids"])
    assert close(dist, golden[f"seq_{space}_D{D}_dists"])


@pytest.mark.parametrize("D", [128, 100, 21])
@pytest.mark.parametrize("space", FLOAT_SPACES)
def test_pairwise_distance(golden, space, D):
    base = golden[f"f32_D{D}_base"]
    pairs, want = golden[f"pair_{space}_D{D}_idx"], golden[f"pair_{space}_D{D}_dists"]
    got = np.array([orc.space_distance(space, base[a], base[b]) for a, b in pairs], np.float32)
    assert close(got, want)


def test_cosine_zero_norm_rule(golden):
    # distcomp_scalar.cc:154-160: a (nearly) zero vector has similarity 0 -> distance 1
    base = golden["f32_D128_base"]
    assert orc.space_distance("cosinesimil", base[7], base[1]) == 1.0
    assert orc.space_distance("cosinesimil", base[3], base[11]) <= 1e-6   # duplicate rows
    q = np.array([1, 0, 0, 0], np.float32)
    rows = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [-1, 0, 0, 0]], np.float32)
    got = [orc.space_distance("cosinesimil", r, q) for r in rows]
    assert got == [1.0, 0.0, 2.0]          # SURVEY.md 8a A4 [measured]


def test_seq_search_u8_bit_exact(golden):
    pos, dist, cnt = orc.seq_search("l2sqr_sift", golden["u8_base"], golden["u8_queries"], 100)
    np.testing.assert_array_equal(dist, golden["seq_l2sqr_sift_dists"])
    np.testing.assert_array_equal(pos, golden["seq_l2sqr_sift_ids"])   # (dist, position) tie order


def _graph(golden, space, D):
    mx, ep, maxM, maxM0 = (int(x) for x in golden[f"hnsw_{space}_meta"][:4])
    return orc.HnswGraph.from_arrays(space, golden[f"f32_D{D}_base"], maxM, maxM0, mx, ep,
                                     golden[f"hnsw_{space}_levels"], golden[f"hnsw_{space}_links0"],
                                     golden[f"hnsw_{space}_up_off"], golden[f"hnsw_{space}_up_links"])


HNSW_CASES = [("l2", 128), ("cosinesimil", 100), ("negdotprod", 21), ("l1", 21)]


@pytest.mark.parametrize("space,D", HNSW_CASES)
def test_hnsw_build_is_the_reference_graph(golden, space, D):
    """Single-threaded build with mt19937(0) levels reproduces the reference's adjacency exactly."""
    g = orc.HnswGraph.build(space, golden[f"f32_D{D}_base"], M=8, efConstruction=50)
    mx, ep = (int(x) for x in golden[f"hnsw_{space}_meta"][:2])
    assert (g.maxlevel, g.enterpoint) == (mx, ep)
    np.testing.assert_array_equal(g.levels(), golden[f"hnsw_{space}_levels"])
    np.testing.assert_array_equal(g.links0(), golden[f"hnsw_{space}_links0"])
    off, up = g.flat_upper()
    np.testing.assert_array_equal(off, golden[f"hnsw_{space}_up_off"])
    np.testing.assert_array_equal(up, golden[f"hnsw_{space}_up_links"])


@pytest.mark.parametrize("algo", ["v1merge", "old"])
@pytest.mark.parametrize("ef", [5, 20, 200])
@pytest.mark.parametrize("space,D", HNSW_CASES)
def test_hnsw_search_same_graph(golden, space, D, ef, algo):
    g = _graph(golden, space, D)
    pos, dist, cnt, ndc, hops = g.search(golden[f"f32_D{D}_queries"], 10, ef, optimized=True, algo=algo)
    np.testing.assert_array_equal(pos, golden[f"hnsw_{space}_ef{ef}_{algo}_ids"])
    assert close(dist, golden[f"hnsw_{space}_ef{ef}_{algo}_dists"])


def test_hnsw_generic_path_angular_and_ndc(golden):
    g = orc.HnswGraph.build("angulardist", golden["f32_D21_base"], M=8, efConstruction=50)
    pos, dist, cnt, ndc, hops = g.search(golden["f32_D21_queries"], 10, 20, optimized=False)
    np.testing.assert_array_equal(pos, golden["hnsw_angulardist_ids"])
    assert close(dist, golden["hnsw_angulardist_dists"])
    np.testing.assert_array_equal(ndc, golden["hnsw_angulardist_ndc"])   # Query::DistanceComputations


def test_hnsw_u8_generic_path_bit_exact(golden):
    g = orc.HnswGraph.build("l2sqr_sift", golden["u8_base"], M=8, efConstruction=50)
    pos, dist, cnt, ndc, hops = g.search(golden["u8_queries"], 100, 150, optimized=False)
    np.testing.assert_array_equal(dist, golden["hnsw_l2sqr_sift_dists"])
    np.testing.assert_array_equal(ndc, golden["hnsw_l2sqr_sift_ndc"])
    want = golden["hnsw_l2sqr_sift_ids"]
    # ids identical except inside equal-distance groups (the reference orders those by heap
    # address): compare as sets per distinct distance
    for q in range(pos.shape[0]):
        for dv in np.unique(dist[q]):
            m = dist[q] == dv
            if m.sum() == 1:
                assert pos[q][m][0] == want[q][m][0]


def test_level_stream_is_mt19937_seed0(golden):
    lv = orc.random_levels(300, M=8, seed=0)
    np.testing.assert_array_equal(lv, golden["hnsw_l2_levels"])


def test_oracle_matches_reference_full_size_fixture_subsample():
    """golden_v3 (the reference's scan of the 1M-row BASELINE sets, independent queries): the oracle reproduces it on a
    few queries -- the pin of the full-size GPU tests travels through the same formulas."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v3.npz"))
    X, Q = refio.s_lowrank(1_000_000, 128, 42), refio.s_lowrank(1024, 128, 43)
    pos, d, _ = orc.seq_search("l2", X, Q[:4], 10)
    np.testing.assert_array_equal(pos, g["c2_ids"][:4])
    np.testing.assert_array_equal(d, g["c2_dists"][:4])
    U, UQ = refio.s_sift_like(1_000_000, 44), refio.s_sift_like(4096, 45)
    pos, d, _ = orc.seq_search("l2sqr_sift", U, UQ[:2], 100)
    np.testing.assert_array_equal(d, g["c4_dists"][:2])
