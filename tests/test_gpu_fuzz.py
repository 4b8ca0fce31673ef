"""-m gpu: seeded random shapes through the C ABI against the oracle -- ragged dimensions, tiny and odd row counts,
k above and below the row count, every space, both methods.  Small cases: the whole file runs in seconds."""
import numpy as np
import pytest

from tests import orc, refio
from tests.gpuutil import FLOAT_SPACES, close_rel, ids_match_modulo_ties, make_index

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    space = (FLOAT_SPACES + ("l2sqr_sift",))[seed % 7]
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 129, 500, 1000, 4097]))
    dim = 128 if space == "l2sqr_sift" else int(rng.choice([1, 2, 3, 7, 8, 9, 31, 33, 64, 100, 127, 128, 129, 200, 257]))
    nq = int(rng.choice([1, 2, 31, 33, 128, 129, 300]))
    k = int(rng.choice([1, 2, 9, 10, 17, 33, 60]))
    return space, n, dim, nq, k


@pytest.mark.parametrize("seed", range(28))
def test_bruteforce_random_shapes(seed):
    space, n, dim, nq, k = _case(seed)
    if space == "l2sqr_sift":
        X, Q = refio.s_sift_like(n, 10 + seed), refio.s_sift_like(nq, 20 + seed)
    else:
        X, Q = refio.s_gauss(n, dim, 10 + seed), refio.s_gauss(nq, dim, 20 + seed)
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    opos, odist, ocnt = orc.seq_search(space, X, Q, k)
    np.testing.assert_array_equal(cnt, ocnt)
    valid = opos >= 0
    if space == "l2sqr_sift":
        np.testing.assert_array_equal(ds[valid], odist[valid])
        assert ids_match_modulo_ties(np.where(valid, ids, -1), np.where(valid, ds, 0), np.where(valid, opos, -1),
                                     np.where(valid, odist, 0))
    else:
        rtol, atol = (1e-4, 1e-5) if space == "angulardist" else (1e-5, 1e-6)      # acos amplifies ulps near 0
        assert close_rel(ds[valid], odist[valid], rtol=rtol, atol=atol), (space, n, dim, nq, k)
        if dim > 3:
            assert (ids[valid] == opos[valid]).mean() >= 0.995, (space, n, dim, nq, k)  # near-equal distances may swap
        else:
            # 1-3 dimensions: distances collapse onto a few values (1-D angular: 0 or pi), any member of a tie group
            # is a correct answer -- check that every returned row really lies at its reported distance
            for q in range(min(nq, 8)):
                for j in range(int(cnt[q])):
                    d = orc.space_distance(space, Q[q], X[ids[q, j]])
                    assert abs(d - ds[q, j]) <= 1e-4 * abs(d) + 1e-4, (space, n, dim, q, j)
    assert (ids[~valid] == -1).all()
    idx.close()


@pytest.mark.parametrize("seed", range(14))
def test_hnsw_random_shapes_same_graph(seed):
    rng = np.random.default_rng(2000 + seed)
    space = ("l2", "cosinesimil", "negdotprod", "l1", "linf", "angulardist", "l2sqr_sift")[seed % 7]
    n = int(rng.choice([1, 2, 5, 64, 65, 300, 2000]))
    dim = 128 if space == "l2sqr_sift" else int(rng.choice([2, 7, 8, 33, 100, 128, 130]))
    M = int(rng.choice([2, 5, 16, 33, 40]))
    efc = int(rng.choice([10, 60, 150]))
    ef = int(rng.choice([1, 10, 77, 300]))
    k = int(rng.choice([1, 10, 25]))
    nq = int(rng.choice([1, 17, 100]))
    if space == "l2sqr_sift":
        X, Q = refio.s_sift_like(n, 30 + seed), refio.s_sift_like(nq, 40 + seed)
    else:
        X, Q = refio.s_gauss(n, dim, 30 + seed), refio.s_gauss(nq, dim, 40 + seed)
    idx = make_index(space, "hnsw", X, M=M, efConstruction=efc, indexThreadQty=1)
    g = orc.HnswGraph.build(space, X, M, efc)
    idx.setQueryTimeParams(efSearch=ef)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    optimized = space in ("l2", "cosinesimil", "negdotprod", "l1", "linf")       # hnsw.cc:369-412: flat index or generic path
    opos, odist, ocnt, _, _ = g.search(Q, k, ef, optimized=optimized)
    np.testing.assert_array_equal(cnt, ocnt)
    valid = opos >= 0
    if space == "l2sqr_sift":
        np.testing.assert_array_equal(ds[valid], odist[valid])
    else:
        rtol, atol = (1e-4, 1e-5) if space == "angulardist" else (1e-5, 1e-6)
        assert close_rel(ds[valid], odist[valid], rtol=rtol, atol=atol), (space, n, dim, M, ef, k)
    if dim > 3:
        assert (ids[valid] == opos[valid]).mean() >= 0.99, (space, n, dim, M, ef, k)
    idx.close()


@pytest.mark.parametrize("seed", range(12))
def test_fast_paths_random_shapes(seed):
    """The large-batch fast paths (sample-fixed thresholds + streaming scan; float rows through the split-bf16 MFMA, bytes
    through the int8 MFMA) on random shapes around their switch-over points: row counts just above 64k that are not
    multiples of the tile, batches that are not multiples of the query tile, k from 1 to 128, D from 8 to 128, data with
    and without a common offset; gaussian data (no structure: the hardest case for a fixed threshold)."""
    rng = np.random.default_rng(3000 + seed)
    space = ("l2", "l2sqr_sift", "negdotprod", "cosinesimil", "l2", "angulardist")[seed % 6]
    n = int(rng.choice([65536, 65537, 70001, 100003]))
    dim = 128 if space == "l2sqr_sift" else int(rng.choice([8, 24, 64, 100, 128]))
    nq = int(rng.choice([512, 513, 700, 1025]))
    k = int(rng.choice([1, 10, 37, 100, 128]))
    if space == "l2sqr_sift":
        X, Q = refio.s_sift_like(n, 50 + seed), refio.s_sift_like(nq, 60 + seed)
    else:
        off = np.float32(50.0 if (space == "l2" and seed % 2) else 0.0)
        X, Q = refio.s_gauss(n, dim, 50 + seed) + off, refio.s_gauss(nq, dim, 60 + seed) + off
    idx = make_index(space, "seq_search", X)
    ids, ds, cnt = idx.knnQueryBatch(Q, k)
    assert (cnt == k).all()
    sel = np.r_[0:6, nq - 6:nq]
    opos, odist, _ = orc.seq_search(space, X, Q[sel], k + 22)
    if space == "l2sqr_sift":
        np.testing.assert_array_equal(ds[sel], odist[:, :k])
        np.testing.assert_array_equal(ids[sel], opos[:, :k])
    else:
        rec = refio.recall_nmslib(ids[sel], opos, odist, k)
        assert rec >= 0.999, (space, n, dim, nq, k, rec)
        rtol, atol = (1e-4, 1e-5) if space == "angulardist" else (1e-5, 1e-6)
        assert close_rel(ds[sel], odist[:, :k], rtol=rtol, atol=atol * max(1.0, float(np.abs(X).max()))), (space, n, dim, nq, k)
    idx.close()
