"""HNSW construction options of round 3 against fixtures from the REAL reference (tests/golden/gen_golden_v4.py):
delaunay_type 0..3 x post 0..2 (hnsw.cc:251-330, hnsw.h:82-256) and M = 64 (lists longer than two words per lane).
The graph half needs no GPU (host builder, gpu_defer=1); the search half is -m gpu."""
import os

import numpy as np
import pytest

import nmslib_zig_amd as nz
from tests import refio
from tests.golden.gen_golden_v4 import COMBOS, inputs_m64, inputs_opts, sha

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v4.npz")


@pytest.fixture(scope="module")
def g4():
    return np.load(GOLDEN)


def build(base, **params):
    idx = nz.Index("l2", "hnsw")
    idx.addDenseBatch(base)
    idx.buildIndex(indexThreadQty=1, **params)
    return idx


def check_graph(idx, g4, tag, tmp_path):
    path = str(tmp_path / tag)
    idx.save(path, False)
    P = refio.parse_optimized_index(path)
    assert [P["maxlevel"], P["enterpoint"], P["maxM"], P["maxM0"]] == [int(v) for v in g4[f"{tag}_meta"]]
    for key in ("levels", "links0", "up_off", "up_links"):
        np.testing.assert_array_equal(P[key], g4[f"{tag}_{key}"], err_msg=f"{tag} {key}")


def test_inputs_are_the_fixtures_inputs(g4):
    base, qs = inputs_opts()
    np.testing.assert_array_equal(sha(base), g4["opts_base_sha"])
    np.testing.assert_array_equal(sha(qs), g4["opts_queries_sha"])
    base, qs = inputs_m64()
    np.testing.assert_array_equal(sha(base), g4["m64_base_sha"])
    np.testing.assert_array_equal(sha(qs), g4["m64_queries_sha"])


@pytest.mark.parametrize("dl,post", COMBOS)
def test_delaunay_and_post_graphs_equal_the_reference(g4, tmp_path, dl, post):
    """Selection heuristics 1 and 3 and the post-processing (second index in reverse order, union of the level-0 lists,
    post=2: ranked again; post=1: kept whole, maxM0 widened): the reference's graph at one thread, list order included."""
    base, _ = inputs_opts()
    idx = build(base, M=6, efConstruction=40, delaunay_type=dl, post=post, gpu_defer=1)
    check_graph(idx, g4, f"d{dl}p{post}", tmp_path)
    idx.close()


def test_m64_graph_equals_the_reference(g4, tmp_path):
    base, _ = inputs_m64()
    idx = build(base, M=64, efConstruction=150, gpu_defer=1)
    check_graph(idx, g4, "m64", tmp_path)
    assert g4["m64_links0"][:, 0].max() > 63
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dl,post", COMBOS)
def test_search_on_post_processed_graphs_equals_the_reference(g4, dl, post):
    from tests.gpuutil import close_rel
    base, qs = inputs_opts()
    idx = build(base, M=6, efConstruction=40, delaunay_type=dl, post=post)
    idx.setQueryTimeParams(efSearch=40)
    ids, ds, _ = idx.knnQueryBatch(qs, 10)
    np.testing.assert_array_equal(ids, g4[f"d{dl}p{post}_ids"])
    assert close_rel(ds, g4[f"d{dl}p{post}_dists"])       # (the reference's hnsw l2 is squared: distances as it reports them)
    idx.close()


@pytest.mark.gpu
def test_search_m64_equals_the_reference_both_algorithms(g4):
    from tests.gpuutil import close_rel
    base, qs = inputs_m64()
    idx = build(base, M=64, efConstruction=150)
    for algo, tag in (("v1merge", "m64"), ("old", "m64_old")):
        idx.setQueryTimeParams(efSearch=60, algoType=algo)
        ids, ds, _ = idx.knnQueryBatch(qs, 10)
        np.testing.assert_array_equal(ids, g4[f"{tag}_ids"])
        assert close_rel(ds, g4[f"{tag}_dists"])
    idx.close()
