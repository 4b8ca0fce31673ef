"""The CPU oracle against the real reference run live (oracle/_ref).  Skipped where the compiled
reference is absent.  Larger/randomised versions of the golden checks."""
import os
import tempfile

import numpy as np
import pytest

from tests import orc, refio

pytestmark = pytest.mark.skipif(not refio.HAVE_REF, reason="oracle/_ref not built (make -C oracle ref)")


@pytest.mark.parametrize("space,D", [("l2", 128), ("l2", 50), ("cosinesimil", 96), ("negdotprod", 64),
                                     ("l1", 40), ("linf", 40)])
def test_build_and_search_match_reference(space, D):
    X, Y = refio.s_lowrank(1500, D, 1), refio.s_lowrank(24, D, 2)
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "idx")
    rid, rd, _, _, _ = refio.run_ref_driver(space, "hnsw", X, Y, 10,
                                            "M=12,efConstruction=80,indexThreadQty=1", "efSearch=64",
                                            save=path)
    P = refio.parse_optimized_index(path)
    g = orc.HnswGraph.build(space, X, 12, 80)
    np.testing.assert_array_equal(g.links0(), P["links0"])
    off, up = g.flat_upper()
    np.testing.assert_array_equal(off, P["up_off"])
    np.testing.assert_array_equal(up, P["up_links"])
    pos, dist, _, _, _ = g.search(Y, 10, 64)
    np.testing.assert_array_equal(pos, rid)
    assert np.all(np.abs(dist - rd) <= 1e-5 * np.abs(rd) + 1e-6)


def test_u8_seq_and_hnsw_match_reference():
    X, Y = refio.s_sift_like(3000, 7), refio.s_sift_like(16, 8)
    rid, rd, _, _, _ = refio.run_ref_driver("l2sqr_sift", "seq_search", X, Y, 50)
    pos, dist, _ = orc.seq_search("l2sqr_sift", X, Y, 50)
    np.testing.assert_array_equal(dist, rd)
    np.testing.assert_array_equal(pos, rid)
    rid, rd, _, rndc, _ = refio.run_ref_driver("l2sqr_sift", "hnsw", X, Y, 50,
                                               "M=8,efConstruction=60,indexThreadQty=1", "efSearch=80")
    g = orc.HnswGraph.build("l2sqr_sift", X, 8, 60)
    pos, dist, _, ndc, _ = g.search(Y, 50, 80, optimized=False)
    np.testing.assert_array_equal(dist, rd)
    np.testing.assert_array_equal(ndc, rndc)
