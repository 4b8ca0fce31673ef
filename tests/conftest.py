import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden2():
    """tests/golden/golden_v2.npz (gen_golden_v2.py): 768-D, wide lists, range queries; the big inputs are
    regenerated from their seeds and checked against the stored SHA-256."""
    import hashlib
    from tests.golden import gen_golden_v2 as g2
    path = os.path.join(ROOT, "tests", "golden", "golden_v2.npz")
    with np.load(path) as z:
        d = {k: z[k] for k in z.files}

    def pinned(arrays, names):
        for a, name in zip(arrays, names):
            got = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
            assert np.array_equal(got, d[name]), f"regenerated input {name} differs from the one the fixture was made with"
        return arrays

    d["d768"] = lambda: pinned(g2.inputs_768(), ("d768_base_sha", "d768_queries_sha"))
    d["hnsw768"] = lambda: pinned(g2.inputs_hnsw768(), ("hnsw768_base_sha", "hnsw768_queries_sha"))
    d["wide"] = lambda: pinned(g2.inputs_wide(), ("wide_base_sha", "wide_queries_sha"))
    return d


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
