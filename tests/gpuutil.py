"""Helpers for the -m gpu tests: every query goes through the C ABI of libnmslib_c.so."""
import numpy as np

import nmslib_zig_amd as nz

FLOAT_SPACES = ("l2", "l1", "linf", "cosinesimil", "angulardist", "negdotprod")


def make_index(space, method, base, ids=None, **index_params):
    u8 = space == "l2sqr_sift"
    idx = nz.Index(space, method, data_type="DenseUInt8Vector" if u8 else "DenseVector",
                   dist_type="Int" if u8 else "Float")
    if u8:
        idx.addUInt8Batch(base, ids)
    else:
        idx.addDenseBatch(base, ids)
    idx.buildIndex(**index_params)
    return idx


def close_rel(a, b, rtol=1e-5, atol=1e-6):
    """north_star's float bar: distances within 1e-5 relative of the reference formula."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= rtol * np.abs(b) + atol))


def ids_match_modulo_ties(got_ids, got_d, want_ids, want_d):
    """ids identical except inside equal-distance groups."""
    for q in range(got_ids.shape[0]):
        for dv in np.unique(want_d[q]):
            m = want_d[q] == dv
            if set(got_ids[q][m].tolist()) != set(want_ids[q][m].tolist()):
                if m.sum() == 1 or not np.isclose(got_d[q][m], dv).all():
                    return False
                # a tie group cut by rank k may legitimately hold other members
                if not (want_d[q][-1] == dv):
                    return False
    return True
