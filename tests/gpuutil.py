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


ULP1 = 2.0 ** -23   # spacing of float32 at 1.0


def sim_close(sim_got, sim_want, ulps=4):
    """Similarities (a dot product of unit vectors, |s| <= 1) within `ulps` float32 steps of each other: the quantity
    the reference computes accurately; 1 - s and acos(s) amplify its last bits when s is near 1."""
    a, b = np.asarray(sim_got, np.float64), np.asarray(sim_want, np.float64)
    return bool(np.all(np.abs(a - b) <= ulps * ULP1))


def ids_match_modulo_near_ties(got_ids, want_ids, want_key, eps, got_key=None):
    """ids identical at every rank whose reference key is separated from both neighbours by more than eps (callable of
    the key -> absolute resolution); inside a run of keys closer than that the SET must agree, except for a run
    that reaches rank k (it may continue past the cut).  Returns the list of offending (query, rank)."""
    bad = []
    for q in range(want_ids.shape[0]):
        key = np.asarray(want_key[q], np.float64)
        k = len(key)
        start = 0
        for i in range(1, k + 1):
            if i < k and abs(key[i] - key[i - 1]) <= eps(max(abs(key[i]), abs(key[i - 1]))):
                continue
            run = slice(start, i)
            if i - start == 1:
                if got_ids[q][start] != want_ids[q][start]:
                    # (the last rank may be a near-tie with the reference's unseen rank k+1: same key, other row)
                    last_tie = (i == k and got_key is not None and
                                abs(float(got_key[q][start]) - key[start]) <= eps(abs(key[start])))
                    if not last_tie:
                        bad.append((q, start))
            elif i < k and set(got_ids[q][run].tolist()) != set(want_ids[q][run].tolist()):
                bad.append((q, start))
            start = i
    return bad
