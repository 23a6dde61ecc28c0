"""Shared comparison helpers for the parity tests."""
import numpy as np
import scipy.sparse as sp

RTOL = 1e-10   # BASELINE.json north_star: factors and solutions within 1e-10 relative


def canon(n, Gp, Gi, Gx):
    """Row-sorted copy of a CSC triple (the oracle's columns are in DFS order)."""
    G = sp.csc_matrix((np.asarray(Gx, dtype=np.float64), np.asarray(Gi), np.asarray(Gp)), shape=(n, n))
    G.sort_indices()
    return G.indptr.astype(np.int32), G.indices.astype(np.int32), G.data.copy()


def rel_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    scale = np.abs(want).max() if want.size else 1.0
    if scale == 0.0:
        scale = 1.0
    return float(np.abs(got - want).max() / scale) if want.size else 0.0


def assert_factor_equal(n, got, want, what):
    """Pattern bit-exact, values within RTOL (norm-wise) of the oracle."""
    gp, gi, gx = canon(n, *got)
    wp, wi, wx = canon(n, *want)
    assert np.array_equal(gp, wp), what + ": column pointers differ"
    assert np.array_equal(gi, wi), what + ": row indices differ"
    err = rel_err(gx, wx)
    assert err <= RTOL, "%s: relative error %.3e > %.1e" % (what, err, RTOL)
    return err


def csc_to_scipy(m, n, Ap, Ai, Ax):
    return sp.csc_matrix((Ax, Ai, Ap), shape=(m, n))


def symmetrized(n, Ap, Ai):
    A = sp.csc_matrix((np.ones(Ap[n]), Ai[:Ap[n]], Ap), shape=(n, n))
    S = (A + A.T).tocsc()
    S.sort_indices()
    return S.indptr.astype(np.int32), S.indices.astype(np.int32)
