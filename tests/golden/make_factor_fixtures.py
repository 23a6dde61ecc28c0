#!/usr/bin/env python3
"""Generate tests/golden/factor_fixtures.npz: oracle-INDEPENDENT known answers for BASELINE configs 1 and 2.

The reference holds nothing to pin the factor/solve path to, and until this file every
hot-path GPU check went through oracle/liboracle.so.  These fixtures are computed WITHOUT the oracle's C code:

  * the pivot order q is the library's own host-side ordering (cs3_analyze needs no GPU), stored as data;
  * L, U (P A Q = L U with P = Q', diagonal pivots) come from a dense Doolittle elimination of A[q][:, q] in
    numpy longdouble (x87 80-bit here: 64-bit significand), rounded to float64 -- good to ~1e-16 relative, three
    orders past the 1e-10 parity bar;
  * the Cholesky factor of the symmetrised matrix the same way;
  * the symbolic fill pattern comes from a boolean elimination of the same dense matrix (no cancellation);
  * x solves A x = b by the same extended-precision factors, and is cross-checked here against
    numpy.linalg.solve (LAPACK, partial pivoting) and scipy.sparse.linalg.splu (SuperLU) before it is saved.

Inputs are the seeded generators of csparse3_amd/synth.py: toy10 (config 1) and jacobian_config2 (config 2 at
its stated ~400 x 400 / ~3k nnz), the 118-bus-sized jacobian_like, a 1500-column grid Jacobian (fronts up to the
LDS-resident block kernels) and a 500-column matrix with a dense 180-column block (the blocked big-front path).

    python tests/golden/make_factor_fixtures.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from csparse3_amd import csc_hip as hip, synth                      # noqa: E402  (host-side analysis only)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "factor_fixtures.npz")
LD = np.longdouble


def dense_lu_nopivot(A):
    """Doolittle LU without pivoting in longdouble.  -> (L unit lower, U upper), longdouble."""
    n = A.shape[0]
    W = A.astype(LD).copy()
    for k in range(n):
        W[k + 1:, k] /= W[k, k]
        W[k + 1:, k + 1:] -= np.outer(W[k + 1:, k], W[k, k + 1:])
    L = np.tril(W, -1) + np.eye(n, dtype=LD)
    U = np.triu(W)
    return L, U


def dense_chol(A):
    n = A.shape[0]
    W = np.tril(A.astype(LD)).copy()
    for k in range(n):
        W[k, k] = np.sqrt(W[k, k])
        W[k + 1:, k] /= W[k, k]
        for j in range(k + 1, n):
            W[j:, j] -= W[j:, k] * W[j, k]
    return W


def symbolic_fill(pattern):
    """Boolean elimination of a structurally symmetric pattern: the exact (cancellation-free) pattern of L + U."""
    P = pattern.copy()
    n = P.shape[0]
    for k in range(n):
        rows = np.flatnonzero(P[k + 1:, k]) + k + 1
        cols = np.flatnonzero(P[k, k + 1:]) + k + 1
        P[np.ix_(rows, cols)] = True
    return P


def case(tag, m, n, Ap, Ai, Ax, rng, out):
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    Ad = A.toarray()
    with hip.Factorization(m, n, Ap, Ai, hip.CS3_LU, hip.ORDER_AMD) as F:
        o = F.ordering()
        inf = F.info
        nnz_l, nnz_u = int(inf.nnz_l), int(inf.nnz_u)
    q = o["q"]
    PAQ = Ad[np.ix_(q, q)]
    L, U = dense_lu_nopivot(PAQ)
    # the diagonal is an acceptable pivot for cs_lu's tol = 1e-3 everywhere (|l_ik| <= 1 / tol)
    assert np.abs(L).max() < 1e3
    pat = symbolic_fill((Ad != 0)[np.ix_(q, q)] | (Ad != 0).T[np.ix_(q, q)] | np.eye(n, dtype=bool))
    assert np.tril(pat).sum() == nnz_l and np.triu(pat).sum() == nnz_u, "fill pattern disagrees with the analysis"
    b = rng.standard_normal(n)
    # x = Q (U \ (L \ (P b))) in longdouble
    y = b[q].astype(LD)
    for k in range(n):
        y[k + 1:] -= L[k + 1:, k] * y[k]
    for k in range(n - 1, -1, -1):
        y[k] /= U[k, k]
        y[:k] -= U[:k, k] * y[k]
    x = np.empty(n)
    x[q] = y.astype(np.float64)
    x_lapack = np.linalg.solve(Ad, b)
    x_superlu = spl.splu(A).solve(b)
    scale = np.abs(x).max()
    assert np.abs(x - x_lapack).max() <= 1e-12 * scale and np.abs(x - x_superlu).max() <= 1e-12 * scale
    resid = np.abs(PAQ - (L @ U).astype(np.float64)).max() / np.abs(Ad).max()
    assert resid < 1e-15, resid
    # Cholesky of the symmetrised matrix (config 5's kind on a config-1/2-sized case)
    S = (A + A.T).tocsc(); S.sort_indices()
    Sd = S.toarray()
    with hip.Factorization(m, n, S.indptr, S.indices, hip.CS3_CHOLESKY, hip.ORDER_AMD) as G:
        qc = G.ordering()["q"]
    Lc = dense_chol(Sd[np.ix_(qc, qc)])
    assert np.abs(Sd[np.ix_(qc, qc)] - (Lc @ Lc.T).astype(np.float64)).max() / np.abs(Sd).max() < 1e-15
    out.update({
        tag + "_n": n, tag + "_Ap": Ap, tag + "_Ai": Ai, tag + "_Ax": Ax, tag + "_b": b,
        tag + "_q_amd": o["q_amd"], tag + "_q": q, tag + "_pinv": o["pinv"],
        tag + "_parent": o["parent"], tag + "_post": o["post"], tag + "_colcount": o["colcount"],
        tag + "_L": L.astype(np.float64), tag + "_U": U.astype(np.float64), tag + "_pattern": np.packbits(pat),
        tag + "_x": x,
        tag + "_S_Ap": S.indptr.astype(np.int32), tag + "_S_Ai": S.indices.astype(np.int32), tag + "_S_Ax": S.data,
        tag + "_S_q": qc, tag + "_S_L": Lc.astype(np.float64),
    })
    print("%-12s n=%4d nnz(A)=%5d nnz(L)=%5d nnz(U)=%5d  |PAQ-LU|/|A|=%.1e  x vs LAPACK %.1e  vs SuperLU %.1e"
          % (tag, n, Ap[n], nnz_l, nnz_u, resid, np.abs(x - x_lapack).max() / scale, np.abs(x - x_superlu).max() / scale))


def main():
    assert np.finfo(LD).nmant >= 63, "longdouble is not extended precision here"
    rng = np.random.default_rng(20240118)
    out = {}
    m, n, Ap, Ai, Ax, b, xt = synth.toy10()
    case("toy10", m, n, Ap, Ai, Ax, rng, out)
    out["toy10_b_known"], out["toy10_x_known"] = b, xt                # b = A (1..10): the answer is 1..10
    case("jac118", *synth.jacobian_like(), rng, out)
    case("config2", *synth.jacobian_config2(), rng, out)
    # beyond the one-wave kernels: fronts of order 33-136 (grid1500) and a dense 180-pivot root on the blocked
    # big-front path (block500) -- the same extended-precision elimination, a few seconds each
    case("grid1500", *synth.grid_jacobian(n=1500, seed=15), rng, out)
    case("block500", *synth.dense_block_matrix(n=500, nd=180, seed=3), rng, out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
