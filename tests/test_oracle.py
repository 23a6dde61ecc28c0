"""The CPU oracle against (a) the reference's own golden data and (b) mathematics.

(a) PINNED: the substrate functions that exist in the reference are checked against
    tests/golden/substrate.npz -- outputs of the reference's own Python kernels
    (tests/golden/make_golden.py) -- and against the hand-written CSC->CSR known
    answer of /root/reference/src/test/cscs_to_csr_test.py:23-25.  Exact equality,
    as the reference's tests do (test1_operations.py:55-61).
(b) PARITY UNPINNED: AMD / etree / counts / LU / Cholesky / triangular solves do not
    exist in the reference.  They are checked against their
    definitions: brute-force etree and column counts from a dense symbolic
    factorisation, P A Q = L U, L L' = P A P', residuals, and SciPy's SuperLU /
    LAPACK as independent solvers.
"""
import os

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from csparse3_amd import synth
from helpers import csc_to_scipy, symmetrized

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "substrate.npz"))


# ------------------------------------------------------------------ pinned --

def test_csc_to_csr_reference_known_answer(orc):
    """The one hand-written golden vector in the reference (cscs_to_csr_test.py:13-25)."""
    data = np.array([4, 3, 3, 9, 7, 8, 4, 8, 8, 9], dtype=np.float64)
    indices = np.array([0, 1, 3, 1, 2, 4, 5, 2, 3, 4], dtype=np.int32)
    indptr = np.array([0, 3, 7, 10], dtype=np.int32)
    Bp = np.zeros(7, dtype=np.int32); Bi = np.empty(10, dtype=np.int32); Bx = np.empty(10)
    orc.csc_to_csr(6, 3, indptr, indices, data, Bp, Bi, Bx)
    assert (Bp == np.array([0, 1, 3, 5, 7, 9, 10])).all()
    assert (Bi == np.array([0, 0, 1, 1, 2, 0, 2, 1, 2, 1])).all()
    assert (Bx == np.array([4, 3, 9, 7, 8, 3, 8, 8, 9, 4], dtype=np.float64)).all()


def test_docstring_matrix_against_reference_outputs(orc):
    m, n = int(GOLD["doc_m"]), int(GOLD["doc_n"])
    Ap, Ai, Ax = GOLD["doc_Ap"], GOLD["doc_Ai"], GOLD["doc_Ax"]
    _, _, Tp, Ti, Tx = orc.csc_transpose(m, n, Ap, Ai, Ax)
    assert (Tp == GOLD["doc_t_p"]).all() and (Ti == GOLD["doc_t_i"]).all() and (Tx == GOLD["doc_t_x"]).all()
    assert (orc.csc_mat_vec_ff(m, n, Ap, Ai, Ax, np.array([1.0, 2.0, 3.0])) == GOLD["doc_matvec"]).all()
    assert orc.csc_norm(n, Ap, Ax) == float(GOLD["doc_norm"]) == 28.0


@pytest.mark.parametrize("tag", ["r1", "r2", "r3"])
def test_substrate_against_reference_outputs(orc, tag):
    g = lambda k: GOLD[tag + "_" + k]
    m, n = int(g("m")), int(g("n"))
    Ap, Ai, Ax = g("Ap"), g("Ai"), g("Ax")
    _, _, Cp, Ci, Cx = orc.csc_add_ff(m, n, Ap, Ai, Ax, m, n, g("Bp"), g("Bi"), g("Bx"), 1.5, -0.25)
    assert (Cp == g("add_p")).all() and (Ci == g("add_i")).all() and (Cx == g("add_x")).all()
    _, _, Tp, Ti, Tx = orc.csc_transpose(m, n, Ap, Ai, Ax)
    assert (Tp == g("t_p")).all() and (Ti == g("t_i")).all() and (Tx == g("t_x")).all()
    Rp = np.zeros(m + 1, dtype=np.int32); Ri = np.empty(Ap[n], dtype=np.int32); Rx = np.empty(Ap[n])
    orc.csc_to_csr(m, n, Ap, Ai, Ax, Rp, Ri, Rx)
    assert (Rp == g("csr_p")).all() and (Ri == g("csr_i")).all() and (Rx == g("csr_x")).all()
    assert (orc.csc_mat_vec_ff(m, n, Ap, Ai, Ax, g("x")) == g("matvec")).all()
    assert orc.csc_norm(n, Ap, Ax) == float(g("norm"))
    c = np.diff(Ap).astype(np.int32); p = np.zeros(n + 1, dtype=np.int32)
    assert orc.csc_cumsum_i(p, c, n) == int(g("cumsum_tot"))
    assert (p == g("cumsum_p")).all() and (c == g("cumsum_c")).all()
    w = np.zeros(m, dtype=np.int32); xw = np.zeros(m); Cw = np.zeros(m, dtype=np.int32)
    nz = orc.csc_scatter_f(Ap, Ai, Ax, 0, 1.0, w, xw, 1, Cw, 0)
    nz = orc.csc_scatter_f(Ap, Ai, Ax, 1, 2.5, w, xw, 1, Cw, nz)
    assert nz == int(g("scatter_nz")) and (w == g("scatter_w")).all()
    assert (xw == g("scatter_x")).all() and (Cw[:nz] == g("scatter_Ci")).all()
    _, _, Kp, Ki, Kx = orc.coo_to_csc(m, n, g("coo_i"), g("coo_j"), g("coo_x"), len(g("coo_i")))
    assert (Kp == g("coo_p")).all() and (Ki == g("coo_ci")).all() and (Kx == g("coo_cx")).all()


def test_stack_4_by_4_against_reference_output(orc):
    g = lambda k: GOLD["st_" + k]
    am, an, bn, cm = int(g("am")), int(g("an")), int(g("bn")), int(g("cm"))
    m, n, Pi, Pp, Px = orc.csc_stack_4_by_4_ff(am, an, g("Ai"), g("Ap"), g("Ax"), am, bn, g("Bi"), g("Bp"), g("Bx"),
                                               cm, an, g("Ci"), g("Cp"), g("Cx"), cm, bn, g("Di"), g("Dp"), g("Dx"))
    assert (m, n) == (int(g("m")), int(g("n")))
    assert (Pi == g("i")).all() and (Pp == g("p")).all() and (Px == g("x")).all()
    with pytest.raises(AssertionError):     # the reference asserts on incompatible blocks (csc_numba.py:679-682)
        orc.csc_stack_4_by_4_ff(am + 1, an, g("Ai"), g("Ap"), g("Ax"), am, bn, g("Bi"), g("Bp"), g("Bx"),
                                cm, an, g("Ci"), g("Cp"), g("Cx"), cm, bn, g("Di"), g("Dp"), g("Dx"))


# ---------------------------------------------------------------- unpinned --

def _small_cases():
    m, n, Ap, Ai, Ax, b, xt = synth.toy10()
    yield "toy10", (m, n, Ap, Ai, Ax)
    yield "jac", synth.jacobian_like()
    yield "grid300", synth.grid_jacobian(n=300, seed=3)
    yield "dense60", synth.dense_block_matrix(n=120, nd=60, seed=4)


SMALL = dict(_small_cases())


def _dense_symbolic(n, pattern):
    """Boolean Cholesky fill of a symmetric pattern by definition (O(n^3))."""
    L = np.tril(pattern | pattern.T | np.eye(n, dtype=bool))
    for k in range(n):
        rows = np.flatnonzero(L[k + 1:, k]) + k + 1
        L[np.ix_(rows, rows)] |= np.tril(np.ones((len(rows), len(rows)), dtype=bool))
    return L


@pytest.mark.parametrize("name", list(SMALL))
def test_amd_is_permutation_with_reasonable_fill(orc, name):
    m, n, Ap, Ai, Ax = SMALL[name]
    q = orc.csc_amd_f(1, n, n, Ap, Ai)
    assert sorted(q.tolist()) == list(range(n))
    assert (orc.csc_amd_f(0, n, n, Ap, Ai) == np.arange(n)).all()
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    nat = spla.splu(A, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    mmd = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    Lp = orc.csc_lu_f(n, n, Ap, Ai, Ax, q, 1e-3)[0]
    # AMD is not unique, so the gate is quality: no worse than 1.3 x SuperLU's MMD(A'+A) or the natural order
    assert Lp[n] <= 1.3 * min(mmd.L.nnz, nat.L.nnz) + n


@pytest.mark.parametrize("name", list(SMALL))
def test_etree_post_counts_match_definitions(orc, name):
    m, n, Ap, Ai, Ax = SMALL[name]
    q = orc.csc_amd_f(1, n, n, Ap, Ai)
    pinv = orc.csc_pinv(q)
    Sp, Si = symmetrized(n, Ap, Ai)
    _, _, Cp, Ci, _ = orc.csc_symperm(n, Sp, Si, None, pinv)
    parent = orc.csc_etree_f(n, Cp, Ci)
    post = orc.csc_post_f(n, parent)
    cc = orc.csc_counts_f(n, Cp, Ci, parent, post)
    P = np.zeros((n, n), dtype=bool)
    cols = np.repeat(np.arange(n), np.diff(Cp))
    P[Ci, cols] = True
    L = _dense_symbolic(n, P)
    want_parent = np.array([(np.flatnonzero(L[j + 1:, j])[0] + j + 1) if L[j + 1:, j].any() else -1
                            for j in range(n)])
    assert (parent == want_parent).all()           # parent[j] = min{i > j : L(i,j) != 0}: unique
    assert (cc == L.sum(axis=0)).all()             # column counts incl. the diagonal
    assert sorted(post.tolist()) == list(range(n))
    pos = np.empty(n, dtype=int); pos[post] = np.arange(n)
    assert all(parent[j] < 0 or pos[j] < pos[parent[j]] for j in range(n))   # children before parents


@pytest.mark.parametrize("name", list(SMALL))
@pytest.mark.parametrize("tol", [1.0, 1e-3])
def test_lu_reconstructs_matrix(orc, name, tol):
    m, n, Ap, Ai, Ax = SMALL[name]
    q = orc.csc_amd_f(1, n, n, Ap, Ai)
    Lp, Li, Lx, Up, Ui, Ux, pinv = orc.csc_lu_f(n, n, Ap, Ai, Ax, q, tol)
    A = csc_to_scipy(m, n, Ap, Ai, Ax).toarray()
    L = sp.csc_matrix((Lx, Li, Lp), shape=(n, n)).toarray()
    U = sp.csc_matrix((Ux, Ui, Up), shape=(n, n)).toarray()
    assert np.allclose(np.triu(L, 1), 0) and np.allclose(np.diag(L), 1) and np.allclose(np.tril(U, -1), 0)
    assert (Li[Lp[:-1]] == np.arange(n)).all() and (Ui[Up[1:] - 1] == np.arange(n)).all()   # diag first / last
    PAQ = np.empty_like(A)
    PAQ[pinv, :] = A
    PAQ = PAQ[:, q]
    assert np.abs(PAQ - L @ U).max() <= 1e-12 * np.abs(A).sum(axis=0).max()
    assert sorted(pinv.tolist()) == list(range(n))


def test_lu_partial_pivoting_leaves_the_diagonal(orc):
    A = np.array([[1e-8, 1.0, 0.0], [1.0, 1.0, 1.0], [0.0, 1.0, 3.0]])
    S = sp.csc_matrix(A)
    Lp, Li, Lx, Up, Ui, Ux, pinv = orc.csc_lu_f(3, 3, S.indptr, S.indices, S.data, None, 1.0)
    assert pinv[1] == 0                              # row 1 holds the largest entry of column 0
    x = orc.csc_lusol_f(0, 3, S.indptr, S.indices, S.data, np.array([1.0, 2.0, 3.0]), 1.0)
    assert np.allclose(A @ x, [1, 2, 3])
    with pytest.raises(orc.SingularMatrix):
        Z = sp.csc_matrix(np.array([[1.0, 2.0], [2.0, 4.0]]))
        orc.csc_lu_f(2, 2, Z.indptr, Z.indices, Z.data, None, 1.0)


@pytest.mark.parametrize("name", list(SMALL))
def test_lusol_against_superlu_and_lapack(orc, name):
    m, n, Ap, Ai, Ax = SMALL[name]
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    b = np.random.default_rng(5).standard_normal(n)
    x = orc.csc_lusol_f(1, n, Ap, Ai, Ax, b, 1.0)
    assert np.abs(x - spla.spsolve(A, b)).max() <= 1e-11 * np.abs(x).max()
    assert np.abs(x - np.linalg.solve(A.toarray(), b)).max() <= 1e-11 * np.abs(x).max()


@pytest.mark.parametrize("n", [40, 300])
def test_cholesky_against_lapack(orc, n):
    ei, ej = synth.spd_grid_pattern(n, seed=n)
    m, n, Ap, Ai, Ax = synth.spd_grid_matrix(n, ei, ej, seed=n + 1)
    pinv, parent, cp, post = orc.csc_schol_f(1, n, Ap, Ai)
    Lp, Li, Lx = orc.csc_chol_f(n, Ap, Ai, Ax, pinv, parent, cp)
    assert (Lp == cp).all()                          # symbolic column counts are exact
    L = sp.csc_matrix((Lx, Li, Lp), shape=(n, n)).toarray()
    A = csc_to_scipy(m, n, Ap, Ai, Ax).toarray()
    PAP = np.empty_like(A)
    PAP[np.ix_(pinv, pinv)] = A
    assert np.abs(L @ L.T - PAP).max() <= 1e-12 * np.abs(A).max()
    assert np.abs(L - sla.cholesky(PAP, lower=True)).max() <= 1e-11 * np.abs(L).max()   # unique for SPD
    b = np.arange(1.0, n + 1.0)
    assert np.abs(orc.csc_cholsol_f(1, n, Ap, Ai, Ax, b) - np.linalg.solve(A, b)).max() <= 1e-10 * n
    Ax2 = Ax.copy(); Ax2[Ap[:-1]] = -1.0             # break positive definiteness
    with pytest.raises(orc.NotPositiveDefinite):
        orc.csc_chol_f(n, Ap, Ai, Ax2, pinv, parent, cp)


def test_triangular_solves_against_dense(orc):
    rng = np.random.default_rng(9)
    n = 50
    Ld = np.tril(rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.2), -1) + np.diag(rng.uniform(1, 2, n))
    L = sp.csc_matrix(Ld); L.sort_indices()
    U = sp.csc_matrix(Ld.T); U.sort_indices()
    b = rng.standard_normal(n)
    for fn, G, D in [(orc.csc_lsolve_f, L, Ld), (orc.csc_usolve_f, U, Ld.T)]:
        x = b.copy(); fn(n, G.indptr, G.indices, G.data, x)
        assert np.abs(D @ x - b).max() <= 1e-12 * np.abs(x).max() * np.abs(D).sum(axis=1).max()
    x = b.copy(); orc.csc_ltsolve_f(n, L.indptr, L.indices, L.data, x)
    assert np.allclose(Ld.T @ x, b)
    x = b.copy(); orc.csc_utsolve_f(n, U.indptr, U.indices, U.data, x)
    assert np.allclose(Ld @ x, b)


# ---- section 8f neighbours: sub-matrix extraction, duplicates, islands (golden = the reference's own outputs) ----

@pytest.mark.parametrize("tag", ["sub1", "sub2", "sub3"])
def test_sub_matrix_matches_reference(orc, tag):
    Ap, Ai, Ax = GOLD["r1_Ap"], GOLD["r1_Ai"], GOLD["r1_Ax"]
    nz, Bp, Bi, Bx = orc.csc_sub_matrix(40, int(Ap[40]), Ap, Ai, Ax, GOLD[tag + "_rows"], GOLD[tag + "_cols"])
    assert nz == int(GOLD[tag + "_nz"])
    assert np.array_equal(Bp, GOLD[tag + "_p"]) and np.array_equal(Bi, GOLD[tag + "_i"]) and np.array_equal(Bx, GOLD[tag + "_x"])


def test_duplicates_and_unsorted_rows_keep_the_reference_order(orc):
    Ap, Ai, Ax = GOLD["dup_Ap"], GOLD["dup_Ai"], GOLD["dup_Ax"]
    _, _, Tp, Ti, Tx = orc.csc_transpose(4, 4, Ap, Ai, Ax)
    assert np.array_equal(Tp, GOLD["dup_t_p"]) and np.array_equal(Ti, GOLD["dup_t_i"]) and np.array_equal(Tx, GOLD["dup_t_x"])
    _, _, Cp, Ci, Cx = orc.csc_add_ff(4, 4, Ap, Ai, Ax, 4, 4, Ap, Ai, Ax, 2.0, 0.5)
    assert np.array_equal(Cp, GOLD["dup_add_p"]) and np.array_equal(Ci, GOLD["dup_add_i"]) and np.array_equal(Cx, GOLD["dup_add_x"])
    _, _, Kp, Ki, Kx = orc.coo_to_csc(4, 4, GOLD["dup_coo_i"], GOLD["dup_coo_j"], GOLD["dup_coo_x"], 9)
    assert np.array_equal(Kp, GOLD["dup_coo_p"]) and np.array_equal(Ki, GOLD["dup_coo_ci"]) and np.array_equal(Kx, GOLD["dup_coo_cx"])
    assert orc.csc_norm(4, Ap, Ax) == float(GOLD["dup_norm"])


def test_find_islands_matches_reference(orc):
    isl = orc.find_islands(30, GOLD["isl_Ap"], GOLD["isl_Ai"])
    assert len(isl) == int(GOLD["isl_count"]) and [len(x) for x in isl] == list(GOLD["isl_sizes"])
    assert np.array_equal(np.concatenate(isl), GOLD["isl_flat"])


@pytest.mark.parametrize("tag", ["isd0", "isd1", "isd2"])
def test_find_islands_unsymmetric_patterns_match_reference(orc, tag):
    """find_islands follows column -> row edges only (csc_numba.py:768-800): on these patterns the islands are not the
    connected components.  The oracle reproduces the reference's own output (tests/golden/make_golden.py)."""
    n = int(GOLD[tag + "_n"])
    isl = orc.find_islands(n, GOLD[tag + "_Ap"], GOLD[tag + "_Ai"])
    assert [len(x) for x in isl] == list(GOLD[tag + "_sizes"])
    assert np.array_equal(np.concatenate(isl), GOLD[tag + "_flat"])
