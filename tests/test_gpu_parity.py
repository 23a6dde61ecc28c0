"""GPU parity: the HIP path through the C ABI against the CPU oracle.

Integer outputs (orderings, trees, patterns) bit-exact; factor values and
solutions within 1e-10 relative (BASELINE.json north_star).  The oracle's
parity with the reference is unpinned for these functions:
what is checked here is HIP path == oracle, and oracle == mathematics
(tests/test_oracle.py).
"""
import numpy as np
import pytest

from csparse3_amd import synth
from helpers import RTOL, assert_factor_equal, csc_to_scipy, rel_err

pytestmark = pytest.mark.gpu


def _cases():
    m, n, Ap, Ai, Ax, b, xt = synth.toy10()
    yield "toy10", (m, n, Ap, Ai, Ax)
    yield "jacobian118", synth.jacobian_like()
    yield "config2", synth.jacobian_config2()                                   # BASELINE configs[1] at its stated size
    yield "grid2k", synth.grid_jacobian(n=2000, seed=7)
    yield "grid20k", synth.grid_jacobian(n=20000, seed=11)
    yield "denseblock300", synth.dense_block_matrix(n=700, nd=300, seed=1)      # fronts beyond the LDS
    yield "denseblock150", synth.dense_block_matrix(n=400, nd=150, seed=2)


CASES = dict(_cases())


@pytest.mark.parametrize("name", list(CASES))
def test_lu_factors_match_oracle(gpu, orc, name):
    m, n, Ap, Ai, Ax = CASES[name]
    Lp, Li, Lx, Up, Ui, Ux, pinv, q = gpu.csc_lu_f(m, n, Ap, Ai, Ax, tol=1e-3)
    oLp, oLi, oLx, oUp, oUi, oUx, opinv = orc.csc_lu_f(n, n, Ap, Ai, Ax, q, 1e-3)
    assert np.array_equal(pinv, opinv), "oracle chose off-diagonal pivots"
    assert_factor_equal(n, (Lp, Li, Lx), (oLp, oLi, oLx), name + " L")
    assert_factor_equal(n, (Up, Ui, Ux), (oUp, oUi, oUx), name + " U")
    # CSparse layout: L diagonal first (unit), U diagonal last
    assert np.array_equal(Li[Lp[:-1]], np.arange(n)) and np.all(Lx[Lp[:-1]] == 1.0)
    assert np.array_equal(Ui[Up[1:] - 1], np.arange(n))


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("k", [1, 5])
def test_lusol_matches_oracle(gpu, orc, name, k):
    m, n, Ap, Ai, Ax = CASES[name]
    rng = np.random.default_rng(3)
    B = rng.standard_normal((n, k)) if k > 1 else rng.standard_normal(n)
    X = gpu.csc_lusol_f(1, m, n, Ap, Ai, Ax, B, tol=1e-3)
    cols = [B] if k == 1 else [np.ascontiguousarray(B[:, t]) for t in range(k)]
    want = np.stack([orc.csc_lusol_f(1, n, Ap, Ai, Ax, c, 1e-3) for c in cols], axis=-1)
    if k == 1:
        want = want[:, 0]
    assert rel_err(X, want) <= RTOL
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    R = A @ X - B
    assert np.abs(R).max() <= 1e-12 * (abs(A).sum(axis=0).max() * np.abs(X).max() + np.abs(B).max())


# --------------------------------------------------------------- Cholesky --

def _spd_cases():
    for n in (200, 4000):
        ei, ej = synth.spd_grid_pattern(n, seed=n)
        yield "spd%d" % n, synth.spd_grid_matrix(n, ei, ej, seed=n + 1)
    m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=500, nd=220, seed=3)
    import scipy.sparse as sp
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    S = (A + A.T).tocsc(); S.sort_indices()               # symmetric, still diagonally dominant => SPD
    yield "spd_denseblock", (n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy())


SPD = dict(_spd_cases())


def _oracle_chol(orc, n, Ap, Ai, Ax, q):
    pinv = orc.csc_pinv(q)
    _, _, Cp, Ci, _ = orc.csc_symperm(n, Ap, Ai, None, pinv)
    parent = orc.csc_etree_f(n, Cp, Ci)
    post = orc.csc_post_f(n, parent)
    cnt = orc.csc_counts_f(n, Cp, Ci, parent, post)
    cp = np.zeros(n + 1, dtype=np.int32); cp[1:] = np.cumsum(cnt)
    return orc.csc_chol_f(n, Ap, Ai, Ax, pinv, parent, cp)


@pytest.mark.parametrize("name", list(SPD))
def test_cholesky_factor_and_solve_match_oracle(gpu, orc, name):
    m, n, Ap, Ai, Ax = SPD[name]
    with gpu.Factorization(m, n, Ap, Ai, kind=gpu.CS3_CHOLESKY) as F:
        F.factor(Ax)
        Lp, Li, Lx, _, _, _ = F.factors()
        q = F.ordering()["q"]
        b = np.random.default_rng(1).standard_normal((n, 3))
        X = F.solve(b)
    assert_factor_equal(n, (Lp, Li, Lx), _oracle_chol(orc, n, Ap, Ai, Ax, q), name + " L")
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    assert np.abs(A @ X - b).max() <= 1e-11 * np.abs(b).max() * n
    # a triangle-only input gives the same factor (cs_chol reads the upper triangle)
    import scipy.sparse as sp
    T = sp.triu(A, format="csc"); T.sort_indices()
    with gpu.Factorization(m, n, T.indptr, T.indices, kind=gpu.CS3_CHOLESKY, q=F_q(q)) as G:
        G.factor(T.data)
        Lp2, Li2, Lx2, _, _, _ = G.factors()
    assert np.array_equal(Lp, Lp2) and np.array_equal(Li, Li2) and rel_err(Lx2, Lx) <= RTOL


def F_q(q):
    return np.asarray(q, dtype=np.int32)


def test_cholesky_rejects_indefinite_matrix(gpu):
    m, n, Ap, Ai, Ax = SPD["spd200"]
    bad = Ax.copy(); bad[Ap[:-1]] *= -1.0               # first entry of each column ... make the diagonal negative
    rows = Ai[:Ap[n]]; cols = np.repeat(np.arange(n), np.diff(Ap))
    bad = Ax.copy(); bad[rows == cols] = -np.abs(bad[rows == cols])
    with gpu.Factorization(m, n, Ap, Ai, kind=gpu.CS3_CHOLESKY) as F:
        with pytest.raises(gpu.NotPositiveDefinite):
            F.factor(bad)
        assert F.info.fail_col >= 0
        with pytest.raises(gpu.Cs3Error):
            F.solve(np.ones(n))                          # no valid factorisation to solve with


# ----------------------------------------------------- pivots and failures --

def test_static_pivot_rejection_is_reported(gpu, orc):
    m, n, Ap, Ai, Ax = CASES["jacobian118"]
    rows = Ai[:Ap[n]]; cols = np.repeat(np.arange(n), np.diff(Ap))
    with gpu.Factorization(m, n, Ap, Ai) as F:
        # a leaf of the elimination tree with entries below the diagonal: its pivot is A's own
        # diagonal entry, untouched by earlier eliminations
        Lp, Li = F.factors(values=False)[:2]
        k0 = next(k for k in range(n) if Lp[k + 1] - Lp[k] > 1 and k not in set(Li[:Lp[k]].tolist()))
        first = F.ordering()["q"][k0]
        weak = Ax.copy(); weak[(rows == cols) & (cols == first)] = 1e-9   # fails |pivot| >= tol * max|column|
        with pytest.raises(gpu.SingularMatrix):
            F.factor(weak, tol=1e-3)
        assert F.info.fail_col == k0
        # the oracle agrees that the diagonal is not acceptable there: it pivots off the diagonal
        opinv = orc.csc_lu_f(n, n, Ap, Ai, weak, F.ordering()["q"], 1e-3)[6]
        assert opinv[first] != k0
        F.factor(weak, tol=0.0)                          # the test can be switched off (cs_lu's tol -> 0)
        zero = Ax.copy(); zero[(rows == cols) & (cols == first)] = 0.0
        with pytest.raises(gpu.SingularMatrix):
            F.factor(zero, tol=0.0)                      # an exactly zero pivot is always an error
        F.factor(Ax, tol=1e-3)                           # the handle recovers with good values
        x = F.solve(np.ones(n))
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    assert np.abs(A @ x - 1.0).max() < 1e-10


# ------------------------------------------------------------ the sweeps ----

@pytest.mark.parametrize("name", ["jacobian118", "grid20k", "denseblock300"])
def test_lsolve_usolve_on_the_handle_match_oracle(gpu, orc, name):
    m, n, Ap, Ai, Ax = CASES[name]
    rng = np.random.default_rng(4)
    with gpu.Factorization(m, n, Ap, Ai) as F:
        F.factor(Ax, 1e-3)
        Lp, Li, Lx, Up, Ui, Ux = F.factors()
        b = rng.standard_normal(n)
        y = F.lsolve(b)
        x = F.usolve(y)
    wy = b.copy(); orc.csc_lsolve_f(n, Lp, Li, Lx, wy)
    wx = wy.copy(); orc.csc_usolve_f(n, Up, Ui, Ux, wx)
    assert rel_err(y, wy) <= RTOL and rel_err(x, wx) <= RTOL


@pytest.mark.parametrize("k", [1, 4])
def test_general_csc_triangular_solves_match_oracle(gpu, orc, k):
    """csc_lsolve_f / csc_usolve_f on caller-supplied factors (here: the oracle's own, rows unsorted)."""
    m, n, Ap, Ai, Ax = CASES["grid2k"]
    q = orc.csc_amd_f(1, n, n, Ap, Ai)
    Lp, Li, Lx, Up, Ui, Ux, pinv = orc.csc_lu_f(n, n, Ap, Ai, Ax, q, 1e-3)
    rng = np.random.default_rng(8)
    B = rng.standard_normal((n, k)) if k > 1 else rng.standard_normal(n)
    X = np.ascontiguousarray(B.copy()); gpu.csc_lsolve_f(n, Lp, Li, Lx, X)
    W = np.ascontiguousarray(B.copy())
    for t in range(k):
        col = np.ascontiguousarray(W[:, t]) if k > 1 else W
        orc.csc_lsolve_f(n, Lp, Li, Lx, col)
        if k > 1: W[:, t] = col
    assert rel_err(X, W) <= RTOL
    gpu.csc_usolve_f(n, Up, Ui, Ux, X)
    for t in range(k):
        col = np.ascontiguousarray(W[:, t]) if k > 1 else W
        orc.csc_usolve_f(n, Up, Ui, Ux, col)
        if k > 1: W[:, t] = col
    assert rel_err(X, W) <= RTOL


def test_matvec_is_bit_exact_with_the_reference_kernel(gpu):
    """csc_mat_vec_ff on the device reproduces the reference's own outputs exactly
    (same summation order, separate multiply and add roundings)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "substrate.npz"))
    for tag in ("r1", "r2", "r3"):
        m, n = int(G[tag + "_m"]), int(G[tag + "_n"])
        y = gpu.csc_mat_vec_ff(m, n, G[tag + "_Ap"], G[tag + "_Ai"], G[tag + "_Ax"], G[tag + "_x"])
        assert np.array_equal(y, G[tag + "_matvec"])
    y = gpu.csc_mat_vec_ff(6, 3, G["doc_Ap"], G["doc_Ai"], G["doc_Ax"], np.array([1.0, 2.0, 3.0]))
    assert np.array_equal(y, G["doc_matvec"])


# ------------------------------------------------------------------ batch ---

def test_batch_of_matrices_sharing_a_pattern(gpu, orc):
    n = 1500
    ei, ej = synth.spd_grid_pattern(n, seed=77)
    mats = [synth.spd_grid_matrix(n, ei, ej, seed=100 + i) for i in range(4)]
    m, n, Ap, Ai, _ = mats[0]
    AX = np.stack([mm[4] for mm in mats])
    B = np.random.default_rng(2).standard_normal((4, n, 2))
    for kind in (gpu.CS3_LU, gpu.CS3_CHOLESKY):
        with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=4) as F:
            F.factor(AX, 1e-3 if kind == gpu.CS3_LU else 0.0)
            X = F.solve(B)
            q = F.ordering()["q"]
            for i in range(4):
                Lp, Li, Lx, Up, Ui, Ux = F.factors(b=i)
                if kind == gpu.CS3_LU:
                    oL = orc.csc_lu_f(n, n, Ap, Ai, AX[i], q, 1e-3)
                    assert_factor_equal(n, (Lp, Li, Lx), oL[0:3], "batch L")
                    assert_factor_equal(n, (Up, Ui, Ux), oL[3:6], "batch U")
                else:
                    assert_factor_equal(n, (Lp, Li, Lx), _oracle_chol(orc, n, Ap, Ai, AX[i], q), "batch chol L")
                A = csc_to_scipy(m, n, Ap, Ai, AX[i])
                assert np.abs(A @ X[i] - B[i]).max() <= 1e-11 * n


@pytest.mark.parametrize("chol", [False, True])
@pytest.mark.parametrize("nd", [150, 260])
def test_batched_big_fronts_in_one_workgroup(gpu, orc, chol, nd):
    """A batch of 48 or more matrices sends fronts beyond the LDS to k_front_wg (one workgroup per front and
    matrix, one launch) and their sweeps to the single-launch block kernels: every matrix must match the oracle
    and the same matrix factorised alone (multi-launch path)."""
    import scipy.sparse as sp
    m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=nd + 350, nd=nd, seed=nd)
    kind = gpu.CS3_LU
    if chol:
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        S = (A + A.T).tocsc(); S.sort_indices()
        Ap, Ai, Ax = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy()
        kind = gpu.CS3_CHOLESKY
    nb = 50
    rng = np.random.default_rng(nd)
    AX = Ax[None, :] * (1.0 + 0.03 * rng.uniform(-1.0, 1.0, size=(nb, len(Ax))))
    if chol:                                           # keep the perturbed matrices symmetric: scale whole matrices instead
        AX = Ax[None, :] * (1.0 + rng.uniform(0.0, 1.0, size=(nb, 1)))
    B = rng.standard_normal((nb, n, 2))
    with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=nb) as F:
        assert F.info.max_front >= nd
        F.factor(AX, 1e-3 if not chol else 0.0)
        X = F.solve(B)
        q = F.ordering()["q"]
        facs = {i: F.factors(b=i) for i in (0, 7, nb - 1)}
        F.factor(AX, 1e-3 if not chol else 0.0)
        assert np.array_equal(F.solve(B), X)                              # run-to-run bitwise
    for i, (Lp, Li, Lx, Up, Ui, Ux) in facs.items():
        if chol:
            assert_factor_equal(n, (Lp, Li, Lx), _oracle_chol(orc, n, Ap, Ai, AX[i], q), "batched big chol %d" % i)
        else:
            oL = orc.csc_lu_f(n, n, Ap, Ai, AX[i], q, 1e-3)
            assert_factor_equal(n, (Lp, Li, Lx), oL[0:3], "batched big L %d" % i)
            assert_factor_equal(n, (Up, Ui, Ux), oL[3:6], "batched big U %d" % i)
    for i in range(nb):
        A = csc_to_scipy(m, n, Ap, Ai, AX[i])
        assert np.abs(A @ X[i] - B[i]).max() <= 1e-11 * n * max(1.0, np.abs(X[i]).max())
    with gpu.Factorization(m, n, Ap, Ai, kind=kind) as G:
        G.factor(AX[7], 1e-3 if not chol else 0.0)
        assert rel_err(X[7], G.solve(B[7])) <= 1e-12
    # a rejected pivot inside a big front of ONE matrix of the batch is reported
    if not chol:
        bad = AX.copy()
        rows = Ai[:Ap[n]]; cols = np.repeat(np.arange(n), np.diff(Ap))
        last = q[-1]
        bad[5, (rows == cols) & (cols == last)] = 0.0
        bad[5, (rows == last) | (cols == last)] = 0.0                     # a zero row and column: the last pivot is exactly zero
        with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=nb) as F:
            with pytest.raises(gpu.SingularMatrix):
                F.factor(bad, 1e-3)
            assert F.info.fail_col == n - 1


@pytest.mark.parametrize("chol", [False, True])
def test_interleaved_batches_lane_is_matrix(gpu, orc, chol):
    """64 or more matrices: fronts of order <= 32 live matrix-interleaved and run lane = matrix (k_front_il, k_fwd_il,
    k_bwd_il); larger fronts read their children's contribution blocks from the interleaved block.  130 matrices =
    two full groups and a partial one.  Every checked matrix equals the oracle (pattern bit-exact, values 1e-10) and the
    same matrix factorised alone; several right-hand sides go through the same kernels."""
    m, n, Ap, Ai, Ax = synth.grid_jacobian(n=3000, seed=13)
    kind = gpu.CS3_LU
    if chol:
        ei, ej = synth.spd_grid_pattern(3000, seed=21)
        m, n, Ap, Ai, Ax = synth.spd_grid_matrix(3000, ei, ej, seed=22)
        kind = gpu.CS3_CHOLESKY
    nb = 130
    rng = np.random.default_rng(130 + chol)
    scale = 1.0 + rng.uniform(0.0, 1.0, size=(nb, 1))
    AX = Ax[None, :] * scale
    if not chol:
        AX = AX * (1.0 + 0.01 * rng.uniform(-1.0, 1.0, size=AX.shape))         # stays diagonally dominant
    B = rng.standard_normal((nb, n, 3))
    tol = 0.0 if chol else 1e-3
    with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=nb) as F:
        F.factor(AX, tol)
        X = F.solve(B)
        x1 = F.solve(np.ascontiguousarray(B[:, :, :1]))
        Y = F.usolve(F.lsolve(B))
        q = F.ordering()["q"]
        facs = {i: F.factors(b=i) for i in (0, 63, 64, 127, 128, 129)}
        F.factor(AX, tol)
        assert np.array_equal(F.solve(B), X)                                     # run-to-run bitwise
    assert rel_err(x1[:, :, 0], X[:, :, 0]) <= 1e-13
    for i, (Lp, Li, Lx, Up, Ui, Ux) in facs.items():
        if chol:
            assert_factor_equal(n, (Lp, Li, Lx), _oracle_chol(orc, n, Ap, Ai, AX[i], q), "interleaved chol %d" % i)
        else:
            oL = orc.csc_lu_f(n, n, Ap, Ai, AX[i], q, 1e-3)
            assert np.array_equal(oL[6], np.argsort(q).astype(np.int32))
            assert_factor_equal(n, (Lp, Li, Lx), oL[0:3], "interleaved L %d" % i)
            assert_factor_equal(n, (Up, Ui, Ux), oL[3:6], "interleaved U %d" % i)
    for i in range(nb):
        A = csc_to_scipy(m, n, Ap, Ai, AX[i])
        bound = 1e-12 * (abs(A).sum(axis=0).max() * np.abs(X[i]).max() + np.abs(B[i]).max())
        assert np.abs(A @ X[i] - B[i]).max() <= bound, i
    with gpu.Factorization(m, n, Ap, Ai, kind=kind) as G:
        G.factor(AX[64], tol)
        assert rel_err(X[64], G.solve(B[64])) <= 1e-12
        assert rel_err(Y[64], G.usolve(G.lsolve(B[64]))) <= 1e-12                # the sweeps alone, in pivot order
    # a bad pivot in ONE matrix of a group is reported (here inside an interleaved leaf front)
    rows = Ai[:Ap[n]]; cols = np.repeat(np.arange(n), np.diff(Ap))
    bad = AX.copy()
    first = q[0]
    bad[77, (rows == cols) & (cols == first)] = 0.0 if not chol else -1.0
    with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=nb) as F:
        with pytest.raises(gpu.NotPositiveDefinite if chol else gpu.SingularMatrix):
            F.factor(bad, tol)
        assert F.info.fail_col == 0
        with pytest.raises(gpu.Cs3Error):
            import torch
            buf = torch.empty(8, dtype=torch.float64, device="cuda")
            F.export_factor_dev(buf.data_ptr())                                  # not offered for interleaved batches


# ------------------------------------------- full-size, size-independent ----

def test_config3_full_size_properties(gpu):
    """50k x 50k: residual, linearity of the solve, refactorisation idempotence,
    run-to-run bitwise reproducibility (no float atomics on the path)."""
    m, n, Ap, Ai, Ax = synth.grid_jacobian()
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    rng = np.random.default_rng(6)
    b1, b2 = rng.standard_normal(n), rng.standard_normal(n)
    with gpu.Factorization(m, n, Ap, Ai) as F:
        F.factor(Ax, 1e-3)
        x1, x2 = F.solve(b1), F.solve(b2)
        x12 = F.solve(2.0 * b1 - 3.0 * b2)
        L1 = F.factors()[2].copy()
        F.factor(Ax, 1e-3)
        assert np.array_equal(F.factors()[2], L1)                      # bitwise reproducible
        assert np.array_equal(F.solve(b1), x1)
        X = F.solve(np.stack([b1, b2], axis=1))
    scale = abs(A).sum(axis=0).max()
    for x, b in ((x1, b1), (x2, b2)):
        assert np.abs(A @ x - b).max() <= 1e-13 * (scale * np.abs(x).max() + np.abs(b).max())
    assert rel_err(x12, 2.0 * x1 - 3.0 * x2) <= 1e-12
    assert rel_err(X[:, 0], x1) <= 1e-13 and rel_err(X[:, 1], x2) <= 1e-13


def test_config4_slice_full_size_properties(gpu, orc):
    """BASELINE configs[3] on one GPU at its per-GPU size: the 50k matrix, factor once, 128 right-hand sides
    (1024 over 8 GPUs).  Residual of every column, agreement of a column with the single-RHS path, oracle parity
    of three columns, bitwise run-to-run reproducibility."""
    m, n, Ap, Ai, Ax = synth.grid_jacobian()
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    B = synth.grid_rhs(n, 128, seed=1024)
    with gpu.Factorization(m, n, Ap, Ai) as F:
        F.factor(Ax, 1e-3)
        X = F.solve(B)
        X2 = F.solve(B)
        x5 = F.solve(np.ascontiguousarray(B[:, 5]))
        Lp, Li, Lx, Up, Ui, Ux = F.factors()
        o = F.ordering()
    assert np.array_equal(X, X2)                                            # no float atomics anywhere
    scale = abs(A).sum(axis=0).max()
    R = A @ X - B
    assert np.abs(R).max() <= 1e-13 * (scale * np.abs(X).max() + np.abs(B).max())
    assert rel_err(X[:, 5], x5) <= 1e-12
    for j in (0, 77, 127):
        w = np.empty(n); w[o["pinv"]] = B[:, j]
        orc.csc_lsolve_f(n, Lp, Li, Lx, w)
        orc.csc_usolve_f(n, Up, Ui, Ux, w)
        want = np.empty(n); want[o["q"]] = w
        assert rel_err(X[:, j], want) <= RTOL


def test_config5_slice_full_size_properties(gpu, orc):
    """BASELINE configs[4] on one GPU at its per-GPU size: 64 SPD 5k x 5k matrices sharing a pattern (512 over
    8 GPUs), Cholesky factor + one solve each through the fused call.  Residual of every matrix, one matrix
    against the oracle (pattern bit-exact, values 1e-10), bitwise equality with the unbatched handle."""
    import torch
    n5, nmat = 5000, 64
    ei, ej = synth.spd_grid_pattern(n5, seed=5000)
    mats = [synth.spd_grid_matrix(n5, ei, ej, seed=5000 + i) for i in range(nmat)]
    m, n, Ap, Ai, _ = mats[0]
    AX = np.stack([mm[4] for mm in mats])
    B = np.random.default_rng(0).standard_normal((nmat, n, 1))
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    with gpu.Factorization(m, n, Ap, Ai, kind=gpu.CS3_CHOLESKY, batch=nmat) as F:
        d_ax = torch.from_numpy(AX).to(dev)
        d_x = torch.from_numpy(B.copy()).to(dev)
        F.factor_solve_dev(d_ax.data_ptr(), d_x.data_ptr(), 1, 0.0, sh)
        F.factor_status(sh)
        X = d_x.cpu().numpy()
        d_x2 = torch.from_numpy(B.copy()).to(dev)
        F.factor_dev(d_ax.data_ptr(), 0.0, sh)
        F.solve_dev(d_x2.data_ptr(), 1, sh)
        F.factor_status(sh)
        assert torch.equal(d_x, d_x2)                                       # fused == factor, then solve
        q = F.ordering()["q"]
        fac = {i: F.factors(b=i) for i in (0, 37, 63)}
    for i in range(nmat):
        A = csc_to_scipy(m, n, Ap, Ai, AX[i])
        r = np.abs(A @ X[i, :, 0] - B[i, :, 0]).max()
        assert r <= 1e-12 * (abs(A).sum(axis=0).max() * np.abs(X[i]).max() + np.abs(B[i]).max()), (i, r)
    for i, (Lp, Li, Lx, _, _, _) in fac.items():
        assert_factor_equal(n, (Lp, Li, Lx), _oracle_chol(orc, n, Ap, Ai, AX[i], q), "config-5 matrix %d" % i)
    with gpu.Factorization(m, n, Ap, Ai, kind=gpu.CS3_CHOLESKY) as G:      # the same matrix alone: same values to 1e-13
        G.factor(AX[37])
        x1 = G.solve(B[37])
    assert rel_err(X[37], x1) <= 1e-12


def test_config4_at_its_own_size_one_gpu(gpu, orc):
    """BASELINE configs[3] whole on one GPU: the 50k matrix, factor once, 1024 right-hand sides resident in HBM --
    the shape `bench.py`'s config-4 leg runs (fused permutations, GEMM sweeps, lane = right-hand-side sweeps).
    Residual of every column, bitwise run-to-run, agreement with the 1-RHS and 128-RHS paths, oracle parity of
    three columns."""
    import torch
    m, n, Ap, Ai, Ax = synth.grid_jacobian()
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    k = 1024
    B = synth.grid_rhs(n, k, seed=1024)
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    with gpu.Factorization(m, n, Ap, Ai) as F:
        F.factor(Ax, 1e-3)
        d_b = torch.from_numpy(B).to(dev)
        d_x = d_b.clone()
        F.solve_dev(d_x.data_ptr(), k, sh)
        d_x2 = d_b.clone()
        F.solve_dev(d_x2.data_ptr(), k, sh)
        torch.cuda.synchronize()
        assert torch.equal(d_x, d_x2)                                       # no float atomics anywhere
        X = d_x.cpu().numpy()
        x5 = F.solve(np.ascontiguousarray(B[:, 5]))
        X128 = F.solve(np.ascontiguousarray(B[:, 896:]))
        Lp, Li, Lx, Up, Ui, Ux = F.factors()
        o = F.ordering()
    scale = abs(A).sum(axis=0).max()
    for c0 in range(0, k, 256):                                             # residual of every column, 256 at a time
        R = A @ X[:, c0:c0 + 256] - B[:, c0:c0 + 256]
        assert np.abs(R).max() <= 1e-13 * (scale * np.abs(X).max() + np.abs(B).max())
    assert rel_err(X[:, 5], x5) <= 1e-12
    assert rel_err(X[:, 896:], X128) <= 1e-12
    for j in (0, 511, 1023):
        w = np.empty(n); w[o["pinv"]] = B[:, j]
        orc.csc_lsolve_f(n, Lp, Li, Lx, w)
        orc.csc_usolve_f(n, Up, Ui, Ux, w)
        want = np.empty(n); want[o["q"]] = w
        assert rel_err(X[:, j], want) <= RTOL


def test_config5_at_its_own_size_one_gpu(gpu, orc):
    """BASELINE configs[4] whole on one GPU: 512 SPD 5k x 5k matrices sharing a pattern, Cholesky factor + one solve
    each through the fused call (eight groups of 64 matrices on the interleaved region).  Fused == factor-then-solve
    bit for bit, residual of every matrix, three matrices against the oracle."""
    import torch
    n5, nmat = 5000, 512
    ei, ej = synth.spd_grid_pattern(n5, seed=5000)
    m, n, Ap, Ai, _ = synth.spd_grid_matrix(n5, ei, ej, seed=5000)
    AX = np.stack([synth.spd_grid_matrix(n5, ei, ej, seed=5000 + i)[4] for i in range(nmat)])
    B = np.random.default_rng(0).standard_normal((nmat, n, 1))
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    with gpu.Factorization(m, n, Ap, Ai, kind=gpu.CS3_CHOLESKY, batch=nmat) as F:
        d_ax = torch.from_numpy(AX).to(dev)
        d_x = torch.from_numpy(B.copy()).to(dev)
        F.factor_solve_dev(d_ax.data_ptr(), d_x.data_ptr(), 1, 0.0, sh)
        F.factor_status(sh)
        d_x2 = torch.from_numpy(B.copy()).to(dev)
        F.factor_dev(d_ax.data_ptr(), 0.0, sh)
        F.solve_dev(d_x2.data_ptr(), 1, sh)
        F.factor_status(sh)
        assert torch.equal(d_x, d_x2)
        X = d_x.cpu().numpy()
        q = F.ordering()["q"]
        fac = {i: F.factors(b=i) for i in (0, 300, 511)}
    import scipy.sparse as sp
    Aall = [csc_to_scipy(m, n, Ap, Ai, AX[i]) for i in range(nmat)]
    for i in range(nmat):
        r = np.abs(Aall[i] @ X[i, :, 0] - B[i, :, 0]).max()
        assert r <= 1e-12 * (abs(Aall[i]).sum(axis=0).max() * np.abs(X[i]).max() + np.abs(B[i]).max()), (i, r)
    for i, (Lp, Li, Lx, _, _, _) in fac.items():
        assert_factor_equal(n, (Lp, Li, Lx), _oracle_chol(orc, n, Ap, Ai, AX[i], q), "config-5 matrix %d" % i)


def test_fused_permutation_graphs_survive_rotating_buffers(gpu):
    """From 256 right-hand sides on the sweeps address the caller's array directly, so solve graphs are cached per
    (count, address), eight at a time: twelve different buffers, visited twice, must all come back solved (the ninth
    empties the cache; a stale graph would write into another buffer)."""
    import torch
    m, n, Ap, Ai, Ax = synth.grid_jacobian(n=2500, seed=12)
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    k = 256
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(5)
    Bs = [rng.standard_normal((n, k)) for _ in range(12)]
    with gpu.Factorization(m, n, Ap, Ai) as F:
        F.factor(Ax, 1e-3)
        bufs = [torch.from_numpy(b).to(dev) for b in Bs]
        first = []
        for x in bufs:
            F.solve_dev(x.data_ptr(), k, sh)
        torch.cuda.synchronize()
        first = [x.cpu().numpy() for x in bufs]
        for x, b in zip(bufs, Bs):                                       # second visit: fresh right-hand sides, same addresses
            x.copy_(torch.from_numpy(b))
        for x in reversed(bufs):
            F.solve_dev(x.data_ptr(), k, sh)
        torch.cuda.synchronize()
        second = [x.cpu().numpy() for x in bufs]
    scale = abs(A).sum(axis=0).max()
    for X, X2, b in zip(first, second, Bs):
        assert np.abs(A @ X - b).max() <= 1e-13 * (scale * np.abs(X).max() + np.abs(b).max())
        assert np.array_equal(X, X2)


def test_fused_step_graphs_per_solution_buffer_survive_rotating_buffers(gpu):
    """cs3_factor_solve_dev keeps a graph with the closing permutation inside for a caller that hands in the same X three
    times in a row (four such graphs at a time): six buffers, four calls each, then interleaved calls, must all hold the
    solution of their own right-hand side, bit-identical to the generic graph's (the first two calls)."""
    import torch
    m, n, Ap, Ai, Ax = synth.grid_jacobian(n=2500, seed=14)
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(6)
    Bs = [rng.standard_normal(n) for _ in range(6)]
    ax = torch.from_numpy(Ax).to(dev)
    scale = abs(A).sum(axis=0).max()
    with gpu.Factorization(m, n, Ap, Ai) as F:
        bufs = [torch.empty(n, dtype=torch.float64, device=dev) for _ in Bs]
        for x, b in zip(bufs, Bs):
            seen = []
            for _ in range(4):
                x.copy_(torch.from_numpy(b))
                F.factor_solve_dev(ax.data_ptr(), x.data_ptr(), 1, 1e-3, sh)
                torch.cuda.synchronize()
                seen.append(x.cpu().numpy())
            assert np.abs(A @ seen[0] - b).max() <= 1e-13 * (scale * np.abs(seen[0]).max() + np.abs(b).max())
            for other in seen[1:]:
                assert np.array_equal(seen[0], other)
        for x, b in list(zip(bufs, Bs))[::-1]:                        # one call each, in another order: generic graph again
            x.copy_(torch.from_numpy(b))
            F.factor_solve_dev(ax.data_ptr(), x.data_ptr(), 1, 1e-3, sh)
        torch.cuda.synchronize()
        for x, b in zip(bufs, Bs):
            X = x.cpu().numpy()
            assert np.abs(A @ X - b).max() <= 1e-13 * (scale * np.abs(X).max() + np.abs(b).max())


# ------------------------------------------------------------- edge cases ----

def _arrow_with_dense_row(n=400, seed=0):
    """One node coupled to 3/4 of the others: exercises AMD's dense-row deferral (degree > 10 sqrt n)."""
    rng = np.random.default_rng(seed)
    hub = 17
    others = np.setdiff1d(np.arange(n), [hub])
    nb = rng.choice(others, size=3 * n // 4, replace=False)
    ei = np.concatenate([np.full(len(nb), hub), np.arange(n - 1)])
    ej = np.concatenate([nb, np.arange(1, n)])
    ei, ej = synth._unique_edges(n, ei, ej)
    return synth._graph_to_matrix(n, ei, ej, rng)


def _block_diagonal(seed=0):
    """Three disconnected islands plus isolated nodes: a forest with several roots."""
    import scipy.sparse as sp
    blocks = [sp.csc_matrix((b[4], b[3], b[2]), shape=(b[1], b[1]))
              for b in (synth.grid_jacobian(n=150, seed=seed), synth.jacobian_like(nbus=30, nedges=40, n_pv=5, seed=seed + 1),
                        synth.grid_jacobian(n=60, seed=seed + 2))]
    blocks.append(sp.diags(np.arange(2.0, 7.0)).tocsc())
    A = sp.block_diag(blocks, format="csc"); A.sort_indices()
    n = A.shape[0]
    return n, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()


EDGE = {
    "n1": (1, 1, np.array([0, 1], dtype=np.int32), np.array([0], dtype=np.int32), np.array([4.0])),
    "diagonal": (6, 6, np.arange(7, dtype=np.int32), np.arange(6, dtype=np.int32), np.arange(1.0, 7.0)),
    "dense_row": _arrow_with_dense_row(),
    "islands": _block_diagonal(),
}


@pytest.mark.parametrize("name", list(EDGE))
def test_edge_structures(gpu, orc, name):
    m, n, Ap, Ai, Ax = EDGE[name]
    assert np.array_equal(gpu.csc_amd_f(1, m, n, Ap, Ai), orc.csc_amd_f(1, m, n, Ap, Ai))
    Lp, Li, Lx, Up, Ui, Ux, pinv, q = gpu.csc_lu_f(m, n, Ap, Ai, Ax, tol=1e-3)
    oL = orc.csc_lu_f(n, n, Ap, Ai, Ax, q, 1e-3)
    assert np.array_equal(pinv, oL[6])
    assert_factor_equal(n, (Lp, Li, Lx), oL[0:3], name + " L")
    assert_factor_equal(n, (Up, Ui, Ux), oL[3:6], name + " U")
    b = np.arange(1.0, n + 1.0)
    x = gpu.csc_lusol_f(1, m, n, Ap, Ai, Ax, b, tol=1e-3)
    assert rel_err(x, orc.csc_lusol_f(1, n, Ap, Ai, Ax, b, 1e-3)) <= RTOL


def test_unsorted_rows_and_slack_storage(gpu, orc):
    """CscMat may hold unsorted columns (csc_numba.py:334-335) and arrays longer than nnz (csc.py:138)."""
    m, n, Ap, Ai, Ax = CASES["grid2k"]
    rng = np.random.default_rng(5)
    Ai2, Ax2 = Ai.copy(), Ax.copy()
    for j in range(n):
        p = rng.permutation(Ap[j + 1] - Ap[j]) + Ap[j]
        Ai2[Ap[j]:Ap[j + 1]] = Ai[p]; Ax2[Ap[j]:Ap[j + 1]] = Ax[p]
    Ai2 = np.concatenate([Ai2, np.full(50, -7, dtype=np.int32)])       # garbage beyond Ap[n] must be ignored
    Ax2 = np.concatenate([Ax2, np.full(50, np.nan)])
    b = rng.standard_normal(n)
    x = gpu.csc_lusol_f(1, m, n, Ap, Ai2, Ax2, b, tol=1e-3)
    assert rel_err(x, orc.csc_lusol_f(1, n, Ap, Ai, Ax, b, 1e-3)) <= RTOL


def test_entry_with_more_than_64_contributions(gpu, orc):
    """A hub whose 90 leaf neighbours each update its diagonal: the gather's long-run path."""
    n = 120
    rng = np.random.default_rng(2)
    hub = n - 1
    ei = np.concatenate([np.arange(90), np.arange(90, n - 2)])
    ej = np.concatenate([np.full(90, hub), np.arange(91, n - 1)])
    m, n, Ap, Ai, Ax = synth._graph_to_matrix(n, *synth._unique_edges(n, ei, ej), rng)
    q = np.arange(n, dtype=np.int32)                                    # natural order keeps the hub last
    with gpu.Factorization(m, n, Ap, Ai, q=q) as F:
        F.factor(Ax, 1e-3)
        Lp, Li, Lx, Up, Ui, Ux = F.factors()
        qq = F.ordering()["q"]
        x = F.solve(np.ones(n))
    oL = orc.csc_lu_f(n, n, Ap, Ai, Ax, qq, 1e-3)
    assert_factor_equal(n, (Up, Ui, Ux), oL[3:6], "hub U")
    assert np.abs(csc_to_scipy(m, n, Ap, Ai, Ax) @ x - 1.0).max() < 1e-12


def test_call_order_and_argument_errors(gpu):
    m, n, Ap, Ai, Ax = CASES["toy10"]
    with gpu.Factorization(m, n, Ap, Ai) as F:
        with pytest.raises(gpu.Cs3Error):
            F.solve(np.ones(n))                              # solve before factor
        F.factor(Ax)
        with pytest.raises(AssertionError):
            F.solve(np.ones(n + 1))                          # wrong length
        x = F.solve(np.ones((n, 3)))
        assert x.shape == (n, 3)


def test_jacobian_block_stacking_matches_the_reference_kernel(gpu):
    """csc_stack_4_by_4_ff on the device against the reference's own output (first row)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "substrate.npz"))
    g = lambda k: G["st_" + k]
    am, an, bn, cm = int(g("am")), int(g("an")), int(g("bn")), int(g("cm"))
    m, n, Pi, Pp, Px = gpu.csc_stack_4_by_4_ff(am, an, g("Ai"), g("Ap"), g("Ax"), am, bn, g("Bi"), g("Bp"), g("Bx"),
                                               cm, an, g("Ci"), g("Cp"), g("Cx"), cm, bn, g("Di"), g("Dp"), g("Dx"))
    assert (m, n) == (int(g("m")), int(g("n")))
    assert np.array_equal(Pi, g("i")) and np.array_equal(Pp, g("p")) and np.array_equal(Px, g("x"))
    with pytest.raises(AssertionError):
        gpu.csc_stack_4_by_4_ff(am + 1, an, g("Ai"), g("Ap"), g("Ax"), am, bn, g("Bi"), g("Bp"), g("Bx"),
                                cm, an, g("Ci"), g("Cp"), g("Cx"), cm, bn, g("Di"), g("Dp"), g("Dx"))
    # and through the class: pack_4_by_4 then factor + solve the assembled Jacobian
    from csparse3_amd.csc import CscMat, pack_4_by_4
    m0, n0, Ap, Ai, Ax = synth.jacobian_like()
    A = CscMat(m0, n0, indptr=Ap, indices=Ai, data=Ax)
    Z = CscMat(m0, n0, indptr=np.zeros(n0 + 1, dtype=np.int32), indices=np.zeros(0, dtype=np.int32), data=np.zeros(0))
    J = pack_4_by_4(A, Z, Z, A)                      # block diagonal of two Jacobians
    b = np.random.default_rng(0).standard_normal(2 * n0)
    x = J.solve(b, tol=1e-3)
    A0 = csc_to_scipy(m0, n0, Ap, Ai, Ax)
    assert np.abs(A0 @ x[:n0] - b[:n0]).max() < 1e-11 and np.abs(A0 @ x[n0:] - b[n0:]).max() < 1e-11


def test_fresh_handles_reproduce_bitwise_on_big_fronts(gpu):
    """New handle, new graphs, same bits -- including the blocked big-front path and the chunked sweeps."""
    m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=1500, nd=650, seed=5)
    b = np.random.default_rng(0).standard_normal((n, 3))
    ref = None
    for _ in range(4):
        with gpu.Factorization(m, n, Ap, Ai) as F:
            F.factor(Ax, 1e-3)
            got = (F.solve(b), F.factors()[2], F.factors()[5])
        assert not np.isnan(got[0]).any()
        if ref is None:
            ref = got
        assert all(np.array_equal(a, c) for a, c in zip(got, ref))
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    assert np.abs(A @ ref[0] - b).max() <= 1e-11 * np.abs(b).max() * n


@pytest.mark.parametrize("case", ["grid20k_lu", "denseblock_lu", "spd_chol"])
@pytest.mark.parametrize("nrhs", [1, 5, 40])
def test_factor_solve_fused_equals_factor_then_solve(gpu, case, nrhs):
    """cs3_factor_solve_dev (forward sweep overlapped with the factorisation) == factor_dev + solve_dev, bit for bit."""
    import torch
    if case == "grid20k_lu":
        m, n, Ap, Ai, Ax = synth.grid_jacobian(n=20000, seed=20000)
        kind = gpu.CS3_LU
    elif case == "denseblock_lu":
        m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=1500, nd=650, seed=5)
        kind = gpu.CS3_LU
    else:
        ei, ej = synth.spd_grid_pattern(5000, seed=3)
        m, n, Ap, Ai, Ax = synth.spd_grid_matrix(5000, ei, ej, seed=4)
        kind = gpu.CS3_CHOLESKY
    b = np.random.default_rng(nrhs).standard_normal((n, nrhs) if nrhs > 1 else n)
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    d_ax = torch.from_numpy(Ax).to(dev)
    with gpu.Factorization(m, n, Ap, Ai, kind) as F:
        x_split = torch.from_numpy(b).to(dev)
        F.factor_dev(d_ax.data_ptr(), 1e-3, sh)
        F.solve_dev(x_split.data_ptr(), nrhs, sh)
        F.factor_status(sh)
        for _ in range(3):                         # first call captures, later calls replay
            x_fused = torch.from_numpy(b).to(dev)
            F.factor_solve_dev(d_ax.data_ptr(), x_fused.data_ptr(), nrhs, 1e-3, sh)
            F.factor_status(sh)
            assert torch.equal(x_fused, x_split)
        d_b = torch.from_numpy(b).to(dev); x_out = torch.zeros_like(d_b)       # out of place: b untouched
        F.factor_solve_bx_dev(d_ax.data_ptr(), d_b.data_ptr(), x_out.data_ptr(), nrhs, 1e-3, sh)
        F.factor_status(sh)
        assert torch.equal(x_out, x_split) and torch.equal(d_b, torch.from_numpy(b).to(dev))
        Lx_a = F.factors()[2]
    with gpu.Factorization(m, n, Ap, Ai, kind) as G:   # fused as the very first call on a handle
        x0 = torch.from_numpy(b).to(dev)
        G.factor_solve_dev(d_ax.data_ptr(), x0.data_ptr(), nrhs, 1e-3, sh)
        G.factor_status(sh)
        assert torch.equal(x0, x_split)
        assert np.array_equal(G.factors()[2], Lx_a)
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    x = x_split.cpu().numpy()
    assert np.abs(A @ x - b).max() <= 1e-10 * max(1.0, np.abs(b).max()) * np.sqrt(n)


def test_factor_solve_fused_reports_rejected_pivot(gpu):
    import torch
    m, n, Ap, Ai, Ax = synth.jacobian_like()
    Ax = Ax.copy()
    Ax[:] = 0.0
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    with gpu.Factorization(m, n, Ap, Ai) as F:
        x = torch.ones(n, dtype=torch.float64, device=dev)
        F.factor_solve_dev(torch.from_numpy(Ax).to(dev).data_ptr(), x.data_ptr(), 1, 1e-3, sh)
        with pytest.raises(gpu.SingularMatrix):
            F.factor_status(sh)


# ----------------------------------------- many right-hand sides (config 4) --

@pytest.mark.parametrize("name", ["jacobian118", "grid2k", "grid20k", "denseblock300"])
@pytest.mark.parametrize("k", [16, 70, 128])
def test_many_rhs_lu_matches_oracle(gpu, orc, name, k):
    """16 or more right-hand sides take the lane = right-hand-side sweeps on fronts of order <= 32;
    every column must equal the oracle's solve, and the one-column path, to the parity tolerance."""
    m, n, Ap, Ai, Ax = CASES[name]
    B = np.random.default_rng(k).standard_normal((n, k))
    with gpu.Factorization(m, n, Ap, Ai) as F:
        F.factor(Ax, 1e-3)
        Lp, Li, Lx, Up, Ui, Ux = F.factors()
        X = F.solve(B)
        Y = F.lsolve(B)
        Z = F.usolve(Y)
        x7 = F.solve(np.ascontiguousarray(B[:, 7]))
    assert X.shape == (n, k) and not np.isnan(X).any()
    for j in (0, 7, k - 1):
        w = B[:, j].copy(); orc.csc_lsolve_f(n, Lp, Li, Lx, w)
        assert rel_err(Y[:, j], w) <= RTOL
        orc.csc_usolve_f(n, Up, Ui, Ux, w)
        assert rel_err(Z[:, j], w) <= RTOL
    assert rel_err(X[:, 7], x7) <= 1e-12
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    assert np.abs(A @ X - B).max() <= 1e-11 * np.abs(B).max() * n


@pytest.mark.parametrize("name", list(SPD))
@pytest.mark.parametrize("k", [16, 100])
def test_many_rhs_cholesky(gpu, name, k):
    m, n, Ap, Ai, Ax = SPD[name]
    B = np.random.default_rng(k + 1).standard_normal((n, k))
    with gpu.Factorization(m, n, Ap, Ai, kind=gpu.CS3_CHOLESKY) as F:
        F.factor(Ax)
        X = F.solve(B)
        x3 = F.solve(np.ascontiguousarray(B[:, 3]))
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    assert np.abs(A @ X - B).max() <= 1e-11 * np.abs(B).max() * n
    assert rel_err(X[:, 3], x3) <= 1e-12


def test_many_rhs_batch_of_matrices(gpu):
    n = 1500
    ei, ej = synth.spd_grid_pattern(n, seed=77)
    mats = [synth.spd_grid_matrix(n, ei, ej, seed=100 + i) for i in range(3)]
    m, n, Ap, Ai, _ = mats[0]
    AX = np.stack([mm[4] for mm in mats])
    B = np.random.default_rng(5).standard_normal((3, n, 40))
    for kind in (gpu.CS3_LU, gpu.CS3_CHOLESKY):
        with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=3) as F:
            F.factor(AX, 1e-3 if kind == gpu.CS3_LU else 0.0)
            X = F.solve(B)
        for i in range(3):
            A = csc_to_scipy(m, n, Ap, Ai, AX[i])
            assert np.abs(A @ X[i] - B[i]).max() <= 1e-11 * n


@pytest.mark.parametrize("nd", [70, 150, 300, 421])
@pytest.mark.parametrize("nrhs", [1, 9])
@pytest.mark.parametrize("chol", [False, True])
def test_fused_root_pipeline_block_and_chunk_combinations(gpu, nd, nrhs, chol):
    """The fused call releases part of the root's forward sweep while the root is still being factorised; how many
    chunks depends on the root's width (blocks of 32, chunks of 128 or 64 columns).  Every combination must equal
    factor-then-solve bit for bit."""
    import torch
    import scipy.sparse as sp
    m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=nd + 400, nd=nd, seed=nd)
    kind = gpu.CS3_LU
    if chol:
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        S = (A + A.T).tocsc(); S.sort_indices()
        Ap, Ai, Ax = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy()
        kind = gpu.CS3_CHOLESKY
    b = np.random.default_rng(nd + nrhs).standard_normal((n, nrhs) if nrhs > 1 else n)
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    d_ax = torch.from_numpy(Ax).to(dev)
    d_b = torch.from_numpy(b).to(dev)
    with gpu.Factorization(m, n, Ap, Ai, kind) as F:
        assert F.info.max_front >= nd
        x_split = d_b.clone()
        F.factor_dev(d_ax.data_ptr(), 1e-3, sh)
        F.solve_dev(x_split.data_ptr(), nrhs, sh)
        F.factor_status(sh)
        for _ in range(2):
            x_fused = torch.zeros_like(d_b)
            F.factor_solve_bx_dev(d_ax.data_ptr(), d_b.data_ptr(), x_fused.data_ptr(), nrhs, 1e-3, sh)
            F.factor_status(sh)
            assert torch.equal(x_fused, x_split)
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    x = x_split.cpu().numpy()
    assert np.abs(A @ x - b).max() <= 1e-10 * max(1.0, np.abs(b).max()) * n


# ----------------------------------------------- residual and iterative refinement ----

def test_matvec_on_resident_data_is_bit_exact_with_the_host_entry_point(gpu):
    """cs3_matvec_dev (the handle's pattern, resident values and vectors) against cs3_csc_matvec: the same sums, bit for bit."""
    import torch
    m, n, Ap, Ai, Ax = synth.grid_jacobian(n=4000, seed=21)
    rng = np.random.default_rng(2)
    X = rng.standard_normal((n, 3))
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    with gpu.Factorization(m, n, Ap, Ai) as F:
        d_ax, d_x = torch.from_numpy(Ax.copy()).to(dev), torch.from_numpy(X).to(dev)
        d_y = torch.empty_like(d_x)
        F.matvec_dev(d_ax.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), 3, sh)
        torch.cuda.synchronize()
        Y = d_y.cpu().numpy()
    for t in range(3):
        assert np.array_equal(Y[:, t], gpu.csc_mat_vec_ff(m, n, Ap, Ai, Ax, np.ascontiguousarray(X[:, t])))


def test_residual_and_refinement_on_resident_data(gpu):
    """cs3_residual_dev reproduces b - csc_mat_vec_ff(A, x) bit for bit (golden matvec outputs of the reference);
    cs3_refine_dev with the factors of a NEARBY matrix (a stale Newton iterate) drives the residual of the current
    system down to rounding in a few rounds."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "substrate.npz"))
    # (i) residual == b - (reference matvec), square golden case r1 (40 x 40)
    m, n = int(G["r1_m"]), int(G["r1_n"])
    Ap, Ai, Ax, x = G["r1_Ap"], G["r1_Ai"], G["r1_Ax"], G["r1_x"]
    import scipy.sparse as sp
    b = np.random.default_rng(0).standard_normal(n)
    with gpu.Factorization(m, n, Ap, Ai, order=gpu.ORDER_NATURAL) as F:
        r = torch.empty(n, dtype=torch.float64, device=dev)
        d_ax, d_b, d_x = T(Ax), T(b), T(x)                              # (named: a temporary would be freed before the launch)
        F.residual_dev(d_ax.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), r.data_ptr(), 1, sh)
        assert np.array_equal(r.cpu().numpy(), b - G["r1_matvec"])
    # (ii) refinement: factors of A0, system with A1 = A0 (1 + 1e-3 noise)
    m, n, Ap, Ai, Ax0 = synth.grid_jacobian(n=20000, seed=11)
    rng = np.random.default_rng(5)
    Ax1 = Ax0 * (1.0 + 1e-3 * rng.uniform(-1.0, 1.0, size=Ax0.shape))
    B = rng.standard_normal((n, 3))
    A1 = csc_to_scipy(m, n, Ap, Ai, Ax1)
    with gpu.Factorization(m, n, Ap, Ai) as F:
        F.factor(Ax0, 1e-3)
        d_ax1, d_b = T(Ax1), T(B)
        d_x = d_b.clone()
        F.solve_dev(d_x.data_ptr(), 3, sh)                              # x0 = A0 \\ b: wrong by ~1e-3 for A1
        res = [np.abs(A1 @ d_x.cpu().numpy() - B).max()]
        for _ in range(4):
            corr = F.refine_dev(d_ax1.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), 3, 1, sh)
            res.append(np.abs(A1 @ d_x.cpu().numpy() - B).max())
        assert res[0] > 1e-5 and res[-1] < 1e-12 and all(b_ < a_ or b_ < 1e-12 for a_, b_ in zip(res, res[1:])), res
        assert corr < 1e-9
        # the device residual agrees with the host one
        r = torch.empty_like(d_b)
        F.residual_dev(d_ax1.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), r.data_ptr(), 3, sh)
        assert np.abs(r.cpu().numpy() - (B - A1 @ d_x.cpu().numpy())).max() < 1e-12


def test_distinct_patterns_one_matrix_per_stream(gpu, orc):
    """BASELINE configs[4], 'one matrix per GPU stream': matrices whose PATTERNS differ each get a handle and run
    concurrently on a pool of streams; every solution equals the same matrix solved alone on the default stream,
    bit for bit, and the oracle's to 1e-10."""
    import torch
    from csparse3_amd.streams import DistinctBatch
    dev = torch.device("cuda", 0)
    mats = []
    for i in range(12):
        n = 1500 + 37 * i
        ei, ej = synth.spd_grid_pattern(n, seed=900 + i, chord_frac=0.01 + 0.002 * i)          # different sizes and chords
        mats.append(synth.spd_grid_matrix(n, ei, ej, seed=950 + i))
    rng = np.random.default_rng(12)
    B = [rng.standard_normal(mm[1]) for mm in mats]
    with DistinctBatch([(mm[0], mm[1], mm[2], mm[3]) for mm in mats], kind=gpu.CS3_CHOLESKY, nstreams=4) as D:
        vals = [torch.from_numpy(mm[4]).to(dev) for mm in mats]
        for rep in range(3):                                           # first call captures the graphs, later calls replay
            rhs = [torch.from_numpy(b.copy()).to(dev) for b in B]
            D.factor_solve(vals, rhs)
            D.status()
        X = [r.cpu().numpy() for r in rhs]
    for i, mm in enumerate(mats):
        m, n, Ap, Ai, Ax = mm
        with gpu.Factorization(m, n, Ap, Ai, kind=gpu.CS3_CHOLESKY) as F:
            want = F.factor(Ax).solve(B[i])
        assert rel_err(X[i], want) <= 1e-13, i                         # (fused call vs factor-then-solve: same kernels)
        A = csc_to_scipy(m, n, Ap, Ai, Ax)
        assert np.abs(A @ X[i] - B[i]).max() <= 1e-11 * n
    assert rel_err(X[5], orc.csc_cholsol_f(1, mats[5][1], mats[5][2], mats[5][3], mats[5][4], B[5])) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("diag_scale", [0.1, 0.005])
def test_inverse_based_sweeps_hold_up_with_large_multipliers(gpu, diag_scale):
    """With 16 or more right-hand sides the fronts of order > 32 are swept with explicit inverses of their 64 x 64 diagonal
    blocks (k_inv_diag / k_gemm_fwd / k_gemm_bwd), which is not backward stable in general; every other many-RHS test is
    diagonally dominant (multipliers << 1).  Here the diagonal is weak: multipliers reach ~200 (tol = 1e-3 admits 1000) and
    the residual of the substitution path itself grows to 1e-15 .. 1e-12.  The same b goes through the one-RHS substitution
    sweeps and as the columns of a 32-column block through the inverse-based ones: the block's residual must stay within a
    small factor of the substitution path's, column for column.  (kernels.hip:2722; measured equal.)"""
    import scipy.sparse as sp
    hip = gpu
    m, n, Ap, Ai, Ax = synth.dense_block_matrix(500, 260, seed=5)
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).tolil()
    A.setdiag(A.diagonal() * diag_scale)
    A = A.tocsc(); A.sort_indices()
    Ap2, Ai2, Ax2 = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, 32))
    with hip.Factorization(n, n, Ap2, Ai2) as F:
        F.factor(Ax2, 1e-3)
        Lx = F.factors()[2]
        assert np.abs(Lx).max() > 50.0, "the case is meant to have large multipliers"
        assert F.info.max_front > 64
        X1 = np.stack([F.solve(B[:, j].copy()) for j in range(4)], axis=1)      # substitution sweeps, one column at a time
        X = F.solve(B.copy())                                                   # inverse-based GEMM sweeps
    scale = np.abs(A).max()
    res = lambda x, b: np.abs(A @ x - b).max() / (scale * np.abs(x).max() + np.abs(b).max())
    for j in range(4):
        r1, rk = res(X1[:, j], B[:, j]), res(X[:, j], B[:, j])
        assert rk <= 8.0 * r1 + 1e-14, (j, r1, rk)
        assert np.abs(X[:, j] - X1[:, j]).max() <= 1e-9 * np.abs(X1[:, j]).max()
    for j in range(4, 32):
        assert res(X[:, j], B[:, j]) <= 1e-10


@pytest.mark.parametrize("chol", [False, True])
@pytest.mark.parametrize("nd", [20, 33, 40, 48, 64, 65, 81, 100, 136, 137, 180])
@pytest.mark.parametrize("nb", [1, 20])
def test_front_orders_across_the_kernel_classes(gpu, orc, chol, nd, nb):
    """One embedded dense block of every order around the boundaries of the factor kernels -- one-wave / four-wave fronts
    with the panel + MFMA Schur complement (<= 32 pivots), the 16 x 16 thread grid (33..64 pivots), the LDS-resident block
    kernel with its equal-width blocks (65..136), the blocked big-front path (> 136) -- single and batched (the batched
    handle takes the one-wave kernel, two-wave fronts and, for Cholesky, the packed images): L and U against the oracle
    entry for entry, residuals, and run-to-run bits."""
    import scipy.sparse as sp
    m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=nd + 120, nd=nd, seed=1000 + nd)
    kind = gpu.CS3_LU
    if chol:
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        S = (A + A.T).tocsc(); S.sort_indices()
        Ap, Ai, Ax = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy()
        kind = gpu.CS3_CHOLESKY
    rng = np.random.default_rng(nd + nb)
    AX = Ax[None, :] * (1.0 + rng.uniform(0.0, 0.5, size=(nb, 1)))
    B = rng.standard_normal((nb, n, 3))
    tol = 0.0 if chol else 1e-3
    with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=nb) as F:
        assert F.info.max_front >= nd
        F.factor(AX if nb > 1 else AX[0], tol)
        X = F.solve(B if nb > 1 else B[0])
        q = F.ordering()["q"]
        picks = sorted({0, nb - 1})
        facs = {i: (F.factors(b=i) if nb > 1 else F.factors()) for i in picks}
        F.factor(AX if nb > 1 else AX[0], tol)
        assert np.array_equal(F.solve(B if nb > 1 else B[0]), X)
    X = X if nb > 1 else X[None]
    for i, (Lp, Li, Lx, Up, Ui, Ux) in facs.items():
        what = "order %d, %s, matrix %d of %d" % (nd, "chol" if chol else "lu", i, nb)
        if chol:
            assert_factor_equal(n, (Lp, Li, Lx), _oracle_chol(orc, n, Ap, Ai, AX[i], q), what)
        else:
            oL = orc.csc_lu_f(n, n, Ap, Ai, AX[i], q, 1e-3)
            assert_factor_equal(n, (Lp, Li, Lx), oL[0:3], what + " L")
            assert_factor_equal(n, (Up, Ui, Ux), oL[3:6], what + " U")
    for i in range(nb):
        A = csc_to_scipy(m, n, Ap, Ai, AX[i])
        assert np.abs(A @ X[i] - B[i]).max() <= 1e-11 * n * max(1.0, np.abs(X[i]).max())
