"""Oracle-INDEPENDENT known answers for BASELINE configs 1 and 2 and two larger cases (tests/golden/factor_fixtures.npz).

The fixtures hold A, b, the pivot order and dense L, U, Cholesky L and x computed in extended precision by
tests/golden/make_factor_fixtures.py (cross-checked there against LAPACK and SuperLU).  Nothing here goes
through oracle/liboracle.so except the one test that pins the ORACLE to the same answers:

  * CPU suite: the library's host-side analysis reproduces the stored integer outputs; the oracle's cs_lu /
    cs_chol / solves reproduce the stored factors and solutions (this is the oracle's pin for the hot path);
  * GPU suite (`-m gpu`): the HIP path, through the C ABI, against the stored factors and solutions directly.
"""
import os

import numpy as np
import pytest

from helpers import RTOL

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "factor_fixtures.npz"))
TAGS = ["toy10", "jac118", "config2", "grid1500", "block500"]    # the last two: fronts beyond the one-wave kernels, a dense 180-pivot root


def _get(tag):
    g = lambda k: FIX[tag + "_" + k]
    n = int(g("n"))
    return n, g


def _dense(n, Gp, Gi, Gx):
    D = np.zeros((n, n))
    for j in range(n):
        D[Gi[Gp[j]:Gp[j + 1]], j] = Gx[Gp[j]:Gp[j + 1]]
    return D


def _pattern(n, Gp, Gi):
    P = np.zeros((n, n), dtype=bool)
    for j in range(n):
        P[Gi[Gp[j]:Gp[j + 1]], j] = True
    return P


def _check_factors(n, g, Lp, Li, Lx, Up, Ui, Ux):
    pat = np.unpackbits(g("pattern"))[:n * n].reshape(n, n).astype(bool)
    assert np.array_equal(_pattern(n, Lp, Li), np.tril(pat)), "pattern of L differs from the boolean elimination"
    assert np.array_equal(_pattern(n, Up, Ui), np.triu(pat)), "pattern of U differs from the boolean elimination"
    L, U = g("L"), g("U")
    eL = np.abs(_dense(n, Lp, Li, Lx) - L).max() / np.abs(L).max()
    eU = np.abs(_dense(n, Up, Ui, Ux) - U).max() / np.abs(U).max()
    assert eL <= RTOL and eU <= RTOL, (eL, eU)
    return eL, eU


@pytest.mark.parametrize("tag", TAGS)
def test_host_analysis_reproduces_the_fixture(hip, tag):
    n, g = _get(tag)
    with hip.Factorization(n, n, g("Ap"), g("Ai")) as F:
        o = F.ordering()
    for k in ("q_amd", "q", "pinv", "parent", "post", "colcount"):
        assert np.array_equal(o[k], g(k)), k
    with hip.Factorization(n, n, g("S_Ap"), g("S_Ai"), kind=hip.CS3_CHOLESKY) as G:
        assert np.array_equal(G.ordering()["q"], g("S_q"))


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_is_pinned_by_the_fixture(orc, tag):
    """The CPU oracle (cs_lu with the stored pivot order, cs_lsolve / cs_usolve, cs_chol) against the
    extended-precision dense answers: its parity anchor for the hot path."""
    n, g = _get(tag)
    Ap, Ai, Ax, q, b = g("Ap"), g("Ai"), g("Ax"), g("q"), g("b")
    Lp, Li, Lx, Up, Ui, Ux, pinv = orc.csc_lu_f(n, n, Ap, Ai, Ax, q, 1e-3)
    assert np.array_equal(pinv, g("pinv")), "the oracle left the diagonal"
    _check_factors(n, g, Lp, Li, Lx, Up, Ui, Ux)
    x = np.empty(n); x[pinv] = b
    orc.csc_lsolve_f(n, Lp, Li, Lx, x)
    orc.csc_usolve_f(n, Up, Ui, Ux, x)
    sol = np.empty(n); sol[q] = x
    assert np.abs(sol - g("x")).max() <= RTOL * np.abs(g("x")).max()
    # Cholesky
    Sp, Si, Sx, qc = g("S_Ap"), g("S_Ai"), g("S_Ax"), g("S_q")
    pinv_c = orc.csc_pinv(qc)
    _, _, Cp, Ci, _ = orc.csc_symperm(n, Sp, Si, None, pinv_c)
    parent = orc.csc_etree_f(n, Cp, Ci)
    cnt = orc.csc_counts_f(n, Cp, Ci, parent, orc.csc_post_f(n, parent))
    cp = np.zeros(n + 1, dtype=np.int32); cp[1:] = np.cumsum(cnt)
    Lcp, Lci, Lcx = orc.csc_chol_f(n, Sp, Si, Sx, pinv_c, parent, cp)
    Lc = g("S_L")
    assert np.abs(_dense(n, Lcp, Lci, Lcx) - Lc).max() <= RTOL * np.abs(Lc).max()


def test_toy10_known_answer(orc):
    """Config 1 as worded in SURVEY.md section 8d: b = A (1..10), so x = 1..10."""
    n, g = _get("toy10")
    x = orc.csc_lusol_f(1, n, g("Ap"), g("Ai"), g("Ax"), FIX["toy10_b_known"], 1e-3)
    assert np.abs(x - FIX["toy10_x_known"]).max() <= 1e-13 * 10


# ------------------------------------------------------------------ GPU: HIP path vs the fixture, no oracle ----

@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_lu_and_solve_match_the_fixture(gpu, tag):
    n, g = _get(tag)
    Ap, Ai, Ax, b = g("Ap"), g("Ai"), g("Ax"), g("b")
    with gpu.Factorization(n, n, Ap, Ai) as F:
        assert np.array_equal(F.ordering()["q"], g("q"))
        F.factor(Ax, 1e-3)
        Lp, Li, Lx, Up, Ui, Ux = F.factors()
        x = F.solve(b)
        y = F.lsolve(b[g("q")])                                  # L y = P b in pivot order
        z = F.usolve(y)
    _check_factors(n, g, Lp, Li, Lx, Up, Ui, Ux)
    assert np.all(Lx[Lp[:-1]] == 1.0) and np.array_equal(Ui[Up[1:] - 1], np.arange(n))     # cs_lu's layout
    want = g("x")
    assert np.abs(x - want).max() <= RTOL * np.abs(want).max()
    assert np.abs(z - want[g("q")]).max() <= RTOL * np.abs(want).max()
    # fused refactor + solve on resident data gives the same answer
    import torch
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    with gpu.Factorization(n, n, Ap, Ai) as F:
        d_ax, d_b = torch.from_numpy(Ax.copy()).to(dev), torch.from_numpy(b.copy()).to(dev)
        d_x = torch.empty_like(d_b)
        F.factor_solve_bx_dev(d_ax.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), 1, 1e-3, sh)
        F.factor_status(sh)
        assert np.abs(d_x.cpu().numpy() - want).max() <= RTOL * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_cholesky_matches_the_fixture(gpu, tag):
    n, g = _get(tag)
    Sp, Si, Sx = g("S_Ap"), g("S_Ai"), g("S_Ax")
    with gpu.Factorization(n, n, Sp, Si, kind=gpu.CS3_CHOLESKY) as F:
        assert np.array_equal(F.ordering()["q"], g("S_q"))
        F.factor(Sx)
        Lp, Li, Lx, _, _, _ = F.factors()
    Lc = g("S_L")
    assert np.abs(_dense(n, Lp, Li, Lx) - Lc).max() <= RTOL * np.abs(Lc).max()
    assert not (~_pattern(n, Lp, Li) & (Lc != 0)).any(), "a non-zero of the dense factor lies outside the pattern"


@pytest.mark.gpu
def test_hip_toy10_known_answer(gpu):
    n, g = _get("toy10")
    x = gpu.csc_lusol_f(1, n, n, g("Ap"), g("Ai"), g("Ax"), FIX["toy10_b_known"], tol=1e-3)
    assert np.abs(x - FIX["toy10_x_known"]).max() <= 1e-13 * 10


@pytest.mark.gpu
@pytest.mark.parametrize("nb", [70, 130])
def test_hip_batch_of_the_fixture_matrix(gpu, nb):
    """Config 2 as a batch of 70 / 130 scaled copies (more than one 64-matrix group; 130: the lane = matrix kernels, which
    take the small fronts from 128 matrices on): every factor and solution is the scaled fixture answer."""
    n, g = _get("config2")
    Ap, Ai, Ax, b = g("Ap"), g("Ai"), g("Ax"), g("b")
    scale = 1.0 + np.arange(nb) / 7.0
    AX = Ax[None, :] * scale[:, None]
    B = np.repeat(b[None, :, None], nb, axis=0)
    with gpu.Factorization(n, n, Ap, Ai, batch=nb) as F:
        F.factor(AX, 1e-3)
        X = F.solve(B)
        for i in (0, 1, 63, 64, 69):
            Lp, Li, Lx, Up, Ui, Ux = F.factors(b=i)
            U = g("U") * scale[i]
            assert np.abs(_dense(n, Up, Ui, Ux) - U).max() <= RTOL * np.abs(U).max()
            assert np.abs(_dense(n, Lp, Li, Lx) - g("L")).max() <= RTOL * np.abs(g("L")).max()
    want = g("x")[None, :] / scale[:, None]
    assert np.abs(X[:, :, 0] - want).max() <= RTOL * np.abs(want).max()
