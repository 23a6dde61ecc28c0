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
