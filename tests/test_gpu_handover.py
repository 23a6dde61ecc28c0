"""The hand-overs between waves inside a kernel must fail loudly (ADVICE: kernels.hip:411).

Waves that share an elimination (eliminate_pair / eliminate_parts in kernels.hip, the shared fronts of forest.hip) pass
multipliers through LDS and poll a counter.  The poll is bounded; a wave that gives up used to carry on with whatever the
buffer held and the step passed as a success.  Now it raises a status word and cs3_factor_status reports CS3_ERR_STATE.
cs3_debug_withhold_handover(1) makes every producer keep its counter back, so every consumer gives up: the test of that
path.  (Not for runs under tools that change kernel timing: it only shortens the bound, nothing waits for real time.)"""
import ctypes as C

import numpy as np
import pytest

from csparse3_amd import synth

pytestmark = pytest.mark.gpu


def _withhold(hip, on):
    lib = hip.lib()
    lib.cs3_debug_withhold_handover.argtypes = [C.c_int]
    assert lib.cs3_debug_withhold_handover(int(on)) == 0


@pytest.mark.parametrize("case", ["grid_lu", "grid_cholesky", "dense_root"])
def test_a_withheld_hand_over_is_reported_not_swallowed(gpu, case):
    hip = gpu
    if case == "grid_lu":                       # shared fronts of the forest + the two-wave eliminations of k_front_mix
        m, n, Ap, Ai, Ax = synth.grid_jacobian(4000, seed=7)
        kind = hip.CS3_LU
    elif case == "grid_cholesky":
        n = 3000
        ei, ej = synth.spd_grid_pattern(n, seed=11)
        m, n, Ap, Ai, Ax = synth.spd_grid_matrix(n, ei, ej, seed=11)
        kind = hip.CS3_CHOLESKY
    else:                                       # a dense root beyond the LDS: the stacked eliminations of k_big_step
        m, n, Ap, Ai, Ax = synth.dense_block_matrix(400, 200, seed=3)
        kind = hip.CS3_LU
    b = np.random.default_rng(1).standard_normal(n)
    with hip.Factorization(m, n, Ap, Ai, kind=kind) as F:
        x_ok = F.factor(Ax, 1e-3 if kind == hip.CS3_LU else 0.0).solve(b)
        try:
            _withhold(hip, True)
            with pytest.raises(hip.Cs3Error) as err:
                F.factor(Ax, 1e-3 if kind == hip.CS3_LU else 0.0)
            assert err.value.code == hip.CS3_ERR_STATE, err.value
            with pytest.raises(hip.Cs3Error):       # ... and the handle refuses to solve with those factors
                F.solve(b)
        finally:
            _withhold(hip, False)
        x_again = F.factor(Ax, 1e-3 if kind == hip.CS3_LU else 0.0).solve(b)     # the handle is usable again
        assert np.array_equal(x_again, x_ok)
