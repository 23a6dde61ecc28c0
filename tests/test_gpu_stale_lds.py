"""The LDS keeps its contents between kernels.  A kernel that multiplies a masked-to-zero operand by an LDS word nobody
wrote is right as long as the stale word is finite and wrong (NaN) when it is not -- it happened once in round 2
(k_fwd_blk / k_bwd_blk).  Here every CU's LDS is filled with NaN patterns (cs3_debug_poison_lds)
before each numeric entry point: factorisation, full solves with 1 / 5 / 40 / 300 right-hand sides, stand-alone
lsolve / usolve, the fused step; LU and Cholesky, a single matrix with a blocked root, and batches on both sides of the
lane = matrix threshold."""
import ctypes as C

import numpy as np
import pytest

from csparse3_amd import synth
from tests.test_gpu_parity import csc_to_scipy


def _poison(gpu):
    import torch
    lib = gpu.lib()
    lib.cs3_debug_poison_lds.argtypes = [C.c_void_p]
    assert lib.cs3_debug_poison_lds(C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0


def _cases():
    yield ("grid6000_lu", 0) + tuple(synth.grid_jacobian(n=6000, seed=21))      # blocked root, every front class
    yield ("jac118_lu", 0) + tuple(synth.jacobian_config2()[:5])
    ei, ej = synth.spd_grid_pattern(3000, seed=31)
    yield ("spd3000_chol", 1) + tuple(synth.spd_grid_matrix(3000, ei, ej, seed=32))
    yield ("denseblock_lu", 0) + tuple(synth.dense_block_matrix(n=700, nd=300, seed=4))


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_numeric_entry_points_with_poisoned_lds(gpu, case):
    import torch
    name, kind, m, n, Ap, Ai, Ax = case
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    scale = abs(A).sum(axis=0).max()
    tol = 1e-3 if kind == 0 else 0.0
    rng = np.random.default_rng(3)
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream

    def check(x, b, what):
        assert np.isfinite(x).all(), (name, what)
        assert np.abs(A @ x - b).max() <= 1e-12 * (scale * np.abs(x).max() + np.abs(b).max()), (name, what)

    with gpu.Factorization(m, n, Ap, Ai, kind=kind) as F:
        _poison(gpu)
        F.factor(Ax, tol)
        for k in (1, 5, 40, 300):
            b = rng.standard_normal((n, k)) if k > 1 else rng.standard_normal(n)
            _poison(gpu)
            check(F.solve(b), b, k)
        q = F.ordering()["q"]
        b = rng.standard_normal(n)
        _poison(gpu)
        y = F.lsolve(b[q])
        _poison(gpu)
        z = F.usolve(y)
        x = np.empty(n)
        x[q] = z
        check(x, b, "lsolve+usolve")
        ax = torch.from_numpy(np.ascontiguousarray(Ax)).to(dev)
        for k in (1, 20):
            b = rng.standard_normal((n, k)) if k > 1 else rng.standard_normal(n)
            xd = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
            for _ in range(3):                                       # third call: the graph with the permutation inside
                xd.copy_(torch.from_numpy(np.ascontiguousarray(b)))
                _poison(gpu)
                F.factor_solve_dev(ax.data_ptr(), xd.data_ptr(), k, tol, sh)
            F.factor_status(sh)
            check(xd.cpu().numpy(), b, ("fused", k))


@pytest.mark.gpu
@pytest.mark.parametrize("nmat", [6, 70, 130])      # (130: the lane = matrix kernels, from 128 matrices on)
@pytest.mark.parametrize("chol", [False, True])
def test_batches_with_poisoned_lds(gpu, nmat, chol):
    n = 1200
    ei, ej = synth.spd_grid_pattern(n, seed=41)
    mats = [synth.spd_grid_matrix(n, ei, ej, seed=200 + i) for i in range(nmat)]
    m, n, Ap, Ai, _ = mats[0]
    AX = np.stack([mm[4] for mm in mats])
    B = np.random.default_rng(9).standard_normal((nmat, n, 2))
    kind = gpu.CS3_CHOLESKY if chol else gpu.CS3_LU
    with gpu.Factorization(m, n, Ap, Ai, kind=kind, batch=nmat) as F:
        _poison(gpu)
        F.factor(AX, 0.0 if chol else 1e-3)
        _poison(gpu)
        X = F.solve(B)
    assert np.isfinite(X).all()
    for i in (0, nmat // 2, nmat - 1):
        A = csc_to_scipy(m, n, Ap, Ai, AX[i])
        assert np.abs(A @ X[i] - B[i]).max() <= 1e-11 * (abs(A).sum(axis=0).max() * np.abs(X[i]).max() + np.abs(B[i]).max())
