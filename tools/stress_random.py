#!/usr/bin/env python3
"""Randomised stress of the numeric path (not a test: run by hand on a GPU box).
Random embedded dense blocks (root widths across the block/chunk boundaries), right-hand-side counts, LU and
Cholesky, batches; checks factor-then-solve against the fused calls bit for bit, and residuals."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scipy.sparse as sp
from csparse3_amd import synth, csc_hip as hip

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0); sh = torch.cuda.current_stream().cuda_stream
worst = 0.0
for case in range(ncase):
    nd = int(rng.choice([0, 40, 65, 97, 129, 160, 200, 257, 330, 449, 520]))
    n = nd + int(rng.integers(50, 900))
    nrhs = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 33, 64, 70, 128, 256, 300]))
    chol = bool(rng.integers(0, 2)); batch = int(rng.choice([1, 1, 1, 3, 16, 64, 70, 130]))     # (130: the lane = matrix kernels, from 128 matrices on)
    if batch >= 16: nrhs = min(nrhs, 70)          # (memory: batch x n x nrhs doubles, three copies)
    if nd:
        m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=n, nd=nd, seed=int(rng.integers(1 << 30)))
    else:
        m, n, Ap, Ai, Ax = synth.grid_jacobian(n=n, seed=int(rng.integers(1 << 30)))
    if chol:
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)); S = (A + A.T).tocsc(); S.sort_indices()
        Ap, Ai, Ax = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy()
    AX = np.stack([Ax * (1.0 + 0.01 * i) for i in range(batch)])
    B = rng.standard_normal((batch, n, nrhs))
    d_ax = torch.from_numpy(AX).to(dev); d_b = torch.from_numpy(B).to(dev)
    with hip.Factorization(m, n, Ap, Ai, hip.CS3_CHOLESKY if chol else hip.CS3_LU, batch=batch) as F:
        xs = d_b.clone()
        F.factor_dev(d_ax.data_ptr(), 1e-3, sh); F.solve_dev(xs.data_ptr(), nrhs, sh); F.factor_status(sh)
        xf = torch.zeros_like(d_b)
        F.factor_solve_bx_dev(d_ax.data_ptr(), d_b.data_ptr(), xf.data_ptr(), nrhs, 1e-3, sh); F.factor_status(sh)
        xi = d_b.clone()
        F.factor_solve_dev(d_ax.data_ptr(), xi.data_ptr(), nrhs, 1e-3, sh); F.factor_status(sh)
        assert torch.equal(xs, xf) and torch.equal(xs, xi), ("fused differs", case, nd, n, nrhs, chol, batch)
        X = xs.cpu().numpy()
        assert not np.isnan(X).any(), ("nan", case, nd, n, nrhs, chol, batch)
        for i in range(batch):
            A = sp.csc_matrix((AX[i], Ai, Ap), shape=(n, n))
            res = np.abs(A @ X[i] - B[i]).max() / (np.abs(B[i]).max() * n)
            worst = max(worst, res)
            assert res < 1e-12, ("residual", res, case, nd, n, nrhs, chol, batch)
    if case % 10 == 9:
        print("case", case + 1, "ok, worst scaled residual so far %.2e" % worst, flush=True)
print("stress ok:", ncase, "cases")
