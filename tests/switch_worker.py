"""Worker of tests/test_gpu_switches.py: one pass over the numeric path under the environment it was started with
(the library reads its switches once per process).  Prints the worst scaled residual."""
import sys
import numpy as np
import scipy.sparse as sp
import torch
from csparse3_amd import csc_hip as hip, synth

worst = 0.0


def check(A, X, B):
    global worst
    r = np.abs(A @ X - B).max() / (abs(A).sum(axis=0).max() * np.abs(X).max() + np.abs(B).max())
    worst = max(worst, float(r))


rng = np.random.default_rng(3)
dev = torch.device("cuda", 0)
sh = torch.cuda.current_stream().cuda_stream
# single matrix with a dense root on the blocked path: fused step, 1 and 300 right-hand sides
m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=900, nd=300, seed=5)
A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
with hip.Factorization(m, n, Ap, Ai) as F:
    for k in (1, 300):
        B = rng.standard_normal((n, k))
        d_ax, d_b = torch.from_numpy(Ax.copy()).to(dev), torch.from_numpy(B).to(dev)
        d_x = torch.empty_like(d_b)
        F.factor_solve_bx_dev(d_ax.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), k, 1e-3, sh)
        F.factor_status(sh)
        check(A, d_x.cpu().numpy(), B)
# interleaved batch of SPD matrices (70 = two groups), Cholesky, with a front beyond the LDS
m, n, Ap, Ai, Ax = synth.dense_block_matrix(n=700, nd=180, seed=6)
S = (sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)) + sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).T).tocsc(); S.sort_indices()
Sp, Si, Sx = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy()
nb = 70
AX = Sx[None, :] * (1.0 + rng.uniform(0.0, 1.0, size=(nb, 1)))
B = rng.standard_normal((nb, n, 2))
with hip.Factorization(m, n, Sp, Si, kind=hip.CS3_CHOLESKY, batch=nb) as F:
    F.factor(AX)
    X = F.solve(B)
for i in (0, 63, 64, 69):
    check(sp.csc_matrix((AX[i], Si, Sp), shape=(n, n)), X[i], B[i])
print("worst %.3e" % worst)
sys.exit(0 if worst < 1e-12 else 1)
