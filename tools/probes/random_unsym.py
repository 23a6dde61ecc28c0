"""Robustness probe: random structurally UNSYMMETRIC matrices (diagonally dominant), LU with 1 / 40 / 300 right-hand sides,
fused refactor + solve, against SciPy's matvec and SuperLU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from csparse3_amd import csc_hip as hip

rng = np.random.default_rng(5)
for n, dens in ((500, 0.01), (3000, 0.002), (20000, 0.0002), (60000, 0.00005)):
    A = sp.random(n, n, density=dens, random_state=int(rng.integers(1 << 30)), format="csc", data_rvs=rng.standard_normal)
    A = (A + sp.diags(np.abs(A).sum(axis=0).A1 + np.abs(A).sum(axis=1).A1 + 1.0)).tocsc()
    A.sum_duplicates(); A.sort_indices()
    Ap, Ai, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    sym = (abs(A) > 0).astype(int); asym = (sym - sym.T).nnz
    try:
        F = hip.Factorization(n, n, Ap, Ai)
    except hip.Cs3Error as e:                      # random patterns fill in catastrophically: the pool is limited to 2^30 doubles
        print("n=%6d nnz=%8d: refused -- %s" % (n, A.nnz, e)); continue
    with F:
        inf = F.info
        F.factor(Ax, 1e-3)
        worst = 0.0
        for k in (1, 40, 300):
            B = rng.standard_normal((n, k))
            X = F.solve(B)
            worst = max(worst, np.abs(A @ X - B).max() / (abs(A).sum(axis=0).max() * np.abs(X).max() + np.abs(B).max()))
        x1 = F.solve(B[:, 0].copy())
    ref = spl.splu(A).solve(B[:, 0])
    print("n=%6d nnz=%8d unsymmetric entries=%7d nnz(L)=%9d max_front=%5d levels=%3d  worst residual %.2e  vs SuperLU %.2e"
          % (n, A.nnz, asym, inf.nnz_l, inf.max_front, inf.nlevels, worst, np.abs(x1 - ref).max() / np.abs(ref).max()), flush=True)
    assert worst < 1e-13
print("random unsymmetric ok")
