"""Robustness probe on matrix families outside synth.py: 2-D and 3-D Laplacians (large dense root fronts), LU and
Cholesky, 1 / 20 / 300 right-hand sides, a batch; residuals against SciPy's matvec."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from csparse3_amd import csc_hip as hip

def lap(dims):
    mats = [sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(d, d)) for d in dims]
    A = None
    for i, m in enumerate(mats):
        term = None
        for j, d in enumerate(dims):
            f = m if j == i else sp.identity(d)
            term = f if term is None else sp.kron(term, f)
        A = term if A is None else A + term
    A = (A + 0.01 * sp.identity(A.shape[0])).tocsc(); A.sort_indices()
    return A

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for name, dims in (("2d_200x200", (200, 200)), ("3d_24", (24, 24, 24)), ("3d_30", (30, 30, 30))):
        A = lap(dims); n = A.shape[0]
        Ap, Ai, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
        for kind, kname in ((hip.CS3_CHOLESKY, "chol"), (hip.CS3_LU, "lu")):
            t0 = time.time()
            with hip.Factorization(n, n, Ap, Ai, kind=kind) as F:
                inf = F.info
                F.factor(Ax, 1e-3)
                worst = 0.0
                for k in (1, 20, 300):
                    B = rng.standard_normal((n, k))
                    X = F.solve(B)
                    worst = max(worst, np.abs(A @ X - B).max() / (abs(A).sum(axis=0).max() * np.abs(X).max() + np.abs(B).max()))
            print("%-12s %-5s n=%6d nnz(L)=%9d max_front=%5d levels=%3d  worst scaled residual %.2e  (%.1f s)"
                  % (name, kname, n, inf.nnz_l, inf.max_front, inf.nlevels, worst, time.time() - t0), flush=True)
            assert worst < 1e-13
    # a small batch of 3-D Laplacians with different shifts (Cholesky, interleaved: 64 matrices)
    A = lap((12, 12, 12)); n = A.shape[0]
    Ap, Ai = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    nb = 64
    AX = np.stack([(A + 0.1 * i * sp.identity(n)).tocsc().data for i in range(nb)])
    B = rng.standard_normal((nb, n, 1))
    with hip.Factorization(n, n, Ap, Ai, kind=hip.CS3_CHOLESKY, batch=nb) as F:
        F.factor(AX)
        X = F.solve(B)
        print("batch max_front", F.info.max_front, "levels", F.info.nlevels)
    for i in (0, 31, 63):
        Ai_ = (A + 0.1 * i * sp.identity(n)).tocsc()
        r = np.abs(Ai_ @ X[i] - B[i]).max() / (abs(Ai_).sum(axis=0).max() * np.abs(X[i]).max() + np.abs(B[i]).max())
        assert r < 1e-13, r
    print("laplacians ok")
