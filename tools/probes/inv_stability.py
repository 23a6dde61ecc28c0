import numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, "/root/repo")
from csparse3_amd import csc_hip as hip, synth
for scale in (1.0, 0.1, 0.02, 0.005, 0.002):
    m, n, Ap, Ai, Ax = synth.dense_block_matrix(500, 260, seed=5)
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).tolil()
    d = A.diagonal()
    A.setdiag(d * scale)
    A = A.tocsc(); A.sort_indices()
    Ap2, Ai2, Ax2 = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    b = np.random.default_rng(0).standard_normal(n)
    try:
        with hip.Factorization(n, n, Ap2, Ai2) as F:
            F.factor(Ax2, 1e-3)
            Lp, Li, Lx, Up, Ui, Ux = F.factors()
            x1 = F.solve(b)
            B = np.tile(b[:, None], (1, 32)).copy()
            X = F.solve(B)
        r1 = np.abs(A @ x1 - b).max() / (np.abs(A).max() * np.abs(x1).max() + np.abs(b).max())
        r2 = np.abs(A @ X[:, 0] - b).max() / (np.abs(A).max() * np.abs(X[:, 0]).max() + np.abs(b).max())
        print("diag scale %g: max|L| %.3g  residual k=1 %.2e  k=32 %.2e  |x1 - X0|/|x1| %.2e" % (scale, np.abs(Lx).max(), r1, r2, np.abs(x1 - X[:, 0]).max() / np.abs(x1).max()))
    except Exception as e:
        print("diag scale %g: %s" % (scale, str(e)[:100]))
