"""Repeat factor + solve / lsolve / usolve on the fixtures in one process and count non-finite or wrong results."""
import sys, numpy as np
sys.path.insert(0, '.')
from csparse3_amd import csc_hip as hip
Z = np.load('tests/golden/factor_fixtures.npz')
bad = 0
for tag in ("block500", "grid1500", "jac118"):
    g = lambda k: Z[tag + "_" + k]
    n = int(g("n")); Ap, Ai, Ax, b, q, want = g("Ap"), g("Ai"), g("Ax"), g("b"), g("q"), g("x")
    for it in range(150):
        with hip.Factorization(n, n, Ap, Ai) as F:
            F.factor(Ax, 1e-3)
            if it % 2: F.factors()
            x = F.solve(b)
            y = F.lsolve(b[q]); zz = F.usolve(y)
        e1 = np.abs(x - want).max(); e2 = np.abs(zz - want[q]).max()
        if not (e1 <= 1e-10 * np.abs(want).max() and e2 <= 1e-10 * np.abs(want).max()):
            bad += 1
            print(tag, it, "x err", e1, "z err", e2, "nan in y", np.isnan(y).sum(), "nan in z", np.isnan(zz).sum(), flush=True)
print("bad", bad)
