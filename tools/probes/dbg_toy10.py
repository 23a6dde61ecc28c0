import numpy as np, sys
sys.path.insert(0, '.')
from csparse3_amd import csc_hip as hip, synth
from oracle import oracle as orc
m, n, Ap, Ai, Ax = synth.toy10()[:5]
F = hip.Factorization(m, n, Ap, Ai)
F.factor(Ax, 1e-3)
Lp, Li, Lx, Up, Ui, Ux = F.factors()
sn = F.supernodes()
print("sn", sn)
import scipy.sparse as sp
L = sp.csc_matrix((Lx, Li, Lp), shape=(n, n)).toarray(); U = sp.csc_matrix((Ux, Ui, Up), shape=(n, n)).toarray()
q = F.ordering()["q"]
A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).toarray()[np.ix_(q, q)]
E = L @ U - A
np.set_printoptions(linewidth=200, precision=3, suppress=True)
print(np.abs(E).max(axis=0)); print(np.abs(E).max(axis=1))
