import sys; sys.path.insert(0,'.')
import numpy as np
from csparse3_amd import csc_hip as hip, synth
m,n,Ap,Ai,Ax = synth.dense_block_matrix(n=500, nd=150, seed=1)
for i in range(int(sys.argv[1])):
    with hip.Factorization(m,n,Ap,Ai, batch=18) as F:
        F.factor(np.tile(Ax,(18,1)), 1e-3)
        x = F.solve(np.ones((18,n,1)))
import torch
print("torch init after", sys.argv[1], "handles:", torch.cuda.is_available()); torch.cuda.init(); print("ok", torch.zeros(3, device='cuda').sum().item())
