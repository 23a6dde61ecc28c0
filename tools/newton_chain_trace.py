#!/usr/bin/env python3
"""Device-resident Newton chain (restack values -> refactor + solve) for a kernel trace: every iteration must consist of
kernels only -- no __amd_rocclr_copyBuffer between the marker kernels.   rocprofv3 --kernel-trace ... -- python3 tools/newton_chain_trace.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csparse3_amd import csc_hip as hip, synth
import scipy.sparse as sp

dev = torch.device("cuda", 0); sh = torch.cuda.current_stream().cuda_stream
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
m, n, Ap, Ai, Ax = synth.jacobian_config2()
J = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)); k = 219
blocks = []
for rs, cs in ((slice(0, k), slice(0, k)), (slice(0, k), slice(k, n)), (slice(k, n), slice(0, k)), (slice(k, n), slice(k, n))):
    B = J[rs, cs].tocsc(); B.sort_indices()
    blocks.append((B.shape[0], B.shape[1], int(B.nnz), T(B.indices.astype(np.int32)), T(B.indptr.astype(np.int32)), T(B.data.copy())))
nz = [b[2] for b in blocks]; nnz = sum(nz)
Pi = torch.empty(nnz, dtype=torch.int32, device=dev); Pp = torch.empty(n + 1, dtype=torch.int32, device=dev)
Px = torch.empty(nnz, dtype=torch.float64, device=dev); mp = torch.empty(nnz, dtype=torch.int32, device=dev)
hip.csc_stack_4_by_4_dev([(a, b_, c, i.data_ptr(), p.data_ptr(), x.data_ptr()) for (a, b_, c, i, p, x) in blocks], Pi.data_ptr(), Pp.data_ptr(), Px.data_ptr(), mp.data_ptr(), sh)
F = hip.Factorization(n, n, Ap, Ai)
d_b = T(np.random.default_rng(0).standard_normal(n)); d_x = torch.empty_like(d_b)
vals = [b[5] for b in blocks]
F.factor_solve_bx_dev(Px.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), 1, 1e-3, sh); F.factor_status(sh)   # capture
marker = torch.zeros(1, device=dev)
torch.cuda.synchronize()
marker.add_(1.0)                                   # marker kernel: loop begins
for it in range(5):
    for v in vals: v.mul_(1.001)                   # new block values, produced on the device
    hip.restack_values_dev(nnz, mp.data_ptr(), nz[0], nz[1], nz[2], *[v.data_ptr() for v in vals], Px.data_ptr(), sh)
    F.factor_solve_bx_dev(Px.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), 1, 1e-3, sh)
marker.add_(1.0)                                   # marker kernel: loop ends
torch.cuda.synchronize()
F.factor_status(sh)
A = sp.csc_matrix((Px.cpu().numpy(), Ai, Ap), shape=(n, n))
print("residual", float(np.abs(A @ d_x.cpu().numpy() - d_b.cpu().numpy()).max()))
