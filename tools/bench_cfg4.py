#!/usr/bin/env python3
"""Config 4 on one GPU: factor the 50k matrix once, time the full solve of --rhs right-hand sides.
    python tools/bench_cfg4.py --rhs 1024 [--reps 20]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csparse3_amd import csc_hip as hip, synth

ap = argparse.ArgumentParser()
ap.add_argument("--rhs", type=int, default=1024)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
sh = torch.cuda.current_stream().cuda_stream
m, n, Ap, Ai, Ax = synth.grid_jacobian()
F = hip.Factorization(m, n, Ap, Ai)
F.factor(Ax, 1e-3)
inf = F.info; nnz_l, nnz_u = int(inf.nnz_l), int(inf.nnz_u)
B = torch.from_numpy(synth.grid_rhs(n, args.rhs)).to(dev); X = torch.empty_like(B)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
def solve(e=None):
    X.copy_(B)                                   # the solve is in place: fresh right-hand sides, outside the timed bracket
    if e: e[0].record()
    F.solve_dev(X.data_ptr(), args.rhs, sh)
    if e: e[1].record()
for _ in range(3): solve()
torch.cuda.synchronize()
for i in range(args.reps): solve(ev[i])
torch.cuda.synchronize(); t = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e-3
k = args.rhs
bytes_solve = (12 * nnz_l + 4 * (n + 1) + 16 * n * k) + (12 * nnz_u + 4 * (n + 1) + 16 * n * k) + 2 * 8 * n * k
import scipy.sparse as sp
A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
res = np.abs(A @ X[:, :4].cpu().numpy() - B[:, :4].cpu().numpy()).max()
print(json.dumps({"rhs": k, "ms": 1e3 * t, "nnz_per_s": (nnz_l + nnz_u) * k / t, "algorithmic_GBs": bytes_solve / t / 1e9,
                  "frac": bytes_solve / t / 1e9 / 8000.0, "residual": float(res)}))
