#!/usr/bin/env python3
"""Secondary measurements: BASELINE configs 4 and 5 on ONE GPU (the per-rank slice of the 8-GPU configs).
    python tools/bench_configs.py [--rhs 128] [--nmat 64]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csparse3_amd import csc_hip as hip, synth

ap = argparse.ArgumentParser()
ap.add_argument("--rhs", type=int, default=128)     # config 4: 1024 RHS over 8 GPUs = 128 per GPU
ap.add_argument("--nmat", type=int, default=64)     # config 5: 512 matrices over 8 GPUs = 64 per GPU
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
sh = torch.cuda.current_stream().cuda_stream
out = {}

def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

# ---- config 4 slice: factor once, many RHS
m, n, Ap, Ai, Ax = synth.grid_jacobian()
F = hip.Factorization(m, n, Ap, Ai)
F.factor(Ax, 1e-3)
inf = F.info; nnz_lu = int(inf.nnz_l + inf.nnz_u)
B = torch.from_numpy(synth.grid_rhs(n, args.rhs)).to(dev); X = torch.empty_like(B)
def solve():
    X.copy_(B); F.solve_dev(X.data_ptr(), args.rhs, sh)
t = timed(solve, args.reps)
bytes_solve = 12 * nnz_lu + 8 * (n + 1) + 2 * 16 * n * args.rhs + 16 * n * args.rhs
A = None
x0 = X[:, 0].cpu().numpy()
import scipy.sparse as sp
A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
res = np.abs(A @ x0 - B[:, 0].cpu().numpy()).max()
out["config4_slice"] = {"rhs": args.rhs, "ms": 1e3 * t, "nnz_per_s": nnz_lu * args.rhs / t,
                        "algorithmic_GBs": bytes_solve / t / 1e9, "residual": float(res)}
F.close()

# ---- config 5 slice: batch of SPD 5k x 5k matrices, same pattern, Cholesky factor + 1 RHS each
n5 = 5000
ei, ej = synth.spd_grid_pattern(n5, seed=5000)
mats = [synth.spd_grid_matrix(n5, ei, ej, seed=5000 + i) for i in range(args.nmat)]
m5, n5, Ap5, Ai5, _ = mats[0]
AX = torch.from_numpy(np.stack([mm[4] for mm in mats])).to(dev)
G = hip.Factorization(m5, n5, Ap5, Ai5, kind=hip.CS3_CHOLESKY, batch=args.nmat)
inf5 = G.info
B5 = torch.from_numpy(np.random.default_rng(0).standard_normal((args.nmat, n5, 1))).to(dev); X5 = torch.empty_like(B5)
def step5():
    G.factor_dev(AX.data_ptr(), 0.0, sh)
    X5.copy_(B5); G.solve_dev(X5.data_ptr(), 1, sh)
t5_split = timed(step5, args.reps)
G.factor_status(sh)
x5_split = X5.clone()
def step5_fused():
    X5.copy_(B5); G.factor_solve_dev(AX.data_ptr(), X5.data_ptr(), 1, 0.0, sh)
t5 = timed(step5_fused, args.reps)
G.factor_status(sh)
assert torch.equal(X5, x5_split)
nnzl = int(inf5.nnz_l)
bytes5 = args.nmat * (12 * (int(Ap5[n5]) // 2 + n5) + 12 * nnzl + 8 * (n5 + 1) + 2 * (12 * nnzl + 16 * n5))
A5 = sp.csc_matrix((mats[3][4], Ai5, Ap5), shape=(n5, n5))
res5 = np.abs(A5 @ X5[3, :, 0].cpu().numpy() - B5[3, :, 0].cpu().numpy()).max()
out["config5_slice"] = {"nmat": args.nmat, "n": n5, "nnz_l": nnzl, "levels": int(inf5.nlevels), "ms": 1e3 * t5,
                        "ms_factor_then_solve": 1e3 * t5_split,
                        "matrices_per_s": args.nmat / t5, "nnz_per_s": args.nmat * 3 * nnzl / t5,
                        "algorithmic_GBs": bytes5 / t5 / 1e9, "residual": float(res5)}
print(json.dumps(out))
