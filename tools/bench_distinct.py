#!/usr/bin/env python3
"""Config 5, second variant: SPD 5k x 5k matrices with DISTINCT patterns, one handle each, factor + solve on a pool of
streams (csparse3_amd.streams.DistinctBatch) -- against the same matrices one after the other on one stream and against
a same-pattern batch of the same size.   python tools/bench_distinct.py [--nmat 64] [--streams 8]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csparse3_amd import csc_hip as hip, synth
from csparse3_amd.streams import DistinctBatch

ap = argparse.ArgumentParser()
ap.add_argument("--nmat", type=int, default=64)
ap.add_argument("--streams", type=int, default=8)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
n5 = 5000
mats = []
for i in range(args.nmat):
    ei, ej = synth.spd_grid_pattern(n5, seed=7000 + i)              # the chords differ from matrix to matrix
    mats.append(synth.spd_grid_matrix(n5, ei, ej, seed=5000 + i))
B = [np.random.default_rng(i).standard_normal(n5) for i in range(args.nmat)]
vals = [torch.from_numpy(mm[4]).to(dev) for mm in mats]

def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / args.reps

out = {"nmat": args.nmat, "n": n5}
for workers in (1, None):                               # analysis of the distinct patterns: in line, then in the thread pool
    with DistinctBatch([(mm[0], mm[1], mm[2], mm[3]) for mm in mats], kind=hip.CS3_CHOLESKY, nstreams=1, workers=workers) as D:
        out["analysis_s_workers_%d" % D.workers] = D.analysis_s
for ns in (1, args.streams):
    with DistinctBatch([(mm[0], mm[1], mm[2], mm[3]) for mm in mats], kind=hip.CS3_CHOLESKY, nstreams=ns) as D:
        rhs = [torch.from_numpy(b.copy()).to(dev) for b in B]
        t = timed(lambda: D.factor_solve(vals, rhs))
        D.status()
        out["streams_%d_ms" % ns] = 1e3 * t
        out["streams_%d_matrices_per_s" % ns] = args.nmat / t
print(json.dumps(out))
