#!/usr/bin/env python3
"""The neighbours of the path at config-3 scale, for a kernel trace: every substrate entry point
once on the 50k x 50k Jacobian (501k nnz).  Run under rocprofv3 --kernel-trace; tools/summarize_substrate.py turns the
trace into GB/s per kernel.   python tools/bench_substrate.py"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csparse3_amd import csc_hip as hip, synth

m, n, Ap, Ai, Ax = synth.grid_jacobian()
nnz = int(Ap[n])
rng = np.random.default_rng(0)
x = rng.standard_normal(n)
out = {"n": n, "nnz": nnz}
def timed(name, fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    out[name + "_host_ms"] = 1e3 * (time.perf_counter() - t0) / reps       # host-pointer API: PCIe + allocation included
    return r
timed("matvec", lambda: hip.csc_mat_vec_ff(m, n, Ap, Ai, Ax, x))
timed("norm", lambda: hip.csc_norm(n, Ap, Ax))
_, _, Tp, Ti, Tx = timed("transpose", lambda: hip.csc_transpose(m, n, Ap, Ai, Ax))
timed("add", lambda: hip.csc_add_ff(m, n, Ap, Ai, Ax, m, n, Tp, Ti, Tx, 1.0, 1.0))
cols = np.repeat(np.arange(n, dtype=np.int32), np.diff(Ap)); perm = rng.permutation(nnz)
timed("coo_to_csc", lambda: hip.coo_to_csc(m, n, Ai[perm], cols[perm], Ax[perm], nnz))
rows = np.sort(rng.choice(n, size=2000, replace=False)).astype(np.int32); cs = np.sort(rng.choice(n, size=2000, replace=False)).astype(np.int32)
timed("sub_matrix_2000x2000", lambda: hip.csc_sub_matrix(m, nnz, Ap, Ai, Ax, rows, cs))
timed("find_islands", lambda: hip.find_islands(n, Ap, Ai))
h = n // 2
import scipy.sparse as sp
A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
blk = []
for rs, c2 in ((slice(0, h), slice(0, h)), (slice(0, h), slice(h, n)), (slice(h, n), slice(0, h)), (slice(h, n), slice(h, n))):
    Bk = A[rs, c2].tocsc(); Bk.sort_indices()
    blk += [Bk.shape[0], Bk.shape[1], Bk.indices.astype(np.int32), Bk.indptr.astype(np.int32), Bk.data]
timed("stack_4_by_4", lambda: hip.csc_stack_4_by_4_ff(*blk))
print(json.dumps(out))
