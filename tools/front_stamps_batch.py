#!/usr/bin/env python3
"""Per-front phase timings (CS3_PROFILE=1 shader-clock stamps) for a BATCH of SPD matrices (config 5 shape).
Every matrix of the batch stamps the same slot (the last writer wins): a sample, not a mean.  Diagnostic only."""
import ctypes as C, os, sys
os.environ["CS3_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from csparse3_amd import csc_hip as hip, synth

nmat = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n5 = 5000
ei, ej = synth.spd_grid_pattern(n5, seed=5000)
m, n, Ap, Ai, _ = synth.spd_grid_matrix(n5, ei, ej, seed=5000)
AX = np.stack([synth.spd_grid_matrix(n5, ei, ej, seed=5000 + i)[4] for i in range(nmat)])
F = hip.Factorization(m, n, Ap, Ai, kind=hip.CS3_CHOLESKY, batch=nmat)
for _ in range(3):
    F.factor(AX)
ns = int(F.info.nsuper)
out = np.zeros((ns, 8), dtype=np.int64)
sched = np.zeros(ns, dtype=np.int32); fr = np.zeros(ns, dtype=np.int32); fw = np.zeros(ns, dtype=np.int32)
L = hip.lib()
L.cs3_debug_front_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
L.cs3_debug_schedule.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 3
assert L.cs3_debug_front_stamps(F._h, out.ctypes.data_as(C.POINTER(C.c_int64))) == 0
L.cs3_debug_schedule(F._h, *[a.ctypes.data_as(C.POINTER(C.c_int32)) for a in (sched, fr, fw)])
lvl = F.supernodes()[2][sched]
names = ["desc", "zero", "gather", "load", "elim", "store"]
d = np.diff(np.concatenate([np.zeros((ns, 1), dtype=np.int64), out[:, :6]], axis=1), axis=1)
for l in range(int(lvl.max()) + 1):
    msk = (lvl == l) & (out[:, 5] > 0)
    for lo, hi in [(0, 16), (16, 32), (32, 64), (64, 136), (136, 100000)]:
        mm = msk & (fr > lo) & (fr <= hi)
        if mm.sum():
            print("level %d r in (%d, %d]: %4d fronts, median r %3d w %3d, median cycles" % (l, lo, hi, mm.sum(), np.median(fr[mm]), np.median(fw[mm])),
                  dict(zip(names, np.median(d[mm], axis=0).astype(int))), "total", int(np.median(out[mm, 5])),
                  "block kernel: eliminations %d updates %d" % (np.median(out[mm, 6]), np.median(out[mm, 7])) if lo >= 64 else "")
big = np.flatnonzero((fr > 136) & (out[:, 5] > 0))
for j in big:
    print("wg front r,w =", fr[j], fw[j], "raw stamps", out[j])
