#!/usr/bin/env python3
"""Per-front phase timings from the in-kernel shader-clock stamps (CS3_PROFILE=1).
Stamps (cycles since the block started): 0 descriptor issued, 1 LDS image zeroed, 2 assembled (gather done),
3 rows/entries in registers, 4 eliminated, 5 stored.  Diagnostic only."""
import ctypes as C, os, sys
os.environ["CS3_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from csparse3_amd import csc_hip as hip, synth

m, n, Ap, Ai, Ax = synth.grid_jacobian()
F = hip.Factorization(m, n, Ap, Ai)
for _ in range(5):
    F.factor(Ax, 1e-3)
ns = int(F.info.nsuper)
out = np.zeros((ns, 8), dtype=np.int64)
sched = np.zeros(ns, dtype=np.int32); fr = np.zeros(ns, dtype=np.int32); fw = np.zeros(ns, dtype=np.int32)
L = hip.lib()
L.cs3_debug_front_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
L.cs3_debug_schedule.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 3
assert L.cs3_debug_front_stamps(F._h, out.ctypes.data_as(C.POINTER(C.c_int64))) == 0
L.cs3_debug_schedule(F._h, *[a.ctypes.data_as(C.POINTER(C.c_int32)) for a in (sched, fr, fw)])
lvl = F.supernodes()[2][sched]
valid = out[:, 5] > 0
names = ["desc", "zero", "gather", "load", "elim", "store"]
d = np.diff(np.concatenate([np.zeros((ns, 1), dtype=np.int64), out[:, :6]], axis=1), axis=1)
for name, lo, hi in [("r<=16", 0, 16), ("r<=32", 16, 32), ("r<=64", 32, 64), ("r<=136", 64, 136)]:
    msk = (fr > lo) & (fr <= hi) & valid
    if msk.sum():
        print(name, "n", msk.sum(), "median cycles per phase", dict(zip(names, np.median(d[msk], axis=0).astype(int))),
              "elim/pivot median %.0f" % np.median(d[msk, 4] / np.maximum(fw[msk], 1)), "w median %d max %d" % (np.median(fw[msk]), fw[msk].max()))
for l in range(int(lvl.max()) + 1):
    msk = (lvl == l) & valid
    if msk.sum():
        j = np.flatnonzero(msk)[np.argmax(out[msk, 5])]
        print("level", l, "fronts", msk.sum(), "slowest: r,w =", fr[j], fw[j], "phases", d[j, :6])
# blocked big fronts: phases of the block-column tile (1, 0) in the step kb = 64
big = np.flatnonzero((fr > 136) & (out[:, 5] > 0))
for j in big:
    print("big front r,w =", fr[j], fw[j], "step kb=64 tile(1,0) cycles since kernel start: loads issued %d, staged %d, updated %d, "
          "D/T in LDS %d, D factored %d, panel solved+stored %d; block-row tile (0,1): eliminated %d, stored %d" % tuple(out[j, :8]))
