#!/usr/bin/env python3
"""Phase timings of the bottom-forest factor kernel from its in-kernel shader-clock stamps (CS3_PROFILE=1): for the
task that finishes last in every tier, the chain of its local levels.  Diagnostic only."""
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
L = hip.lib()
L.cs3_debug_front_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
L.cs3_debug_forest.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 4
L.cs3_debug_forest.restype = C.c_int64
L.cs3_debug_schedule.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 3
assert L.cs3_debug_front_stamps(F._h, out.ctypes.data_as(C.POINTER(C.c_int64))) == 0
sn = np.zeros(ns, np.int32); task = np.zeros(ns, np.int32); lvl = np.zeros(ns, np.int32); tier = np.zeros(ns, np.int32)
p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
nf = L.cs3_debug_forest(F._h, p(sn), p(task), p(lvl), p(tier))
sched = np.zeros(ns, np.int32); fr = np.zeros(ns, np.int32); fw = np.zeros(ns, np.int32)
L.cs3_debug_schedule(F._h, p(sched), p(fr), p(fw))
r = np.zeros(ns, np.int32); w = np.zeros(ns, np.int32); r[sched] = fr; w[sched] = fw
par = F.supernodes()[1]
nch = np.bincount(par[par >= 0], minlength=ns)
print("forest fronts", nf, "of", ns)
st = out[:nf]
for t in range(int(tier[:nf].max()) + 1):
    idx = np.flatnonzero(tier[:nf] == t)
    last = idx[np.argmax(st[idx, 5])]
    k = task[last]
    mine = idx[task[idx] == k]
    print("tier", t, ": slowest task", k, "fronts", len(mine), "ends at", st[last, 5], "cycles")
    for l in range(int(lvl[mine].max()) + 1):
        fl = mine[lvl[mine] == l]
        j = fl[np.argmax(st[fl, 5])]
        if w[sn[j]] >= 6:      # (ForestLimits::coop_w) shared by four waves: stamps per part
            print("  level %2d fronts %3d  shared r,w,children = %2d,%2d,%2d  start %7d | part 0 stored %7d | parts 1-3 (assembled, eliminated): %s" % (
                l, len(fl), r[sn[j]], w[sn[j]], nch[sn[j]], st[j, 0], st[j, 1], " ".join("(%d, %d)" % (st[j, 2 * q], st[j, 2 * q + 1]) for q in (1, 2, 3))))
            continue
        d = np.diff(st[j, :6])
        print("  level %2d fronts %3d  slowest r,w,children = %2d,%2d,%2d  start %7d  A %5d children %5d regs %5d elim %6d (%4.0f/pivot) store %5d  end %7d  barrier %7d" % (
            l, len(fl), r[sn[j]], w[sn[j]], nch[sn[j]], st[j, 0], d[0], d[1], d[2], d[3], d[3] / max(1, w[sn[j]]), d[4], st[j, 5], st[fl, 6].max()))
if os.environ.get("CS3_STAMPS_LEVEL0"):
    idx = np.flatnonzero(tier[:nf] == 0)
    last = idx[np.argmax(st[idx, 5])]
    mine = idx[task[idx] == task[last]]
    fl = mine[lvl[mine] == 0]
    for q, j in enumerate(fl):
        d = np.diff(st[j, :6])
        print("  wave %d  r,w = %2d,%2d  start %6d  A %5d children %5d regs %5d elim %6d store %5d (L %5d U %5d CB %5d) end %6d" % (q % 8, r[sn[j]], w[sn[j]], st[j, 0], d[0], d[1], d[2], d[3], d[4], st[j, 6] - st[j, 4], st[j, 7] - st[j, 6], st[j, 5] - st[j, 7], st[j, 5]))
if os.environ.get("CS3_STAMPS_ALL"):
    idx = np.flatnonzero(tier[:nf] == 0)
    last = idx[np.argmax(st[idx, 5])]
    mine = idx[task[idx] == task[last]]
    for j in sorted(mine, key=lambda j: st[j, 0]):
        print("  lvl %d r,w = %2d,%2d  stamps %s" % (lvl[j], r[sn[j]], w[sn[j]], " ".join("%7d" % v for v in st[j])))
