#!/usr/bin/env python3
"""Phase timings of k_front_wg (one workgroup per big front and matrix) on the config-5 batch, from the in-kernel
shader-clock stamps (CS3_PROFILE=1).  Diagnostic only.   python tools/wg_stamps.py [nmat]"""
import ctypes as C, os, sys
os.environ["CS3_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from csparse3_amd import csc_hip as hip, synth

nmat = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n5 = 5000
ei, ej = synth.spd_grid_pattern(n5, seed=5000)
mats = [synth.spd_grid_matrix(n5, ei, ej, seed=5000 + i) for i in range(nmat)]
m, n, Ap, Ai, _ = mats[0]
AX = np.stack([mm[4] for mm in mats])
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
for j in np.flatnonzero(fr > 64):
    print("front r=%d w=%d: zeroed %d gathered %d | panels %d update %d steps %d | end %d cycles" %
          (fr[j], fw[j], out[j, 0], out[j, 1], out[j, 2], out[j, 3], out[j, 4], out[j, 5]))
