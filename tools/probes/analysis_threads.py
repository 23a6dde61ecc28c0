"""Probe: wall time of 64 distinct analyses in a thread pool, with and without glibc malloc tuned to keep memory."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "mallopt":
    libc = C.CDLL("libc.so.6")
    print("mallopt", libc.mallopt(-3, 1 << 30), libc.mallopt(-1, 1 << 30))
from csparse3_amd import csc_hip as hip, synth
from concurrent.futures import ThreadPoolExecutor
mats = []
for i in range(64):
    ei, ej = synth.spd_grid_pattern(5000, seed=7000 + i); mats.append(synth.spd_grid_matrix(5000, ei, ej, seed=5000 + i))
make = lambda p: hip.Factorization(p[0], p[1], p[2], p[3], kind=hip.CS3_CHOLESKY)
for rep in range(2):
    for w in (1, 4, 8, 16, 32):
        t = time.perf_counter()
        with ThreadPoolExecutor(w) as pool: hs = list(pool.map(make, mats))
        dt = time.perf_counter() - t
        for h in hs: h.close()
        print("workers %2d: %.1f ms" % (w, 1e3 * dt))
