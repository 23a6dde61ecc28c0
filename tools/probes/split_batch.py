"""Probe: one 512-matrix handle against P handles of 512 / P matrices on P streams (config 5 shape)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from csparse3_amd import synth, csc_hip as hip

nmat, n5 = 512, 5000
ei, ej = synth.spd_grid_pattern(n5, seed=5000)
m, n, Ap, Ai, _ = synth.spd_grid_matrix(n5, ei, ej, seed=5000)
AX = np.stack([synth.spd_grid_matrix(n5, ei, ej, seed=5000 + i)[4] for i in range(nmat)])
B = np.random.default_rng(1).standard_normal((nmat, n, 1))
dev = torch.device("cuda", 0)
d_ax = torch.from_numpy(AX).to(dev); d_b = torch.from_numpy(B).to(dev)
out = {}
for P in (1, 2, 1, 2, 3):
    per = nmat // P
    if nmat % P: per = (nmat // P // 64) * 64
    hs = [hip.Factorization(m, n, Ap, Ai, kind=hip.CS3_CHOLESKY, batch=per) for _ in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    xs = [torch.empty_like(d_b[i * per:(i + 1) * per]) for i in range(P)]
    def step():
        for i, h in enumerate(hs):
            h.factor_solve_bx_dev(d_ax[i * per:(i + 1) * per].data_ptr(), d_b[i * per:(i + 1) * per].data_ptr(), xs[i].data_ptr(), 1, 0.0, streams[i].cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); out.setdefault(P, []).append(1e3 * (time.perf_counter() - t0) / 10)
    for h in hs: h.close()
print(json.dumps(out))
