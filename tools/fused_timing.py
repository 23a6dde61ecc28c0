"""Split (factor_dev + solve_dev) vs fused (factor_solve_dev) step time on BASELINE config 3."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csparse3_amd import synth, csc_hip as hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rhs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = 200
m, n, Ap, Ai, Ax = synth.grid_jacobian(n=n, seed=n)
b = synth.grid_rhs(n, rhs, seed=1024)
dev = torch.device("cuda", 0)
F = hip.Factorization(m, n, Ap, Ai, hip.CS3_LU, hip.ORDER_AMD)
d_ax = torch.from_numpy(Ax).to(dev); d_b = torch.from_numpy(np.ascontiguousarray(b)).to(dev); d_x = torch.empty_like(d_b)
sh = torch.cuda.current_stream().cuda_stream

def split():
    F.factor_dev(d_ax.data_ptr(), 1e-3, sh); d_x.copy_(d_b); F.solve_dev(d_x.data_ptr(), rhs, sh)
def fused():
    d_x.copy_(d_b); F.factor_solve_dev(d_ax.data_ptr(), d_x.data_ptr(), rhs, 1e-3, sh)

out = {}
for name, fn in (("split", split), ("fused", fused), ("split2", split), ("fused2", fused)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); out[name] = 1e3 * (time.perf_counter() - t0) / steps
    F.factor_status(sh)
    out[name + "_x0"] = float(d_x.flatten()[0])
print(json.dumps(out))
