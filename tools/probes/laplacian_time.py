import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, torch
from csparse3_amd import csc_hip as hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from laplacians import lap
dev = torch.device("cuda", 0); sh = torch.cuda.current_stream().cuda_stream
for name, dims in (("2d_200x200", (200, 200)), ("3d_30", (30, 30, 30))):
    A = lap(dims); n = A.shape[0]
    Ap, Ai, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    for kind, kname in ((hip.CS3_CHOLESKY, "chol"), (hip.CS3_LU, "lu")):
        with hip.Factorization(n, n, Ap, Ai, kind=kind) as F:
            d_ax = torch.from_numpy(Ax).to(dev)
            for _ in range(3): F.factor_dev(d_ax.data_ptr(), 1e-3, sh)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): F.factor_dev(d_ax.data_ptr(), 1e-3, sh)
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
            F.factor_status(sh)
            fl = float(F.info.flops_factor)
            print("%-12s %-5s factor %.3f ms  %.1f GFLOP  %.2f TFLOP/s" % (name, kname, 1e3 * t, fl / 1e9, fl / t / 1e12), flush=True)
