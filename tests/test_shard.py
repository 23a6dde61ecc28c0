"""The N > 1 paths: world_size-2 gloo runs on CPU (two real processes), with the oracle standing in
for the GPU backend so that partitioning and collectives are what is under test; plus a GPU test of
the factor export / import that the broadcast relies on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from csparse3_amd import shard, synth


class OracleBackend:
    """CPU stand-in with the HipBackend interface (test infrastructure only)."""

    def __init__(self, m, n, Ap, Ai, kind=0, batch=1):
        from oracle import oracle as orc
        self.orc, self.n, self.Ap, self.Ai, self.kind, self.batch = orc, n, Ap, Ai, kind, batch
        self.device = torch.device("cpu")
        self.q = orc.csc_amd_f(1, n, n, Ap, Ai)
        self.f = None

    def factor(self, Ax, tol=0.0):
        Ax = np.asarray(Ax, dtype=np.float64).reshape(self.batch, -1)
        self.f = [self.orc.csc_lu_f(self.n, self.n, self.Ap, self.Ai, Ax[i], self.q, max(tol, 1e-3))
                  for i in range(self.batch)]
        self.sizes = [len(self.f[0][2]), len(self.f[0][5])]

    def export_factor(self):
        return torch.from_numpy(np.concatenate([np.concatenate([f[2], f[5]]) for f in self.f]))

    def empty_factor(self):
        self.factor(self._probe_values)            # sizes come from the shared symbolic structure
        return torch.empty(sum(self.sizes) * self.batch, dtype=torch.float64)

    def import_factor(self, buf):
        v = buf.numpy().reshape(self.batch, -1)
        self.f = [(f[0], f[1], v[i, :self.sizes[0]].copy(), f[3], f[4], v[i, self.sizes[0]:].copy(), f[6])
                  for i, f in enumerate(self.f)]

    def to_device(self, a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)

    def solve(self, B):
        X = B.numpy()
        Xb = X.reshape(self.batch, self.n, -1)
        for i, (Lp, Li, Lx, Up, Ui, Ux, pinv) in enumerate(self.f):
            for t in range(Xb.shape[2]):
                x = np.empty(self.n); x[pinv] = Xb[i, :, t]
                self.orc.csc_lsolve_f(self.n, Lp, Li, Lx, x)
                self.orc.csc_usolve_f(self.n, Up, Ui, Ux, x)
                Xb[i, self.q, t] = x
        return B


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if mode == "rhs":
            m, n, Ap, Ai, Ax = synth.grid_jacobian(n=400, seed=3)
            B = np.random.default_rng(0).standard_normal((n, 7))          # 7 columns over 2 ranks: 4 + 3
            be = OracleBackend(m, n, Ap, Ai)
            be._probe_values = Ax
            tm = {}
            X = shard.solve_many_rhs(be, Ax if rank == 0 else None, B if rank == 0 else None, tol=1e-3,
                                     timings=tm, sync=dist.barrier)
            assert set(tm) == {"factor", "broadcast", "scatter", "solve", "gather"} and all(v >= 0.0 for v in tm.values())
            if rank == 0:
                np.save(out, X.numpy())
        else:
            n = 300
            ei, ej = synth.spd_grid_pattern(n, seed=5)
            mats = [synth.spd_grid_matrix(n, ei, ej, seed=50 + i) for i in range(5)]   # 5 matrices: 3 + 2
            m, n, Ap, Ai, _ = mats[0]
            AX = np.stack([mm[4] for mm in mats])
            B = np.random.default_rng(1).standard_normal((5, n, 2))
            X = shard.solve_many_matrices(lambda b: OracleBackend(m, n, Ap, Ai, batch=b), AX, B.copy(), tol=1e-3)   # (the CPU stand-in solves in place)
            # every rank holding only its own slice of the values gives the same result
            lo, hi = shard.shard_range(5, world, rank)
            tm = {}
            X2 = shard.solve_many_matrices(lambda b: OracleBackend(m, n, Ap, Ai, batch=b), AX[lo:hi], B.copy(), tol=1e-3,
                                           local_values=True, total=5, timings=tm, sync=dist.barrier)
            assert set(tm) == {"upload", "factor_solve", "gather"}
            if rank == 0:
                assert torch.equal(X, X2)
                np.save(out, X.numpy())
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_balanced_partition():
    for total in (0, 1, 7, 64, 1024):
        for world in (1, 2, 3, 8):
            parts = [shard.shard_range(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("mode", ["rhs", "matrices"])
def test_two_ranks_gloo(tmp_path, mode):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    out = str(tmp_path / "x.npy")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    X = np.load(out)
    if mode == "rhs":
        m, n, Ap, Ai, Ax = synth.grid_jacobian(n=400, seed=3)
        B = np.random.default_rng(0).standard_normal((n, 7))
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        assert X.shape == (n, 7)
        assert np.abs(A @ X - B).max() <= 1e-11
        assert np.abs(X - spla.spsolve(A, B)).max() <= 1e-10 * np.abs(X).max()
    else:
        n = 300
        ei, ej = synth.spd_grid_pattern(n, seed=5)
        B = np.random.default_rng(1).standard_normal((5, n, 2))
        assert X.shape == (5, n, 2)
        for i in range(5):
            m, n, Ap, Ai, Ax = synth.spd_grid_matrix(n, ei, ej, seed=50 + i)
            A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
            assert np.abs(A @ X[i] - B[i]).max() <= 1e-10


def _gpu_worker(rank, world, port, mode, out):
    """Two ranks sharing cuda:0, collectives over gloo: the N > 1 code path of bench.py's sharding legs with the REAL
    backend (factor export -> broadcast -> import, RHS slabs, matrix slices)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        if mode == "rhs":
            m, n, Ap, Ai, Ax = synth.grid_jacobian(n=3000, seed=3)
            B = np.random.default_rng(0).standard_normal((n, 37))         # 37 columns over 2 ranks: 19 + 18 (lane = RHS path)
            be = shard.HipBackend(m, n, Ap, Ai, device=dev)
            X = shard.solve_many_rhs(be, Ax if rank == 0 else None, B if rank == 0 else None, tol=1e-3)
            if rank == 0:
                np.save(out, X.cpu().numpy())
        else:
            n = 1200
            ei, ej = synth.spd_grid_pattern(n, seed=5)
            m, n, Ap, Ai, _ = synth.spd_grid_matrix(n, ei, ej, seed=50)
            AX = np.stack([synth.spd_grid_matrix(n, ei, ej, seed=50 + i)[4] for i in range(5)])
            B = np.random.default_rng(1).standard_normal((5, n, 2))
            from csparse3_amd import csc_hip
            X = shard.solve_many_matrices(lambda b: shard.HipBackend(m, n, Ap, Ai, kind=csc_hip.CS3_CHOLESKY, batch=b, device=dev),
                                          AX, B, tol=0.0)
            if rank == 0:
                np.save(out, X.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rhs", "matrices"])
def test_two_ranks_gloo_on_one_gpu(gpu, tmp_path, mode):
    import scipy.sparse as sp
    out = str(tmp_path / "x.npy")
    mp.spawn(_gpu_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    X = np.load(out)
    if mode == "rhs":
        m, n, Ap, Ai, Ax = synth.grid_jacobian(n=3000, seed=3)
        B = np.random.default_rng(0).standard_normal((n, 37))
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        assert X.shape == (n, 37)
        assert np.abs(A @ X - B).max() <= 1e-10 * np.abs(B).max()
        with gpu.Factorization(m, n, Ap, Ai) as F:                          # the same through one handle
            F.factor(Ax, 1e-3)
            assert np.abs(X - F.solve(B)).max() <= 1e-12 * np.abs(X).max()
    else:
        n = 1200
        ei, ej = synth.spd_grid_pattern(n, seed=5)
        B = np.random.default_rng(1).standard_normal((5, n, 2))
        assert X.shape == (5, n, 2)
        for i in range(5):
            m, n, Ap, Ai, Ax = synth.spd_grid_matrix(n, ei, ej, seed=50 + i)
            A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
            assert np.abs(A @ X[i] - B[i]).max() <= 1e-10 * np.abs(B[i]).max()


@pytest.mark.gpu
def test_factor_export_import_roundtrip_on_the_gpu(gpu):
    """What solve_many_rhs broadcasts: a second handle with the same analysis solves from imported panels."""
    m, n, Ap, Ai, Ax = synth.grid_jacobian(n=3000, seed=4)
    b = np.random.default_rng(3).standard_normal((n, 4))
    be0 = shard.HipBackend(m, n, Ap, Ai)
    be1 = shard.HipBackend(m, n, Ap, Ai)
    be0.factor(Ax, 1e-3)
    be1.import_factor(be0.export_factor().clone())
    be1.F.factor_status(be1.stream())          # an importing handle never ran a prologue: its status word must read clean
    assert be1.F.info.fail_col == -1
    x0 = be0.solve(be0.to_device(b)).cpu().numpy()
    x1 = be1.solve(be1.to_device(b)).cpu().numpy()
    assert np.array_equal(x0, x1)
    # and the single-process path of both drivers
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        X = shard.solve_many_rhs(shard.HipBackend(m, n, Ap, Ai), Ax, b, tol=1e-3).cpu().numpy()
        assert np.array_equal(X, x0)
    finally:
        dist.destroy_process_group()
