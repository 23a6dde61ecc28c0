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
            # the same tile by tile (tiles of 2 columns: 2 + 2 on rank 0, 2 + 1 on rank 1) ...
            tm2 = {}
            Xt = shard.solve_many_rhs(be, Ax if rank == 0 else None, B.copy() if rank == 0 else None, tol=1e-3, tile=2,
                                      timings=tm2, sync=dist.barrier)
            assert set(tm2) == {"factor", "broadcast", "pipeline"}
            # ... and with the right-hand sides resident per rank (every rank cuts its own slab), gathered or left sharded
            lo, hi = shard.shard_range(7, world, rank)
            Xr = shard.solve_many_rhs(be, Ax if rank == 0 else None, B[:, lo:hi].copy(), tol=1e-3, resident=True)
            Xs = shard.solve_many_rhs(be, Ax if rank == 0 else None, B[:, lo:hi].copy(), tol=1e-3, resident=True, gather=False)
            assert Xs.shape == (n, hi - lo)
            if rank == 0:
                assert torch.equal(X, Xt) and torch.equal(X, Xr) and torch.equal(X[:, lo:hi], Xs)
                np.save(out, X.numpy())
            else:
                assert Xt is None and Xr is None
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


def _gpu_worker_full(rank, world, port, mode, out):
    """Two ranks sharing cuda:0 at the shapes BASELINE.json quotes for the sharded configurations: 50 000 x 1024 right-hand
    sides split 512 / 512 (tile-pipelined, then resident per rank), 512 matrices of 5000 columns split 256 / 256."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        if mode == "rhs":
            m, n, Ap, Ai, Ax = synth.grid_jacobian()
            k = 1024
            B = synth.grid_rhs(n, k)
            be = shard.HipBackend(m, n, Ap, Ai, device=dev)
            X = shard.solve_many_rhs(be, Ax if rank == 0 else None, B if rank == 0 else None, tol=1e-3, tile=128)
            lo, hi = shard.shard_range(k, world, rank)
            Xs = shard.solve_many_rhs(be, Ax if rank == 0 else None, torch.from_numpy(B[:, lo:hi].copy()).to(dev), tol=1e-3,
                                      resident=True, gather=False)
            cols = [0, 127, 128, 511, 512, 640, 1023]
            if rank == 0:
                assert X.shape == (n, k)
                np.savez(out, X=X[:, cols].cpu().numpy(), Xs=Xs[:, [c for c in cols if c < hi]].cpu().numpy())
            else:
                np.savez(out + ".rank1", Xs=Xs[:, [c - lo for c in cols if c >= lo]].cpu().numpy())
        else:
            from csparse3_amd import csc_hip
            n, nmat = 5000, 512
            ei, ej = synth.spd_grid_pattern(n, seed=5000)
            m, n, Ap, Ai, _ = synth.spd_grid_matrix(n, ei, ej, seed=5000)
            lo, hi = shard.shard_range(nmat, world, rank)
            AX = np.stack([synth.spd_grid_matrix(n, ei, ej, seed=5000 + i)[4] for i in range(lo, hi)])
            B = np.random.default_rng(1).standard_normal((nmat, n, 1))
            X = shard.solve_many_matrices(lambda b: shard.HipBackend(m, n, Ap, Ai, kind=csc_hip.CS3_CHOLESKY, batch=b, device=dev),
                                          AX, B, tol=0.0, local_values=True, total=nmat)
            if rank == 0:
                assert X.shape == (nmat, n, 1)
                np.save(out, X[[0, 63, 64, 255, 256, 300, 511]].cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rhs", "matrices"])
def test_two_ranks_at_the_configurations_own_shapes(gpu, tmp_path, mode):
    """the 2-rank path at config 4's and config 5's OWN shapes, real backend, one GPU, gloo."""
    import scipy.sparse as sp
    out = str(tmp_path / ("x.npz" if mode == "rhs" else "x.npy"))
    mp.spawn(_gpu_worker_full, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    if mode == "rhs":
        m, n, Ap, Ai, Ax = synth.grid_jacobian()
        B = synth.grid_rhs(n, 1024)
        A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
        cols = [0, 127, 128, 511, 512, 640, 1023]
        got = np.load(out)
        X = got["X"]
        assert np.abs(A @ X - B[:, cols]).max() <= 1e-10 * np.abs(B).max()
        Xs = np.concatenate([got["Xs"], np.load(out + ".rank1.npz")["Xs"]], axis=1)      # the resident slabs, left sharded
        assert np.array_equal(Xs, X)
    else:
        n = 5000
        ei, ej = synth.spd_grid_pattern(n, seed=5000)
        B = np.random.default_rng(1).standard_normal((512, n, 1))
        X = np.load(out)
        for j, i in enumerate([0, 63, 64, 255, 256, 300, 511]):
            m, n, Ap, Ai, Ax = synth.spd_grid_matrix(n, ei, ej, seed=5000 + i)
            A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
            A = A + sp.tril(A, -1).T if (sp.triu(A, 1).nnz == 0) else A
            assert np.abs(A @ X[j] - B[i]).max() <= 1e-10 * np.abs(B[i]).max()


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
