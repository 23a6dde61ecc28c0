"""Multi-GPU sharding of the two workloads of the path that shard.

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  A single factorisation never spans
GPUs; what shards is

  * many right-hand sides of ONE matrix (BASELINE config 4): rank 0 factorises,
    the factor panels are BROADCAST once (one collective of factor_bytes), every
    rank solves its own slab of RHS columns, the solution slabs are GATHERED;
  * many independent matrices (BASELINE config 5): every rank factorises and
    solves its own slice of the batch -- no data-path collective at all, only
    the final gather of the solutions.

The numeric work goes through a `backend` object so that the partitioning and
the collectives can be exercised on CPU with gloo (tests/test_shard.py uses the
oracle as the backend there); `HipBackend` is the real one.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous, balanced [lo, hi) of `total` items for `rank` of `world`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class HipBackend:
    """The MI355X backend: a csc_hip.Factorization plus torch tensors for HBM buffers."""

    def __init__(self, m, n, Ap, Ai, kind=0, batch=1, device=None):
        from csparse3_amd import csc_hip
        self.hip = csc_hip
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.F = csc_hip.Factorization(m, n, Ap, Ai, kind=kind, batch=batch)
        self.n, self.batch = n, batch

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def factor(self, Ax, tol=0.0):
        ax = torch.as_tensor(np.ascontiguousarray(Ax), dtype=torch.float64).to(self.device)
        self.F.factor_dev(ax.data_ptr(), tol, self.stream())
        self.F.factor_status(self.stream())

    def factor_size(self):
        return int(self.F.info.factor_bytes // 8) * self.batch

    def export_factor(self):
        """Factor panels as one flat device tensor (what gets broadcast)."""
        buf = torch.empty(self.factor_size(), dtype=torch.float64, device=self.device)
        self.F.export_factor_dev(buf.data_ptr(), self.stream())
        return buf

    def import_factor(self, buf):
        self.F.import_factor_dev(buf.data_ptr(), self.stream())

    def empty_factor(self):
        return torch.empty(self.factor_size(), dtype=torch.float64, device=self.device)

    def solve(self, B):
        """B: torch tensor [n, k] (or [batch, n, k]) on the device; solved in place, returned."""
        k = B.shape[-1]
        self.F.solve_dev(B.data_ptr(), k, self.stream())
        return B

    def factor_solve(self, Ax, B, tol=0.0):
        """Factorise and solve in one call (cs3_factor_solve_dev: the forward sweep is partly hidden
        behind the factorisation); B as in solve()."""
        ax = torch.as_tensor(np.ascontiguousarray(Ax), dtype=torch.float64).to(self.device)
        self.F.factor_solve_dev(ax.data_ptr(), B.data_ptr(), B.shape[-1], tol, self.stream())
        self.F.factor_status(self.stream())
        return B

    def factor_solve_resident(self, ax_dev, B, tol=0.0):
        """The same with the values already in HBM (a torch tensor): nothing crosses PCIe inside the call."""
        self.F.factor_solve_dev(ax_dev.data_ptr(), B.data_ptr(), B.shape[-1], tol, self.stream())
        self.F.factor_status(self.stream())
        return B

    def to_device(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, dtype=torch.float64)
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.device)


class _Phases:
    """Wall-clock of the phases of a sharded call: `sync()` (device synchronise + barrier, supplied by the caller)
    closes a phase, so the times are max-over-ranks comparable.  Without `timings` nothing is synchronised."""

    def __init__(self, timings, sync):
        self.t, self.sync = timings, sync
        self.t0 = None
        if timings is not None and sync is not None:
            sync()
            import time
            self.clock = time.perf_counter
            self.t0 = self.clock()

    def mark(self, name):
        if self.t0 is None:
            return
        self.sync()
        now = self.clock()
        self.t[name] = self.t.get(name, 0.0) + (now - self.t0)
        self.t0 = now


def solve_many_rhs(backend, Ax, B, tol=0.0, group=None, root=0, timings=None, sync=None, tile=None, resident=False,
                   gather=True):
    """Config 4: A X = B with B [n, k] known on `root` (a NumPy array, or a torch tensor already in HBM);
    returns X [n, k] on root, None elsewhere.

    Collectives: one broadcast of the factor panels, one gather of the solution slabs; the RHS slabs leave the
    root as one batch of point-to-point sends (every xGMI link of the root busy at once).
    `timings` (dict) receives seconds per phase: factor, broadcast, scatter, solve, gather.

    tile (columns): the slabs move and are swept TILE BY TILE -- while a rank sweeps tile t, tile t + 1 of its
      right-hand sides arrives and tile t - 1 of its solutions leaves (one group of point-to-point operations per stage on
      both sides, so the pairs always match); the phases "scatter", "solve", "gather" then collapse into "pipeline".
    resident=True: B is THIS RANK's slab [n, k_r] already in its HBM (right-hand sides generated or assembled in place:
      SURVEY.md section 8e allows it) -- nothing is scattered; with gather=False the solutions stay sharded too and
      every rank gets its own slab back, so the only data-path collective is the broadcast of the factors.
    """
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if world_ready() else (0, 1)
    ph = _Phases(timings, sync)
    if resident:
        return _solve_resident_slabs(backend, Ax, B, tol, group, root, ph, gather, rank, world)
    n, k = B.shape if rank == root else (None, None)
    meta = torch.tensor([n or 0, k or 0], dtype=torch.int64)
    if world > 1:
        meta = meta.to(_comm_device(backend))
        dist.broadcast(meta, src=root, group=group)
    n, k = int(meta[0]), int(meta[1])
    # ---- factor once, broadcast the panels
    fac = _factor_and_broadcast(backend, Ax, tol, group, root, ph, rank, world)
    if tile is not None and world > 1:
        return _pipelined_slabs(backend, B, n, k, int(tile), fac.device, group, root, ph, rank, world)
    # ---- scatter the RHS columns as contiguous slabs [n, k_r]
    lo, hi = shard_range(k, world, rank)
    if world > 1:
        if rank == root:
            Bd = backend.to_device(B)
            slabs = [Bd[:, slice(*shard_range(k, world, r))].contiguous() for r in range(world)]
        else:
            slabs = None
        mine = torch.empty((n, hi - lo), dtype=torch.float64, device=fac.device)
        _scatter_uneven(mine, slabs, root, group)
    else:
        mine = backend.to_device(B).contiguous()
        if isinstance(B, torch.Tensor) and mine.data_ptr() == B.data_ptr():
            mine = mine.clone()                      # the sweeps run in place; the caller keeps its right-hand sides
    ph.mark("scatter")
    if hi > lo:
        backend.solve(mine)
    ph.mark("solve")
    # ---- gather the solution slabs
    if world == 1:
        ph.mark("gather")
        return mine
    out = _gather_uneven(mine, [(n, shard_range(k, world, r)[1] - shard_range(k, world, r)[0])
                                for r in range(world)], root, group)
    X = torch.cat(out, dim=1) if rank == root else None
    ph.mark("gather")
    return X


def _factor_and_broadcast(backend, Ax, tol, group, root, ph, rank, world):
    if rank == root:
        backend.factor(Ax, tol)
        ph.mark("factor")
        fac = backend.export_factor()
    else:
        ph.mark("factor")
        fac = backend.empty_factor()
    if world > 1:
        _settle(fac, group)
        dist.broadcast(fac, src=root, group=group)
        _settle(fac, group)
        if rank != root:
            backend.import_factor(fac)
    ph.mark("broadcast")
    return fac


def _solve_resident_slabs(backend, Ax, B, tol, group, root, ph, gather, rank, world):
    """solve_many_rhs with the right-hand sides resident per rank: factor on root, broadcast, sweep the local slab."""
    _factor_and_broadcast(backend, Ax, tol, group, root, ph, rank, world)
    mine = backend.to_device(B).contiguous()
    if isinstance(B, torch.Tensor) and mine.data_ptr() == B.data_ptr():
        mine = mine.clone()
    ph.mark("scatter")
    if mine.shape[-1] > 0:
        backend.solve(mine)
    ph.mark("solve")
    if world == 1 or not gather:
        ph.mark("gather")
        return mine
    widths = torch.tensor([mine.shape[1]], dtype=torch.int64, device=_comm_device(backend))
    allw = [torch.zeros_like(widths) for _ in range(world)]
    dist.all_gather(allw, widths, group=group)
    out = _gather_uneven(mine, [(mine.shape[0], int(w[0])) for w in allw], root, group)
    X = torch.cat(out, dim=1) if rank == root else None
    ph.mark("gather")
    return X


def _pipelined_slabs(backend, B, n, k, tile, dev, group, root, ph, rank, world):
    """Scatter -> sweep -> gather of solve_many_rhs, tile by tile.  Stage s = one group of point-to-point operations
    {right-hand sides of tile s + 1 root -> peers, solutions of tile s - 1 peers -> root} posted before the sweep of tile s
    and waited for after it: with RCCL the group runs on the communicator's stream beside the sweep.  Every rank derives
    the same tile shapes from (k, world, tile), so both sides of a pair always post the same stages."""
    ranges = [shard_range(k, world, r) for r in range(world)]
    tiles = [[(lo + t, min(lo + t + tile, hi)) for t in range(0, hi - lo, tile)] for lo, hi in ranges]
    nstage = max(len(t) for t in tiles)
    Bd = backend.to_device(B) if rank == root else None
    mine = tiles[rank]
    bufs = [torch.empty((n, c1 - c0), dtype=torch.float64, device=dev) for c0, c1 in mine]
    # root: solutions of the peers' tiles as they arrive
    xbuf = {(r, t): torch.empty((n, c1 - c0), dtype=torch.float64, device=dev)
            for r in range(world) if rank == root and r != root for t, (c0, c1) in enumerate(tiles[r])}
    keep = []                                       # send buffers stay alive until their stage has been waited for

    def stage_ops(s):
        ops = []
        if rank == root:
            for r in range(world):
                if r == root:
                    continue
                if s + 1 < len(tiles[r]):           # right-hand sides of tile s + 1 of rank r
                    c0, c1 = tiles[r][s + 1]
                    t_ = Bd[:, c0:c1].contiguous()
                    keep.append(t_)
                    ops.append(dist.P2POp(dist.isend, t_, r, group))
                if 0 <= s - 1 < len(tiles[r]):      # solutions of its tile s - 1
                    ops.append(dist.P2POp(dist.irecv, xbuf[(r, s - 1)], r, group))
        else:
            if s + 1 < len(mine):
                ops.append(dist.P2POp(dist.irecv, bufs[s + 1], root, group))
            if 0 <= s - 1 < len(mine):
                ops.append(dist.P2POp(dist.isend, bufs[s - 1], root, group))
        return ops

    def post(s):
        ops = stage_ops(s)
        _settle(bufs[0] if bufs else None, group)
        return dist.batch_isend_irecv(ops) if ops else []

    def wait(reqs):
        for q in reqs:
            q.wait()
        _settle(bufs[0] if bufs else None, group)

    wait(post(-1))                                  # tile 0 of every peer
    for s in range(nstage):
        if rank == root and s < len(mine):
            c0, c1 = mine[s]
            bufs[s].copy_(Bd[:, c0:c1])
        reqs = post(s)
        if s < len(mine):
            backend.solve(bufs[s])
        wait(reqs)
    wait(post(nstage))                              # the last tile's solutions
    ph.mark("pipeline")
    if rank != root:
        return None
    cols = []
    for r in range(world):
        cols += bufs if r == root else [xbuf[(r, t)] for t in range(len(tiles[r]))]
    return torch.cat(cols, dim=1) if cols else torch.empty((n, 0), dtype=torch.float64, device=dev)


def world_ready():
    return dist.is_available() and dist.is_initialized()


def solve_many_matrices(make_backend, AX, B, tol=0.0, group=None, root=0, timings=None, sync=None,
                        local_values=False, total=None):
    """Config 5: matrices sharing one pattern, values AX [nmat, nnz], right-hand sides B [nmat, n, k] (known on
    every rank: synthetic inputs are generated from seeds); each rank factorises and solves its contiguous
    slice -- no data-path collective -- and the solutions are gathered.  Returns X [nmat, n, k] on root, None
    elsewhere.  `make_backend(batch)` builds a backend for `batch` matrices.
    local_values: AX holds only THIS rank's slice (rows lo..hi of the batch of `total` matrices).
    `timings` (dict) receives seconds per phase: upload, factor_solve, gather."""
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if world_ready() else (0, 1)
    ph = _Phases(timings, sync)
    nmat = total if local_values else AX.shape[0]
    lo, hi = shard_range(nmat, world, rank)
    X = None
    be = None
    if hi > lo:
        be = make_backend(hi - lo)
        Bd = be.to_device(B[lo:hi]).contiguous()
        vals = AX if local_values else AX[lo:hi]
        if hasattr(be, "to_device") and hasattr(be, "factor_solve_resident"):
            ax_dev = be.to_device(vals)
            ph.mark("upload")
            X = be.factor_solve_resident(ax_dev, Bd, tol)
        elif hasattr(be, "factor_solve"):
            ph.mark("upload")
            X = be.factor_solve(vals, Bd, tol)
        else:
            ph.mark("upload")
            be.factor(vals, tol)
            X = be.solve(Bd)
    else:
        ph.mark("upload")
    ph.mark("factor_solve")
    if world == 1:
        ph.mark("gather")
        return X
    shapes = [(shard_range(nmat, world, r)[1] - shard_range(nmat, world, r)[0],) + tuple(B.shape[1:])
              for r in range(world)]
    dev = X.device if X is not None else _comm_device(be)
    if X is None:
        X = torch.empty((0,) + tuple(B.shape[1:]), dtype=torch.float64, device=dev)
    out = _gather_uneven(X, shapes, root, group)
    res = torch.cat(out, dim=0) if rank == root else None
    ph.mark("gather")
    return res


def _comm_device(backend):
    dev = getattr(backend, "device", None)
    return dev if dev is not None else torch.device("cpu")


def _settle(t, group):
    """RCCL orders a collective with the caller's stream on both sides.  Other backends (gloo: the one-GPU rehearsal)
    move device tensors through host copies on streams of their own, which neither wait for kernels queued on the
    caller's stream nor are waited for by it -- a slab could leave before its sweep had run, or land after it.  So
    around every transfer of a device tensor they get a device-wide synchronisation; RCCL gets none."""
    if t is not None and t.is_cuda and dist.get_backend(group) != "nccl":
        torch.cuda.synchronize(t.device)


def _scatter_uneven(recv, slabs, root, group):
    """Fan-out from root as ONE batch of point-to-point sends (slabs may differ in width, so no scatter collective;
    batched, RCCL drives all of the root's xGMI links at once instead of one peer after the other)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ops = []
    if rank == root:
        for r in range(world):
            if r == root:
                recv.copy_(slabs[r])
            elif slabs[r].numel() > 0:
                ops.append(dist.P2POp(dist.isend, slabs[r], r, group))
    elif recv.numel() > 0:
        ops.append(dist.P2POp(dist.irecv, recv, root, group))
    _settle(recv, group)
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    _settle(recv, group)


def _gather_uneven(send, shapes, root, group):
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ops, out = [], None
    if rank == root:
        out = []
        for r in range(world):
            if r == root:
                out.append(send)
                continue
            t = torch.empty(shapes[r], dtype=send.dtype, device=send.device)
            out.append(t)
            if t.numel() > 0:
                ops.append(dist.P2POp(dist.irecv, t, r, group))
    elif send.numel() > 0:
        ops.append(dist.P2POp(dist.isend, send, root, group))
    _settle(send, group)
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    _settle(send, group)
    return out
