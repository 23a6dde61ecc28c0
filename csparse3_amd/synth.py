"""Seeded synthetic inputs for the five BASELINE.json configurations.

No power-grid data ships with the reference beyond a 5-bus line list
(/root/reference/src/test/test3_lil_matrix.py:18-24), so every input is
synthesised from numpy.random.default_rng(seed) as SURVEY.md section 8d lays out.
All matrices come back in the reference's flat CSC convention:
(m, n, indptr int32, indices int32, data float64), rows sorted, no duplicates.
"""
import numpy as np


def _coo_to_sorted_csc(n, rows, cols, vals):
    """Assemble unique (row, col, val) triplets into row-sorted CSC."""
    order = np.lexsort((rows, cols))
    rows, cols, vals = rows[order], cols[order], vals[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, cols + 1, 1)
    indptr = np.cumsum(indptr)
    return (n, n, indptr.astype(np.int32), rows.astype(np.int32),
            vals.astype(np.float64))


def _graph_to_matrix(n, ei, ej, rng, asym=0.05, shift=1.0, symmetric_values=False):
    """Edges (ei < ej, unique) -> diagonally dominant matrix with that pattern.

    off-diagonal ~ -U(0.1, 1); the two triangles differ by a +-asym relative
    perturbation unless symmetric_values; diag = sum |row| + shift.
    """
    w = rng.uniform(0.1, 1.0, size=len(ei))
    if symmetric_values:
        lo = up = -w
    else:
        lo = -w * (1.0 + asym * rng.uniform(-1.0, 1.0, size=len(ei)))
        up = -w * (1.0 + asym * rng.uniform(-1.0, 1.0, size=len(ei)))
    rowsum = np.zeros(n)
    np.add.at(rowsum, ej, np.abs(lo))    # entry (ej, ei) lies in row ej
    np.add.at(rowsum, ei, np.abs(up))    # entry (ei, ej) lies in row ei
    if symmetric_values:
        diag = rowsum + shift
    else:
        diag = rowsum + shift
    rows = np.concatenate([ej, ei, np.arange(n)])
    cols = np.concatenate([ei, ej, np.arange(n)])
    vals = np.concatenate([lo, up, diag])
    return _coo_to_sorted_csc(n, rows, cols, vals)


def _unique_edges(n, ei, ej):
    lo = np.minimum(ei, ej)
    hi = np.maximum(ei, ej)
    keep = lo != hi
    key = np.unique(lo[keep].astype(np.int64) * n + hi[keep])
    return (key // n).astype(np.int64), (key % n).astype(np.int64)


def toy10(seed=1):
    """Config 1: 10x10, tridiagonal + 3 symmetric off-band pairs, strictly
    diagonally dominant.  -> (m, n, Ap, Ai, Ax, b, x_true), b = A (1..10)."""
    rng = np.random.default_rng(seed)
    n = 10
    ei = np.concatenate([np.arange(n - 1), np.array([0, 2, 4])])
    ej = np.concatenate([np.arange(1, n), np.array([5, 8, 9])])
    ei, ej = _unique_edges(n, ei, ej)
    m, n, Ap, Ai, Ax = _graph_to_matrix(n, ei, ej, rng)
    x_true = np.arange(1.0, n + 1.0)
    b = np.zeros(n)
    for j in range(n):
        for p in range(Ap[j], Ap[j + 1]):
            b[Ai[p]] += Ax[p] * x_true[j]
    return m, n, Ap, Ai, Ax, b, x_true


def _connected_graph(nbus, nedges, rng):
    """Random spanning tree (each node attaches to an earlier one, preferring
    recent nodes, which gives the long radial feeders of a grid) + chords."""
    parent = np.zeros(nbus, dtype=np.int64)
    for v in range(1, nbus):
        lo = max(0, v - 8)
        parent[v] = rng.integers(lo, v) if rng.random() < 0.8 else rng.integers(0, v)
    ei = parent[1:]
    ej = np.arange(1, nbus)
    ei, ej = _unique_edges(nbus, ei, ej)
    while len(ei) < nedges:
        need = nedges - len(ei)
        a = rng.integers(0, nbus, size=2 * need)
        b = rng.integers(0, nbus, size=2 * need)
        ei, ej = _unique_edges(nbus, np.concatenate([ei, a]), np.concatenate([ej, b]))
        if len(ei) > nedges:
            # drop surplus chords, never tree edges (tree edges have ej-parent relation)
            is_tree = parent[ej] == ei
            chord_idx = np.flatnonzero(~is_tree)
            drop = rng.choice(chord_idx, size=len(ei) - nedges, replace=False)
            keep = np.ones(len(ei), dtype=bool)
            keep[drop] = False
            ei, ej = ei[keep], ej[keep]
    return ei, ej


def jacobian_like(nbus=118, nedges=186, n_pv=34, seed=118):
    """Config 2: "118-bus-like" power-flow Jacobian.

    Y-bus of a random connected graph (branch r, x ~ U(0.01, 0.1)); the
    Jacobian J = [[H, N], [M, L]] is laid out exactly as the reference's
    pack_4_by_4 stacks it (/root/reference/src/CSparse3/csc.py:588-606):
    H is (pvpq x pvpq), N (pvpq x pq), M (pq x pvpq), L (pq x pq), bus 0 the
    slack.  Structurally symmetric, values unsymmetric; flat-start
    sensitivities with a small diagonal shift so it is diagonally dominant
    enough for diagonal pivots.  True IEEE-118 data is not available offline.
    -> (m, n, Ap, Ai, Ax)
    """
    rng = np.random.default_rng(seed)
    ei, ej = _connected_graph(nbus, nedges, rng)
    r = rng.uniform(0.01, 0.1, size=len(ei))
    x = rng.uniform(0.01, 0.1, size=len(ei))
    y = 1.0 / (r + 1j * x)
    g, bsus = y.real, y.imag                       # series admittance g + jb, b < 0
    G = np.zeros((nbus, nbus))
    B = np.zeros((nbus, nbus))
    for k in range(len(ei)):
        i, j = ei[k], ej[k]
        G[i, j] -= g[k]; G[j, i] -= g[k]; G[i, i] += g[k]; G[j, j] += g[k]
        B[i, j] -= bsus[k]; B[j, i] -= bsus[k]; B[i, i] += bsus[k]; B[j, j] += bsus[k]
    th = rng.uniform(-0.05, 0.05, size=nbus)       # small angle spread
    v = rng.uniform(0.98, 1.04, size=nbus)
    pv = np.sort(rng.choice(np.arange(1, nbus), size=n_pv, replace=False))
    pq = np.setdiff1d(np.arange(1, nbus), pv)
    pvpq = np.concatenate([pv, pq])
    dth = th[:, None] - th[None, :]
    vv = v[:, None] * v[None, :]
    mask = (G != 0) | (B != 0)
    Hf = vv * (G * np.sin(dth) - B * np.cos(dth)) * mask
    Nf = vv * (G * np.cos(dth) + B * np.sin(dth)) * mask / v[None, :]
    Mf = -vv * (G * np.cos(dth) + B * np.sin(dth)) * mask
    Lf = vv * (G * np.sin(dth) - B * np.cos(dth)) * mask / v[None, :]
    P = (vv * (G * np.cos(dth) + B * np.sin(dth))).sum(axis=1)
    Q = (vv * (G * np.sin(dth) - B * np.cos(dth))).sum(axis=1)
    idx = np.arange(nbus)
    Hf[idx, idx] = -Q - B[idx, idx] * v * v
    Nf[idx, idx] = P / v + G[idx, idx] * v
    Mf[idx, idx] = P - G[idx, idx] * v * v
    Lf[idx, idx] = Q / v - B[idx, idx] * v
    H = Hf[np.ix_(pvpq, pvpq)]
    N = Nf[np.ix_(pvpq, pq)]
    M = Mf[np.ix_(pq, pvpq)]
    L = Lf[np.ix_(pq, pq)]
    J = np.block([[H, N], [M, L]])
    pat = np.block([[mask[np.ix_(pvpq, pvpq)], mask[np.ix_(pvpq, pq)]],
                    [mask[np.ix_(pq, pvpq)], mask[np.ix_(pq, pq)]]])
    n = J.shape[0]
    rows, cols = np.nonzero(pat.T)                 # iterate column-major
    cols, rows = rows, cols
    vals = J[rows, cols]
    return _coo_to_sorted_csc(n, rows.astype(np.int64), cols.astype(np.int64), vals)


def jacobian_config2(seed=118):
    """BASELINE.json configs[1] at the size it states (~400 x 400, ~3k nnz): the same Jacobian builder on a
    220-bus / 340-branch network with 38 PV buses (2 * 219 - 38 = 400 unknowns, 2 986 entries).  The true
    IEEE-118 case is not available offline and would give a 181 x 181 Jacobian (117 + 64 unknowns), so the
    config's own size figures are used; `jacobian_like()` stays as the 118-bus-sized case.
    -> (m, n, Ap, Ai, Ax)"""
    return jacobian_like(nbus=220, nedges=340, n_pv=38, seed=seed)


def grid_graph_edges(n, offsets, chord_frac, rng):
    """1-D chain with extra local ties i <-> i+k for k in offsets, plus
    chord_frac * n random long chords."""
    ei, ej = [], []
    for k in offsets:
        ei.append(np.arange(0, n - k))
        ej.append(np.arange(k, n))
    nch = int(round(chord_frac * n))
    if nch > 0:
        ei.append(rng.integers(0, n, size=nch))
        ej.append(rng.integers(0, n, size=nch))
    return _unique_edges(n, np.concatenate(ei), np.concatenate(ej))


def grid_jacobian(n=50000, seed=50000, offsets=(1, 2, 3, 4), thin=(0.5,),
                  thin_offsets=(9,), chord_frac=0.01):
    """Config 3 / 4: synthetic banded power-grid Jacobian, ~10 nnz per row.

    Radial-plus-meshed graph: chain ties at +-offsets on every node, ties at
    +-thin_offsets on a random fraction `thin` of the nodes, and chord_frac * n
    random long chords; off-diagonal ~ -U(0.1, 1) with a +-5 % asymmetric
    perturbation, diag = sum |row| + 1.  -> (m, n, Ap, Ai, Ax)
    """
    rng = np.random.default_rng(seed)
    ei, ej = grid_graph_edges(n, offsets, chord_frac, rng)
    extra_i, extra_j = [ei], [ej]
    for frac, k in zip(thin, thin_offsets):
        start = np.flatnonzero(rng.random(n - k) < frac)
        extra_i.append(start)
        extra_j.append(start + k)
    ei, ej = _unique_edges(n, np.concatenate(extra_i), np.concatenate(extra_j))
    return _graph_to_matrix(n, ei, ej, rng)


def grid_rhs(n, k=1, seed=1024):
    """Right-hand sides: (n,) for k == 1 else row-major (n, k), the
    reference's multi-vector layout (/root/reference/src/CSparse3/csc.py:409-414)."""
    rng = np.random.default_rng(seed)
    if k == 1:
        return rng.standard_normal(n)
    return rng.standard_normal((n, k))


def spd_grid_pattern(n=5000, seed=5000, width=None, chord_frac=0.01):
    """Config 5 pattern: 2-D-ish grid (row length `width`) + chords.
    -> (ei, ej) with ei < ej."""
    rng = np.random.default_rng(seed)
    if width is None:
        width = int(round(np.sqrt(n)))
    idx = np.arange(n)
    right = idx[(idx % width) != width - 1]
    right = right[right + 1 < n]
    down = idx[idx + width < n]
    nch = int(round(chord_frac * n))
    ei = np.concatenate([right, down, rng.integers(0, n, size=nch)])
    ej = np.concatenate([right + 1, down + width, rng.integers(0, n, size=nch)])
    return _unique_edges(n, ei, ej)


def spd_grid_matrix(n, ei, ej, seed, shift=1e-2, lower_only=False):
    """Config 5 values: weighted graph Laplacian + shift * I on a given pattern
    (same pattern, different values per matrix: contingency-style).
    -> (m, n, Ap, Ai, Ax); full symmetric storage unless lower_only."""
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.1, 1.0, size=len(ei))
    diag = np.full(n, shift)
    np.add.at(diag, ei, w)
    np.add.at(diag, ej, w)
    if lower_only:
        rows = np.concatenate([ej, np.arange(n)])
        cols = np.concatenate([ei, np.arange(n)])
        vals = np.concatenate([-w, diag])
    else:
        rows = np.concatenate([ej, ei, np.arange(n)])
        cols = np.concatenate([ei, ej, np.arange(n)])
        vals = np.concatenate([-w, -w, diag])
    return _coo_to_sorted_csc(n, rows, cols, vals)


def dense_block_matrix(n=700, nd=300, seed=1):
    """Sparse matrix with an embedded nd x nd dense block: its factorisation ends in
    fronts too large for the LDS, which exercises the blocked big-front kernels.
    Structurally symmetric, diagonally dominant.  -> (m, n, Ap, Ai, Ax)"""
    rng = np.random.default_rng(seed)
    nz = 3 * n
    ei = rng.integers(0, n, size=nz)
    ej = rng.integers(0, n, size=nz)
    idx = rng.choice(n, size=nd, replace=False)
    bi, bj = np.meshgrid(idx, idx, indexing="ij")
    ei, ej = _unique_edges(n, np.concatenate([ei, bi.ravel()]), np.concatenate([ej, bj.ravel()]))
    return _graph_to_matrix(n, ei, ej, rng)
