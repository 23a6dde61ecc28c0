"""Device versions of the reference's substrate kernels, through the C ABI.
Bit-exact against (a) the golden vectors = outputs of the reference's own Python kernels
(tests/golden/make_golden.py) and (b) the CPU oracle on larger seeded inputs."""
import os

import numpy as np
import pytest

from helpers import csc_to_scipy

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "substrate.npz"))


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("tag", ["doc", "r1", "r2", "r3"])
def test_transpose_to_csr_norm_match_golden(gpu, tag):
    m, n = int(GOLD[tag + "_m"]), int(GOLD[tag + "_n"])
    Ap, Ai, Ax = GOLD[tag + "_Ap"], GOLD[tag + "_Ai"], GOLD[tag + "_Ax"]
    tn, tm, Tp, Ti, Tx = gpu.csc_transpose(m, n, Ap, Ai, Ax)
    assert (tn, tm) == (n, m)                                   # the reference returns the swapped shape
    assert _same((Tp, Ti, Tx), (GOLD[tag + "_t_p"], GOLD[tag + "_t_i"], GOLD[tag + "_t_x"]))
    Bp = np.zeros(m + 1, dtype=np.int32); Bi = np.empty(int(Ap[n]), dtype=np.int32); Bx = np.empty(int(Ap[n]))
    gpu.csc_to_csr(m, n, Ap, Ai, Ax, Bp, Bi, Bx)
    assert _same((Bp, Bi, Bx), (GOLD[tag + "_csr_p"], GOLD[tag + "_csr_i"], GOLD[tag + "_csr_x"]))
    assert gpu.csc_norm(n, Ap, Ax) == float(GOLD[tag + "_norm"])


@pytest.mark.parametrize("tag", ["r1", "r2", "r3"])
def test_add_and_coo_match_golden(gpu, tag):
    m, n = int(GOLD[tag + "_m"]), int(GOLD[tag + "_n"])
    Ap, Ai, Ax = GOLD[tag + "_Ap"], GOLD[tag + "_Ai"], GOLD[tag + "_Ax"]
    _, _, Cp, Ci, Cx = gpu.csc_add_ff(m, n, Ap, Ai, Ax, m, n, GOLD[tag + "_Bp"], GOLD[tag + "_Bi"], GOLD[tag + "_Bx"], 1.5, -0.25)
    assert _same((Cp, Ci, Cx), (GOLD[tag + "_add_p"], GOLD[tag + "_add_i"], GOLD[tag + "_add_x"]))
    ti = GOLD[tag + "_coo_i"]
    _, _, Kp, Ki, Kx = gpu.coo_to_csc(m, n, ti, GOLD[tag + "_coo_j"], GOLD[tag + "_coo_x"], len(ti))
    assert _same((Kp, Ki, Kx), (GOLD[tag + "_coo_p"], GOLD[tag + "_coo_ci"], GOLD[tag + "_coo_cx"]))


def test_duplicates_and_unsorted_rows_keep_the_reference_order(gpu):
    Ap, Ai, Ax = GOLD["dup_Ap"], GOLD["dup_Ai"], GOLD["dup_Ax"]
    _, _, Tp, Ti, Tx = gpu.csc_transpose(4, 4, Ap, Ai, Ax)
    assert _same((Tp, Ti, Tx), (GOLD["dup_t_p"], GOLD["dup_t_i"], GOLD["dup_t_x"]))
    _, _, Cp, Ci, Cx = gpu.csc_add_ff(4, 4, Ap, Ai, Ax, 4, 4, Ap, Ai, Ax, 2.0, 0.5)
    assert _same((Cp, Ci, Cx), (GOLD["dup_add_p"], GOLD["dup_add_i"], GOLD["dup_add_x"]))
    _, _, Kp, Ki, Kx = gpu.coo_to_csc(4, 4, GOLD["dup_coo_i"], GOLD["dup_coo_j"], GOLD["dup_coo_x"], 9)
    assert _same((Kp, Ki, Kx), (GOLD["dup_coo_p"], GOLD["dup_coo_ci"], GOLD["dup_coo_cx"]))
    assert gpu.csc_norm(4, Ap, Ax) == float(GOLD["dup_norm"])


@pytest.mark.parametrize("tag", ["sub1", "sub2", "sub3"])
def test_sub_matrix_matches_golden(gpu, tag):
    Ap, Ai, Ax = GOLD["r1_Ap"], GOLD["r1_Ai"], GOLD["r1_Ax"]
    nz, Bp, Bi, Bx = gpu.csc_sub_matrix(40, int(Ap[40]), Ap, Ai, Ax, GOLD[tag + "_rows"], GOLD[tag + "_cols"])
    assert nz == int(GOLD[tag + "_nz"])
    assert _same((Bp, Bi, Bx), (GOLD[tag + "_p"], GOLD[tag + "_i"], GOLD[tag + "_x"]))


def test_sub_matrix_with_repeated_rows_is_refused_not_overrun(gpu):
    """Repeated rows / columns make the result larger than nnz(A), which is all the reference allocates
    (csc_numba.py:476-478: it would index past its arrays).  The library reports it instead of writing."""
    Ap, Ai, Ax = GOLD["r1_Ap"], GOLD["r1_Ai"], GOLD["r1_Ax"]
    rows = np.tile(np.arange(40, dtype=np.int32), 3)
    cols = np.tile(np.arange(40, dtype=np.int32), 2)
    with pytest.raises(gpu.Cs3Error) as e:
        gpu.csc_sub_matrix(40, int(Ap[40]), Ap, Ai, Ax, rows, cols)
    assert "room for" in str(e.value)
    # once each, the same call is fine and returns A itself with the reference's row numbering
    nz, Bp, Bi, Bx = gpu.csc_sub_matrix(40, int(Ap[40]), Ap, Ai, Ax, rows[:40], cols[:40])
    assert nz == int(Ap[40])


def test_find_islands_matches_golden(gpu):
    isl = gpu.find_islands(30, GOLD["isl_Ap"], GOLD["isl_Ai"])
    assert len(isl) == int(GOLD["isl_count"]) and [len(x) for x in isl] == list(GOLD["isl_sizes"])
    assert np.array_equal(np.concatenate(isl), GOLD["isl_flat"])


@pytest.mark.parametrize("tag", ["isd0", "isd1", "isd2"])
def test_find_islands_on_unsymmetric_patterns_matches_the_reference(gpu, orc, tag):
    """Direction matters: find_islands follows column -> row edges only (csc_numba.py:768-800), so these patterns give
    other islands than the connected components.  Golden = the reference's own output; oracle and device both match."""
    n = int(GOLD[tag + "_n"])
    Ap, Ai = GOLD[tag + "_Ap"], GOLD[tag + "_Ai"]
    want_flat, want_sizes = GOLD[tag + "_flat"], list(GOLD[tag + "_sizes"])
    isl = gpu.find_islands(n, Ap, Ai)
    assert [len(x) for x in isl] == want_sizes and np.array_equal(np.concatenate(isl), want_flat)
    oi = orc.find_islands(n, Ap, Ai)
    assert [len(x) for x in oi] == want_sizes and np.array_equal(np.concatenate(oi), want_flat)
    # the weakly connected components are something else here
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csg
    G = sp.csc_matrix((np.ones(len(Ai)), Ai, Ap), shape=(n, n))
    assert csg.connected_components(G, directed=False)[0] < len(want_sizes)


def test_find_islands_directed_chain_converges_fast(gpu, orc):
    """A one-directional chain 0 -> 1 -> ... -> n-1 plus a back-pointing tail: pointer doubling keeps the rounds
    logarithmic; result = the reference's semantics (oracle restatement)."""
    n = 20000
    cols = np.arange(n - 1, dtype=np.int32)                  # column v holds row v + 1: edge v -> v + 1
    Ap = np.concatenate([np.arange(n, dtype=np.int32), [n - 1]]).astype(np.int32)
    Ai = (cols + 1).astype(np.int32)
    isl = gpu.find_islands(n, Ap, Ai)
    assert len(isl) == 1 and len(isl[0]) == n
    # reversed edges: v -> v - 1: every node opens its own island except that 0 is reached by nobody smaller
    Ap2 = np.concatenate([[0], np.arange(n, dtype=np.int32)]).astype(np.int32)
    Ai2 = np.arange(n - 1, dtype=np.int32)
    isl2 = gpu.find_islands(n, Ap2, Ai2)
    oi2 = orc.find_islands(n, Ap2, Ai2)
    assert [len(x) for x in isl2] == [len(x) for x in oi2] and np.array_equal(np.concatenate(isl2), np.concatenate(oi2))


def _random_csc(rng, m, n, per_col, dup=False):
    cols = np.repeat(np.arange(n), per_col)
    rows = rng.integers(0, m, size=len(cols))
    if not dup:
        key = np.unique(cols.astype(np.int64) * m + rows)
        cols, rows = (key // m), (key % m)
        order = rng.permutation(len(key))                       # rows of a column in arbitrary order
        order = order[np.argsort(cols[order], kind="stable")]
        cols, rows = cols[order], rows[order]
    Ap = np.zeros(n + 1, dtype=np.int32); np.add.at(Ap, cols + 1, 1); Ap = np.cumsum(Ap).astype(np.int32)
    return Ap, rows.astype(np.int32), rng.standard_normal(len(rows))


@pytest.mark.parametrize("dup", [False, True])
def test_conversions_match_oracle_at_size(gpu, orc, dup):
    rng = np.random.default_rng(31 + dup)
    m, n = 30011, 20000
    Ap, Ai, Ax = _random_csc(rng, m, n, 9, dup)
    Bp, Bi, Bx = _random_csc(rng, m, n, 5, dup)
    assert _same(gpu.csc_transpose(m, n, Ap, Ai, Ax)[2:], orc.csc_transpose(m, n, Ap, Ai, Ax)[2:])
    assert gpu.csc_norm(n, Ap, Ax) == orc.csc_norm(n, Ap, Ax)
    assert _same(gpu.csc_add_ff(m, n, Ap, Ai, Ax, m, n, Bp, Bi, Bx, 0.75, -2.0)[2:],
                 orc.csc_add_ff(m, n, Ap, Ai, Ax, m, n, Bp, Bi, Bx, 0.75, -2.0)[2:])
    cols = np.repeat(np.arange(n, dtype=np.int32), np.diff(Ap))
    perm = rng.permutation(len(Ai))
    assert _same(gpu.coo_to_csc(m, n, Ai[perm], cols[perm], Ax[perm], len(perm))[2:],
                 orc.coo_to_csc(m, n, Ai[perm], cols[perm], Ax[perm], len(perm))[2:])
    # transposing twice sorts the rows of every column and gives the matrix back
    A = csc_to_scipy(m, n, Ap, Ai, Ax)
    _, _, Tp, Ti, Tx = gpu.csc_transpose(m, n, Ap, Ai, Ax)
    _, _, Up, Ui, Ux = gpu.csc_transpose(n, m, Tp, Ti, Tx)
    assert abs(csc_to_scipy(m, n, Up, Ui, Ux) - A).max() == 0.0
    assert all(np.all(np.diff(Ui[Up[j]:Up[j + 1]]) >= 0) for j in range(0, n, 997))


def test_sub_matrix_and_islands_match_oracle_at_size(gpu, orc):
    from csparse3_amd import synth
    rng = np.random.default_rng(5)
    m, n, Ap, Ai, Ax = synth.grid_jacobian(n=6000, seed=6)
    rows = rng.choice(n, size=300, replace=False).astype(np.int32)
    cols = rng.choice(n, size=400, replace=False).astype(np.int32)
    got = gpu.csc_sub_matrix(m, int(Ap[n]), Ap, Ai, Ax, rows, cols)
    want = orc.csc_sub_matrix(m, int(Ap[n]), Ap, Ai, Ax, rows, cols)
    assert got[0] == want[0] and _same(got[1:], want[1:])
    # a network that falls apart into islands: cut the grid at a few places (pattern stays symmetric)
    import scipy.sparse as sp
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).tolil()
    comp = np.minimum(np.arange(n) // 700, 7) + 8 * (rng.random(n) < 0.001)      # 8 slabs + a few isolated nodes
    A = A.tocoo()
    keep = comp[A.row] == comp[A.col]
    P = sp.csc_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=(n, n)); P.sort_indices()
    got = gpu.find_islands(n, P.indptr.astype(np.int32), P.indices.astype(np.int32))
    want = orc.find_islands(n, P.indptr.astype(np.int32), P.indices.astype(np.int32))
    assert len(got) == len(want) >= 8 and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_substrate_rejects_bad_indices(gpu):
    Ap = np.array([0, 1, 2], dtype=np.int32); Ai = np.array([0, 5], dtype=np.int32); Ax = np.ones(2)
    with pytest.raises(gpu.Cs3Error):
        gpu.csc_transpose(2, 2, Ap, Ai, Ax)
    with pytest.raises(gpu.Cs3Error):
        gpu.coo_to_csc(2, 2, np.array([0], dtype=np.int32), np.array([9], dtype=np.int32), np.ones(1), 1)
    # empty inputs
    z = np.zeros(4, dtype=np.int32)
    _, _, Tp, Ti, Tx = gpu.csc_transpose(3, 3, z, np.zeros(0, dtype=np.int32), np.zeros(0))
    assert np.array_equal(Tp, z) and len(Ti) == 0 and gpu.csc_norm(3, z, np.zeros(0)) == 0.0
    assert [list(x) for x in gpu.find_islands(3, z, np.zeros(0, dtype=np.int32))] == [[0], [1], [2]]


# ----------------------------------------------- device-resident assembly -> refactor -> solve ----

def _jacobian_blocks():
    """The config-2 Jacobian cut into the four blocks pack_4_by_4 stacks (csc.py:588-606): [[H, N], [M, L]]."""
    import scipy.sparse as sp
    from csparse3_amd import synth
    m, n, Ap, Ai, Ax = synth.jacobian_config2()
    J = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    k = 219                                                 # pvpq unknowns of the 220-bus case
    blocks = []
    for rs, cs in ((slice(0, k), slice(0, k)), (slice(0, k), slice(k, n)), (slice(k, n), slice(0, k)), (slice(k, n), slice(k, n))):
        Bk = J[rs, cs].tocsc(); Bk.sort_indices()
        blocks.append((Bk.shape[0], Bk.shape[1], Bk.indices.astype(np.int32), Bk.indptr.astype(np.int32), Bk.data.copy()))
    return (m, n, Ap, Ai, Ax), blocks


def test_device_resident_stack_refactor_solve_chain(gpu):
    """Newton-loop shape on resident data: stack the four blocks in HBM (cs3_csc_stack_4_by_4_dev), then per iteration
    restack the new values through the cached map (cs3_restack_values_dev) and refactor + solve (cs3_factor_solve_bx_dev).
    Bit-exact with the golden stacking, with the host-pointer stacking, and with the host-pointer factor + solve."""
    import torch
    dev = torch.device("cuda", 0)
    sh = torch.cuda.current_stream().cuda_stream
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # (i) the reference's own stacking output (tests/golden), on device arrays
    g = lambda k: GOLD["st_" + k]
    am, an, bn, cm = int(g("am")), int(g("an")), int(g("bn")), int(g("cm"))
    blk = [(am, an, g("Ai"), g("Ap"), g("Ax")), (am, bn, g("Bi"), g("Bp"), g("Bx")), (cm, an, g("Ci"), g("Cp"), g("Cx")), (cm, bn, g("Di"), g("Dp"), g("Dx"))]
    dblk = [(m_, n_, int(p_[n_]), T(i_), T(p_), T(x_)) for (m_, n_, i_, p_, x_) in blk]
    nnz = sum(b[2] for b in dblk)
    Pi, Pp, Px = torch.empty(nnz, dtype=torch.int32, device=dev), torch.empty(an + bn + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.float64, device=dev)
    gpu.csc_stack_4_by_4_dev([(m_, n_, z_, i_.data_ptr(), p_.data_ptr(), x_.data_ptr()) for (m_, n_, z_, i_, p_, x_) in dblk],
                             Pi.data_ptr(), Pp.data_ptr(), Px.data_ptr(), 0, sh)
    assert np.array_equal(Pi.cpu().numpy(), g("i")) and np.array_equal(Pp.cpu().numpy(), g("p")) and np.array_equal(Px.cpu().numpy(), g("x"))
    # (ii) the Jacobian of config 2 from its four blocks, then four Newton-style value updates
    (m, n, Ap, Ai, Ax), blocks = _jacobian_blocks()
    dblk = [(m_, n_, int(p_[n_]), T(i_), T(p_), T(x_)) for (m_, n_, i_, p_, x_) in blocks]
    nz = [b[2] for b in dblk]
    nnz = sum(nz)
    Pi, Pp, Px = torch.empty(nnz, dtype=torch.int32, device=dev), torch.empty(n + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.float64, device=dev)
    mp = torch.empty(nnz, dtype=torch.int32, device=dev)
    gpu.csc_stack_4_by_4_dev([(m_, n_, z_, i_.data_ptr(), p_.data_ptr(), x_.data_ptr()) for (m_, n_, z_, i_, p_, x_) in dblk],
                             Pi.data_ptr(), Pp.data_ptr(), Px.data_ptr(), mp.data_ptr(), sh)
    assert np.array_equal(Pp.cpu().numpy(), Ap) and np.array_equal(Pi.cpu().numpy(), Ai) and np.array_equal(Px.cpu().numpy(), Ax)
    hm, hn, hPi, hPp, hPx = gpu.csc_stack_4_by_4_ff(*[v for (m_, n_, i_, p_, x_) in blocks for v in (m_, n_, i_, p_, x_)])
    assert (hm, hn) == (n, n) and np.array_equal(hPi, Ai) and np.array_equal(hPp, Ap) and np.array_equal(hPx, Ax)
    b = np.random.default_rng(1).standard_normal(n)
    d_b, d_x = T(b), torch.empty(n, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(2)
    with gpu.Factorization(n, n, Ap, Ai) as F, gpu.Factorization(n, n, Ap, Ai) as H:
        for it in range(4):
            new = [x_ * (1.0 + 0.02 * rng.uniform(-1.0, 1.0, size=x_.shape)) for (_, _, _, _, x_) in blocks]
            dnew = [T(v) for v in new]                       # (in a power-flow code these are produced on the device)
            gpu.restack_values_dev(nnz, mp.data_ptr(), nz[0], nz[1], nz[2], *[v.data_ptr() for v in dnew], Px.data_ptr(), sh)
            F.factor_solve_bx_dev(Px.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), 1, 1e-3, sh)
            F.factor_status(sh)
            # host-pointer path: stack on the host side of the ABI, factor, solve
            _, _, _, _, hx = gpu.csc_stack_4_by_4_ff(*[v for (blkv, nv) in zip(blocks, new) for v in (blkv[0], blkv[1], blkv[2], blkv[3], nv)])
            assert np.array_equal(Px.cpu().numpy(), hx)
            want = H.factor(hx, 1e-3).solve(b)
            assert np.array_equal(d_x.cpu().numpy(), want)
            A = csc_to_scipy(n, n, Ap, Ai, hx)
            assert np.abs(A @ want - b).max() < 1e-10
