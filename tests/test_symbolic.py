"""Host side of the HIP library (no GPU needed): ordering and symbolic analysis
against the oracle, bit-exact; the C ABI exports every symbol the header declares.

Every test here runs TWICE: unmarked in the CPU suite (`-m "not gpu"`) and with the `gpu` marker in the
driver's GPU run (`-m gpu`), so that the AMD / etree / postorder / column-count / pattern parity is part of
the recorded GPU evidence as well (the code under test is the same host C++ either way)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from csparse3_amd import synth
from helpers import symmetrized

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True, params=[pytest.param("cpu-suite"), pytest.param("gpu-suite", marks=pytest.mark.gpu)])
def suite(request):
    return request.param


def _cases():
    m, n, Ap, Ai, Ax, b, xt = synth.toy10()
    yield "toy10", (m, n, Ap, Ai, Ax)
    yield "jacobian118", synth.jacobian_like()
    yield "config2", synth.jacobian_config2()
    yield "grid3k", synth.grid_jacobian(n=3000, seed=5)
    yield "grid50k", synth.grid_jacobian()
    yield "denseblock", synth.dense_block_matrix(n=400, nd=150, seed=2)
    ei, ej = synth.spd_grid_pattern(2000, seed=8)
    yield "spd2k", synth.spd_grid_matrix(2000, ei, ej, seed=9)


CASES = dict(_cases())


def test_header_symbols_are_exported(hip):
    lib = hip.lib()
    header = open(os.path.join(ROOT, "include", "csparse3_amd.h")).read()
    names = sorted(set(re.findall(r"\b(cs3_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), "libcsparse3_hip.so does not export " + name
    assert lib.cs3_version() >= 100


@pytest.mark.parametrize("name", list(CASES))
def test_amd_bit_exact(hip, orc, name):
    m, n, Ap, Ai, Ax = CASES[name]
    assert np.array_equal(hip.csc_amd_f(1, m, n, Ap, Ai), orc.csc_amd_f(1, m, n, Ap, Ai))
    assert np.array_equal(hip.csc_amd_f(0, m, n, Ap, Ai), np.arange(n))


def test_amd_ignores_row_order_inside_columns_like_the_oracle(hip, orc):
    """CscMat rows are not guaranteed sorted (csc_numba.py:334-335); both sides see the same lists."""
    m, n, Ap, Ai, Ax = synth.grid_jacobian(n=500, seed=1)
    rng = np.random.default_rng(0)
    Ai2 = Ai.copy()
    for j in range(n):
        rng.shuffle(Ai2[Ap[j]:Ap[j + 1]])
    assert np.array_equal(hip.csc_amd_f(1, m, n, Ap, Ai2), orc.csc_amd_f(1, m, n, Ap, Ai2))


@pytest.mark.parametrize("name", list(CASES))
def test_etree_post_counts_bit_exact(hip, orc, name):
    m, n, Ap, Ai, Ax = CASES[name]
    with hip.Factorization(m, n, Ap, Ai) as F:
        o = F.ordering()
    q = orc.csc_amd_f(1, n, n, Ap, Ai)
    assert np.array_equal(o["q_amd"], q)
    Sp, Si = symmetrized(n, Ap, Ai)
    _, _, Cp, Ci, _ = orc.csc_symperm(n, Sp, Si, None, orc.csc_pinv(q))
    parent = orc.csc_etree_f(n, Cp, Ci)
    post = orc.csc_post_f(n, parent)
    counts = orc.csc_counts_f(n, Cp, Ci, parent, post)
    assert np.array_equal(o["parent"], parent)
    assert np.array_equal(o["post"], post)
    assert np.array_equal(o["colcount"], counts)
    # the flat entry points
    assert np.array_equal(hip.csc_etree_f(n, Cp, Ci), parent)
    assert np.array_equal(hip.csc_post_f(n, parent), post)
    assert np.array_equal(hip.csc_counts_f(n, Cp, Ci, parent, post), counts)
    # the pivot order is the fill-reducing order composed with a postorder of that tree
    assert sorted(o["q"].tolist()) == list(range(n))
    assert np.array_equal(o["pinv"][o["q"]], np.arange(n))


@pytest.mark.parametrize("name", ["toy10", "jacobian118", "grid3k", "denseblock"])
def test_factor_pattern_equals_oracle_lu_pattern(hip, orc, name):
    m, n, Ap, Ai, Ax = CASES[name]
    with hip.Factorization(m, n, Ap, Ai) as F:
        o = F.ordering()
        Lp, Li, _, Up, Ui, _ = F.factors(values=False)
        inf = F.info
    oLp, oLi, _, oUp, oUi, _, opinv = orc.csc_lu_f(n, n, Ap, Ai, Ax, o["q"], 1e-3)
    assert np.array_equal(opinv, o["pinv"])
    assert np.array_equal(Lp, oLp) and np.array_equal(Up, oUp)
    for j in range(n):
        assert np.array_equal(Li[Lp[j]:Lp[j + 1]], np.sort(oLi[oLp[j]:oLp[j + 1]]))
        assert np.array_equal(Ui[Up[j]:Up[j + 1]], np.sort(oUi[oUp[j]:oUp[j + 1]]))
    assert inf.nnz_l == Lp[n] and inf.nnz_u == Up[n]


def test_supernode_partition_is_consistent(hip):
    m, n, Ap, Ai, Ax = CASES["grid3k"]
    with hip.Factorization(m, n, Ap, Ai) as F:
        sn_ptr, sn_parent, sn_level = F.supernodes()
        inf = F.info
    assert sn_ptr[0] == 0 and sn_ptr[-1] == n and (np.diff(sn_ptr) > 0).all()
    ns = len(sn_parent)
    assert inf.nsuper == ns and inf.nlevels == sn_level.max() + 1
    for s in range(ns):
        p = sn_parent[s]
        assert p == -1 or (p > s and sn_level[p] > sn_level[s])


def test_given_order_and_natural_order(hip, orc):
    m, n, Ap, Ai, Ax = CASES["jacobian118"]
    q = np.random.default_rng(1).permutation(n).astype(np.int32)
    with hip.Factorization(m, n, Ap, Ai, q=q) as F:
        o = F.ordering()
    assert np.array_equal(o["q_amd"], q)
    with hip.Factorization(m, n, Ap, Ai, order=hip.ORDER_NATURAL) as F:
        assert np.array_equal(F.ordering()["q_amd"], np.arange(n))


def test_bad_input_is_rejected(hip):
    m, n, Ap, Ai, Ax = CASES["toy10"]
    with pytest.raises(hip.Cs3Error):
        hip.Factorization(m, n, Ap, Ai, q=np.zeros(n, dtype=np.int32))       # not a permutation
    bad = Ai.copy(); bad[3] = n + 5
    with pytest.raises(hip.Cs3Error):
        hip.Factorization(m, n, Ap, bad)                                     # row index out of range
    dup = Ai.copy(); dup[Ap[2] + 1] = dup[Ap[2]]
    with pytest.raises(hip.Cs3Error):
        hip.Factorization(m, n, Ap, dup)                                     # duplicate entry
    with pytest.raises(AssertionError):
        hip.Factorization(m, n + 1, Ap, Ai)                                  # not square


def test_stand_alone_symbolic_entry_points_validate_their_pattern(hip):
    """cs3_amd / cs3_etree / cs3_counts index arrays of length n by Ai: bad patterns must come back as errors."""
    m, n, Ap, Ai, Ax = CASES["toy10"]
    bad_row = Ai.copy(); bad_row[2] = n + 3
    bad_ptr = Ap.copy(); bad_ptr[3] = bad_ptr[2] - 1
    for ap, ai in ((Ap, bad_row), (bad_ptr, Ai)):
        with pytest.raises(hip.Cs3Error):
            hip.csc_amd_f(1, m, n, ap, ai)
        with pytest.raises(hip.Cs3Error):
            hip.csc_etree_f(n, ap, ai)
    parent = hip.csc_etree_f(n, Ap, Ai)
    bad_parent = parent.copy(); bad_parent[0] = n + 1
    with pytest.raises(hip.Cs3Error):
        hip.csc_post_f(n, bad_parent)
    with pytest.raises(hip.Cs3Error):
        hip.csc_counts_f(n, Ap, Ai, bad_parent, np.arange(n, dtype=np.int32))
    # null row indices with a non-empty pattern
    assert hip.lib().cs3_amd(1, n, n, Ap.ctypes.data_as(C.POINTER(C.c_int32)), None,
                             np.empty(n, np.int32).ctypes.data_as(C.POINTER(C.c_int32))) == -1


def test_amd_fill_quality(hip):
    """Oracle-independent check of the ordering: a permutation whose fill is in the range of SciPy's SuperLU
    minimum-degree orderings on the same pattern (AMD is not unique, so there is no bit-exact external answer;
    SURVEY.md section 8c.3)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    m, n, Ap, Ai, Ax = CASES["grid3k"]
    q = hip.csc_amd_f(1, m, n, Ap, Ai)
    assert sorted(q.tolist()) == list(range(n))
    with hip.Factorization(m, n, Ap, Ai) as F:
        nnz_l = int(F.info.nnz_l)
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    lu = spl.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    assert nnz_l <= 1.15 * lu.L.nnz, "AMD fill %d vs SuperLU MMD_AT_PLUS_A %d" % (nnz_l, lu.L.nnz)
    with hip.Factorization(m, n, Ap, Ai, order=hip.ORDER_NATURAL) as F:
        assert nnz_l < int(F.info.nnz_l)                  # and it beats no ordering at all


@pytest.mark.parametrize("which", ["config3_jacobian_50k", "config5_spd_5k"])
def test_amd_quality_gate_at_the_configurations_own_sizes(hip, which):
    """The same oracle-independent gate at the sizes BASELINE.json quotes: cs3_amd returns a
    permutation, the same one on a second run, and its fill stays within 5 % of SuperLU's MMD_AT_PLUS_A on the same
    pattern (the 50 000-column Jacobian of configs[2..3], the 5 000-column SPD pattern of configs[4])."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    if which == "config3_jacobian_50k":
        m, n, Ap, Ai, Ax = CASES["grid50k"]
        kind = hip.CS3_LU
    else:
        ei, ej = synth.spd_grid_pattern(5000, seed=5000)
        m, n, Ap, Ai, Ax = synth.spd_grid_matrix(5000, ei, ej, seed=5000)
        kind = hip.CS3_CHOLESKY
    q = hip.csc_amd_f(1, m, n, Ap, Ai)
    assert np.array_equal(np.sort(q), np.arange(n)), "not a permutation"
    assert np.array_equal(q, hip.csc_amd_f(1, m, n, Ap, Ai)), "two runs, two orders"
    with hip.Factorization(m, n, Ap, Ai, kind=kind) as F:
        nnz_l = int(F.info.nnz_l)
        assert np.array_equal(F.ordering()["q_amd"], q)
    A = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n))
    if kind == hip.CS3_CHOLESKY:                     # one triangle is stored: SuperLU gets the full symmetric matrix
        A = (A + A.T - sp.diags(A.diagonal())).tocsc()
    lu = spl.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    assert nnz_l <= 1.05 * lu.L.nnz, "AMD fill %d vs SuperLU MMD_AT_PLUS_A %d" % (nnz_l, lu.L.nnz)


def test_cscmat_reanalyses_when_the_pattern_changes(hip):
    """CscMat caches its symbolic analysis; the cache key includes a digest of indptr / indices, so changing
    the pattern in place (or reassigning the arrays) between two factorisations cannot reuse a stale one."""
    from csparse3_amd.csc import CscMat
    m, n, Ap, Ai, Ax = CASES["jacobian118"]
    A = CscMat(m, n, indptr=Ap.copy(), indices=Ai.copy(), data=Ax.copy())
    F1 = A._analysis(hip.CS3_LU, hip.ORDER_AMD, None)
    assert A._analysis(hip.CS3_LU, hip.ORDER_AMD, None) is F1            # unchanged pattern: cached
    A.data[:] *= 2.0
    assert A._analysis(hip.CS3_LU, hip.ORDER_AMD, None) is F1            # values do not matter
    nnz1 = int(F1.info.nnz_a)
    m2, n2, Ap2, Ai2, Ax2 = synth.jacobian_like(seed=7)                  # same size, another pattern
    assert n2 == n
    A.indptr, A.indices, A.data = Ap2, Ai2, Ax2
    F2 = A._analysis(hip.CS3_LU, hip.ORDER_AMD, None)
    assert F2 is not F1 and int(F2.info.nnz_a) == int(Ap2[n]) and (nnz1 != int(Ap2[n]) or not np.array_equal(Ai, Ai2))
    col = int(np.argmax(np.diff(A.indptr) >= 3))
    p = A.indptr[col]
    free = np.setdiff1d(np.arange(n), A.indices[A.indptr[col]:A.indptr[col + 1]])
    A.indices[p if A.indices[p] != col else p + 1] = free[0]            # in-place edit of one row index
    F3 = A._analysis(hip.CS3_LU, hip.ORDER_AMD, None)
    assert F3 is not F2


def test_numeric_entry_points_fail_loudly_without_a_gpu(hip):
    if hip.device_count() > 0:
        pytest.skip("a GPU is visible here")
    m, n, Ap, Ai, Ax = CASES["toy10"]
    with hip.Factorization(m, n, Ap, Ai) as F:
        with pytest.raises(hip.Cs3Error) as e:
            F.factor(Ax)
        assert "no HIP device" in str(e.value)
