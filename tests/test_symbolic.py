"""Host side of the HIP library (no GPU needed): ordering and symbolic analysis
against the oracle, bit-exact; the C ABI exports every symbol the header declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from csparse3_amd import synth
from helpers import symmetrized

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    m, n, Ap, Ai, Ax, b, xt = synth.toy10()
    yield "toy10", (m, n, Ap, Ai, Ax)
    yield "jacobian118", synth.jacobian_like()
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


def test_numeric_entry_points_fail_loudly_without_a_gpu(hip):
    if hip.device_count() > 0:
        pytest.skip("a GPU is visible here")
    m, n, Ap, Ai, Ax = CASES["toy10"]
    with hip.Factorization(m, n, Ap, Ai) as F:
        with pytest.raises(hip.Cs3Error) as e:
            F.factor(Ax)
        assert "no HIP device" in str(e.value)
