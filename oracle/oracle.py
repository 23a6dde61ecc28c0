"""ctypes front end of the CPU oracle (oracle/cs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by csparse3_amd/.  Function names and
argument order follow the reference's flat-array convention
(/root/reference/src/CSparse3/csc_numba.py: loose (m, n, Ap, Ai, Ax), int32
indices, float64 values, tuple returns).

"parity unpinned" for amd / etree / post / counts / lu / chol / lsolve / usolve
(the reference has none of them); the substrate functions
are pinned by tests/golden/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class _Csc(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("nzmax", C.c_int64),
                ("p", _i32p), ("i", _i32p), ("x", _f64p)]


_CscP = C.POINTER(_Csc)


def build(force=False):
    """Compile liboracle.so with gcc (no-op when it is newer than its sources)."""
    srcs = [os.path.join(_HERE, f) for f in ("cs_oracle.c", "cs_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_native = False


def use_native():
    """Switch this module to oracle/_native/liboracle_native.so (-O3 -march=native, built on the spot by
    `make native`): the CPU-BASELINE build for bench.py, compiled on the host it is timed on.  Returns the
    compiler flags used, or None when the build failed and the portable library stays in use."""
    global _lib, _native
    path = os.path.join(_HERE, "_native", "liboracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = None
        _native = True
        lib()
        return "gcc -O3 -march=native"
    except Exception:
        _native = False
        _lib = None
        return None


def lib():
    global _lib
    if _lib is None:
        if _native:
            _lib = C.CDLL(os.path.join(_HERE, "_native", "liboracle_native.so"))
        elif os.environ.get("ORC_LIB_PATH"):                  # sanitizer build of the same source (tools/asan_host.sh)
            _lib = C.CDLL(os.environ["ORC_LIB_PATH"])
        else:
            build()
            _lib = C.CDLL(_LIB_PATH)
        _lib.orc_add.restype = _CscP
        _lib.orc_transpose.restype = _CscP
        _lib.orc_coo_to_csc.restype = _CscP
        _lib.orc_stack_4_by_4.restype = _CscP
        _lib.orc_symperm.restype = _CscP
        _lib.orc_permute.restype = _CscP
        _lib.orc_norm.restype = C.c_double
        _lib.orc_sub_matrix.restype = _CscP
        _lib.orc_find_islands.restype = C.c_int64
        _lib.orc_cumsum.restype = C.c_int64
        _lib.orc_scatter.restype = C.c_int64
        _lib.orc_csc_free.argtypes = [_CscP]
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def _pf(a):
    return None if a is None else a.ctypes.data_as(_f64p)


def _take(ptr, values=True):
    """Copy a heap orc_csc into NumPy-owned arrays and free it."""
    if not ptr:
        raise MemoryError("oracle returned NULL")
    s = ptr.contents
    n = int(s.n)
    Cp = np.ctypeslib.as_array(s.p, shape=(n + 1,)).copy()
    nz = int(Cp[n])
    Ci = np.ctypeslib.as_array(s.i, shape=(max(nz, 1),))[:nz].copy()
    Cx = None
    if values and s.x:
        Cx = np.ctypeslib.as_array(s.x, shape=(max(nz, 1),))[:nz].copy()
    m = int(s.m)
    lib().orc_csc_free(ptr)
    return m, n, Cp, Ci, Cx


I64 = C.c_int64
F64 = C.c_double

# ---------------------------------------------------------------- substrate


def csc_cumsum_i(p, c, n):
    return int(lib().orc_cumsum(_pi(p), _pi(c), I64(n)))


def csc_scatter_f(Ap, Ai, Ax, j, beta, w, x, mark, Ci, nz):
    return int(lib().orc_scatter(_pi(Ap), _pi(Ai), _pf(Ax), I64(j), F64(beta),
                                 _pi(w), _pf(x), I64(mark), _pi(Ci), I64(nz)))


def csc_add_ff(Am, An, Ap, Ai, Ax, Bm, Bn, Bp, Bi, Bx, alpha, beta):
    Ap, Ai, Ax, Bp, Bi, Bx = _i32(Ap), _i32(Ai), _f64(Ax), _i32(Bp), _i32(Bi), _f64(Bx)
    r = lib().orc_add(I64(Am), I64(An), _pi(Ap), _pi(Ai), _pf(Ax),
                      I64(Bm), I64(Bn), _pi(Bp), _pi(Bi), _pf(Bx), F64(alpha), F64(beta))
    return _take(r)


def csc_transpose(m, n, Ap, Ai, Ax):
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    return _take(lib().orc_transpose(I64(m), I64(n), _pi(Ap), _pi(Ai), _pf(Ax)))


def csc_to_csr(m, n, Ap, Ai, Ax, Bp, Bi, Bx):
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    lib().orc_to_csr(I64(m), I64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pi(Bp), _pi(Bi), _pf(Bx))


def csc_mat_vec_ff(m, n, Ap, Ai, Ax, x):
    Ap, Ai, Ax, x = _i32(Ap), _i32(Ai), _f64(Ax), _f64(x)
    y = np.empty(m, dtype=np.float64)
    lib().orc_mat_vec(I64(m), I64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pf(x), _pf(y))
    return y


def csc_mat_vecs(m, n, Ap, Ai, Ax, X):
    Ap, Ai, Ax, X = _i32(Ap), _i32(Ai), _f64(Ax), _f64(X)
    k = X.shape[1]
    Y = np.empty((m, k), dtype=np.float64)
    lib().orc_mat_vecs(I64(m), I64(n), I64(k), _pi(Ap), _pi(Ai), _pf(Ax), _pf(X), _pf(Y))
    return Y


def csc_norm(n, Ap, Ax):
    Ap, Ax = _i32(Ap), _f64(Ax)
    return float(lib().orc_norm(I64(n), _pi(Ap), _pf(Ax)))


def csc_sub_matrix(Am, Anz, Ap, Ai, Ax, rows, cols):
    """-> (nnz, Bp, Bi, Bx) as csc_sub_matrix (csc_numba.py:464-502)."""
    Ap, Ai, Ax, rows, cols = _i32(Ap), _i32(Ai), _f64(Ax), _i32(rows), _i32(cols)
    r = lib().orc_sub_matrix(I64(len(Ap) - 1), _pi(Ap), _pi(Ai), _pf(Ax), _pi(rows), I64(len(rows)), _pi(cols), I64(len(cols)))
    _, _, Bp, Bi, Bx = _take(r)
    return int(Bp[len(cols)]), Bp, Bi, Bx


def find_islands(node_number, indptr, indices):
    """Islands as find_islands lists them (csc_numba.py:744-808), each sorted as CscMat.islands does (csc.py:515-521)."""
    Ap, Ai = _i32(indptr), _i32(indices)
    of = np.zeros(max(node_number, 1), dtype=np.int32)
    cnt = int(lib().orc_find_islands(I64(node_number), _pi(Ap), _pi(Ai), _pi(of)))
    if cnt < 0:
        raise MemoryError("oracle: find_islands")
    of = of[:node_number]
    return [np.flatnonzero(of == c).astype(np.int32) for c in range(cnt)]


def coo_to_csc(m, n, Ti, Tj, Tx, nz):
    Ti, Tj, Tx = _i32(Ti), _i32(Tj), _f64(Tx)
    return _take(lib().orc_coo_to_csc(I64(m), I64(n), _pi(Ti), _pi(Tj), _pf(Tx), I64(nz)))


def csc_stack_4_by_4_ff(am, an, Ai, Ap, Ax, bm, bn, Bi, Bp, Bx,
                        cm, cn, Ci, Cp, Cx, dm, dn, Di, Dp, Dx):
    """Argument order (m, n, indices, indptr, data) and return order
    (m, n, indices, indptr, data) as csc_numba.py:640-720."""
    a = [_i32(Ai), _i32(Ap), _f64(Ax), _i32(Bi), _i32(Bp), _f64(Bx),
         _i32(Ci), _i32(Cp), _f64(Cx), _i32(Di), _i32(Dp), _f64(Dx)]
    r = lib().orc_stack_4_by_4(
        I64(am), I64(an), _pi(a[0]), _pi(a[1]), _pf(a[2]),
        I64(bm), I64(bn), _pi(a[3]), _pi(a[4]), _pf(a[5]),
        I64(cm), I64(cn), _pi(a[6]), _pi(a[7]), _pf(a[8]),
        I64(dm), I64(dn), _pi(a[9]), _pi(a[10]), _pf(a[11]))
    if not r:
        raise AssertionError("incompatible block shapes")
    m, n, Pp, Pi, Px = _take(r)
    return m, n, Pi, Pp, Px

# ------------------------------------------------------- ordering / symbolic


def csc_amd_f(order, m, n, Ap, Ai):
    Ap, Ai = _i32(Ap), _i32(Ai)
    q = np.empty(n, dtype=np.int32)
    st = lib().orc_amd(I64(order), I64(m), I64(n), _pi(Ap), _pi(Ai), _pi(q))
    if st != 0:
        raise ValueError("orc_amd failed: %d" % st)
    return q


def csc_pinv(p):
    p = _i32(p)
    out = np.empty_like(p)
    lib().orc_pinv(_pi(p), _pi(out), I64(len(p)))
    return out


def csc_symperm(n, Ap, Ai, Ax, pinv):
    Ap, Ai = _i32(Ap), _i32(Ai)
    Ax = None if Ax is None else _f64(Ax)
    pinv = None if pinv is None else _i32(pinv)
    return _take(lib().orc_symperm(I64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pi(pinv)),
                 values=Ax is not None)


def csc_permute(m, n, Ap, Ai, Ax, pinv, q):
    Ap, Ai = _i32(Ap), _i32(Ai)
    Ax = None if Ax is None else _f64(Ax)
    pinv = None if pinv is None else _i32(pinv)
    q = None if q is None else _i32(q)
    return _take(lib().orc_permute(I64(m), I64(n), _pi(Ap), _pi(Ai), _pf(Ax),
                                   _pi(pinv), _pi(q)), values=Ax is not None)


def csc_etree_f(n, Ap, Ai):
    """etree of the symmetric matrix whose UPPER triangle is (Ap, Ai)."""
    Ap, Ai = _i32(Ap), _i32(Ai)
    parent = np.empty(n, dtype=np.int32)
    lib().orc_etree(I64(n), _pi(Ap), _pi(Ai), _pi(parent))
    return parent


def csc_post_f(n, parent):
    parent = _i32(parent)
    post = np.empty(n, dtype=np.int32)
    lib().orc_post(I64(n), _pi(parent), _pi(post))
    return post


def csc_counts_f(n, Ap, Ai, parent, post):
    Ap, Ai, parent, post = _i32(Ap), _i32(Ai), _i32(parent), _i32(post)
    cc = np.empty(n, dtype=np.int32)
    lib().orc_counts(I64(n), _pi(Ap), _pi(Ai), _pi(parent), _pi(post), _pi(cc))
    return cc

# -------------------------------------------------------------------- numeric


class SingularMatrix(ArithmeticError):
    pass


class NotPositiveDefinite(ArithmeticError):
    pass


def csc_lu_f(m, n, Ap, Ai, Ax, q=None, tol=1.0):
    """-> (Lp, Li, Lx, Up, Ui, Ux, pinv).  q: column order (None = natural)."""
    assert m == n
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    q = None if q is None else _i32(q)
    pinv = np.empty(n, dtype=np.int32)
    L, U = _CscP(), _CscP()
    st = lib().orc_lu(I64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pi(q), F64(tol),
                      C.byref(L), C.byref(U), _pi(pinv))
    if st != 0:
        raise SingularMatrix("no pivot at step %d" % (-st - 1))
    _, _, Lp, Li, Lx = _take(L)
    _, _, Up, Ui, Ux = _take(U)
    return Lp, Li, Lx, Up, Ui, Ux, pinv


def csc_chol_f(n, Ap, Ai, Ax, pinv, parent, cp):
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    pinv = None if pinv is None else _i32(pinv)
    parent, cp = _i32(parent), _i32(cp)
    L = _CscP()
    st = lib().orc_chol(I64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pi(pinv), _pi(parent),
                        _pi(cp), C.byref(L))
    if st != 0:
        raise NotPositiveDefinite("non-positive pivot at step %d" % (-st - 1))
    _, _, Lp, Li, Lx = _take(L)
    return Lp, Li, Lx


def csc_schol_f(order, n, Ap, Ai):
    """Symbolic Cholesky analysis: -> (pinv, parent, cp, post)."""
    q = csc_amd_f(order, n, n, Ap, Ai)
    pinv = csc_pinv(q)
    _, _, Cp, Ci, _ = csc_symperm(n, Ap, Ai, None, pinv)
    parent = csc_etree_f(n, Cp, Ci)
    post = csc_post_f(n, parent)
    cnt = csc_counts_f(n, Cp, Ci, parent, post)
    cp = np.zeros(n + 1, dtype=np.int32)
    csc_cumsum_i(cp, cnt.copy(), n)
    return pinv, parent, cp, post


def _solve_inplace(fn, n, Gp, Gi, Gx, x):
    Gp, Gi, Gx = _i32(Gp), _i32(Gi), _f64(Gx)
    assert x.dtype == np.float64 and x.flags.c_contiguous
    fn(I64(n), _pi(Gp), _pi(Gi), _pf(Gx), _pf(x))


def csc_lsolve_f(n, Lp, Li, Lx, x):
    _solve_inplace(lib().orc_lsolve, n, Lp, Li, Lx, x)


def csc_usolve_f(n, Up, Ui, Ux, x):
    _solve_inplace(lib().orc_usolve, n, Up, Ui, Ux, x)


def csc_ltsolve_f(n, Lp, Li, Lx, x):
    _solve_inplace(lib().orc_ltsolve, n, Lp, Li, Lx, x)


def csc_utsolve_f(n, Up, Ui, Ux, x):
    _solve_inplace(lib().orc_utsolve, n, Up, Ui, Ux, x)


def csc_lusol_f(order, n, Ap, Ai, Ax, b, tol=1.0):
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    x = np.array(b, dtype=np.float64, copy=True)
    st = lib().orc_lusol(I64(order), I64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pf(x), F64(tol))
    if st != 0:
        raise SingularMatrix("orc_lusol failed: %d" % st)
    return x


def csc_cholsol_f(order, n, Ap, Ai, Ax, b):
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    x = np.array(b, dtype=np.float64, copy=True)
    st = lib().orc_cholsol(I64(order), I64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pf(x))
    if st != 0:
        raise NotPositiveDefinite("orc_cholsol failed: %d" % st)
    return x
