"""ctypes binding of libcsparse3_hip.so -- the MI355X factor/solve kernels behind
the reference's flat-array calling convention.

Every function takes and returns what a kernel module of the reference does
(/root/reference/src/CSparse3/csc_numba.py): scalars int64, index arrays int32,
values float64, matrices as loose (m, n, Ap, Ai, Ax) arguments, results as
tuples of freshly allocated NumPy arrays, or in place for the solves.  The
binding idiom is the reference author's own (research/mkl.py:3-5,33-35):
ctypes.CDLL + ndarray.ctypes.data_as.

There is no CPU fallback: if the shared library is missing or no GPU is
visible the numeric entry points raise.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CS3_LIB_PATH") or os.path.join(_HERE, "libcsparse3_hip.so")   # (override: sanitizer builds, tools/asan_host.sh)

CS3_LU, CS3_CHOLESKY = 0, 1
CS3_ERR_ARG, CS3_ERR_ALLOC, CS3_ERR_HIP, CS3_ERR_PIVOT, CS3_ERR_NOT_SPD, CS3_ERR_STATE = -1, -2, -3, -4, -5, -6      # include/csparse3_amd.h
ORDER_NATURAL, ORDER_AMD, ORDER_GIVEN = 0, 1, 2

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
I64 = C.c_int64


class Cs3Info(C.Structure):
    _fields_ = [("n", C.c_int64), ("nnz_a", C.c_int64), ("nnz_l", C.c_int64), ("nnz_u", C.c_int64),
                ("nsuper", C.c_int64), ("nlevels", C.c_int64), ("max_front", C.c_int64),
                ("max_width", C.c_int64), ("factor_bytes", C.c_int64), ("update_bytes", C.c_int64),
                ("batch", C.c_int64), ("fail_col", C.c_int64), ("flops_factor", C.c_double),
                ("t_order_s", C.c_double), ("t_symbolic_s", C.c_double)]


class Cs3Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cs3 error %d: %s" % (code, msg))
        self.code = code


class SingularMatrix(Cs3Error, ArithmeticError):
    """A static diagonal pivot was zero, non-finite or rejected by tol (CS3_ERR_PIVOT)."""


class NotPositiveDefinite(Cs3Error, ArithmeticError):
    """Cholesky met a non-positive pivot (CS3_ERR_NOT_SPD)."""


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so / libhsa-runtime64.so; the
    stream handles and device pointers that callers hand to this library (torch.cuda.current_stream(),
    Tensor.data_ptr()) only mean something to the runtime that made them, and a second runtime copy in the
    process does not even see the GPU once the first has opened it (measured on the GPU box: loading this
    library first and importing torch afterwards leaves torch with "No HIP GPUs are available").  So when
    torch is installed but not imported yet, its runtime is loaded here, globally, before our library: the
    loader then binds our libamdhip64.so.7 dependency to that copy, and a later `import torch` finds it too.
    Without torch the system ROCm runtime is used.  CS3_SYSTEM_HIP=1 skips this."""
    if "torch" in sys.modules or os.environ.get("CS3_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load the shared library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        _share_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C csparse3_amd/csrc` -- there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.cs3_last_error.restype = C.c_char_p
        vp = C.c_void_p
        L.cs3_analyze.argtypes = [I64, I64, I64, _i32p, _i32p, _i32p, I64, C.POINTER(vp)]
        L.cs3_free.argtypes = [vp]
        L.cs3_get_info.argtypes = [vp, C.POINTER(Cs3Info)]
        L.cs3_get_ordering.argtypes = [vp] + [_i32p] * 6
        L.cs3_get_supernodes.argtypes = [vp] + [_i32p] * 3
        L.cs3_factor.argtypes = [vp, _f64p, C.c_double]
        L.cs3_factor_dev.argtypes = [vp, vp, C.c_double, vp]
        L.cs3_factor_status.argtypes = [vp, vp]
        L.cs3_factor_solve_dev.argtypes = [vp, vp, C.c_double, vp, I64, vp]
        L.cs3_factor_solve_bx_dev.argtypes = [vp, vp, C.c_double, vp, vp, I64, vp]
        for f in (L.cs3_solve, L.cs3_lsolve, L.cs3_usolve):
            f.argtypes = [vp, _f64p, I64]
        for f in (L.cs3_solve_dev, L.cs3_lsolve_dev, L.cs3_usolve_dev):
            f.argtypes = [vp, vp, I64, vp]
        L.cs3_residual_dev.argtypes = [vp, vp, vp, vp, vp, I64, vp]
        L.cs3_matvec_dev.argtypes = [vp, vp, vp, vp, I64, vp]
        L.cs3_refine_dev.argtypes = [vp, vp, vp, vp, I64, I64, C.POINTER(C.c_double), vp]
        L.cs3_export_factor_dev.argtypes = [vp, vp, vp]
        L.cs3_import_factor_dev.argtypes = [vp, vp, vp]
        L.cs3_get_factors.argtypes = [vp, I64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.cs3_amd.argtypes = [I64, I64, I64, _i32p, _i32p, _i32p]
        L.cs3_etree.argtypes = [I64, _i32p, _i32p, _i32p]
        L.cs3_post.argtypes = [I64, _i32p, _i32p]
        L.cs3_counts.argtypes = [I64, _i32p, _i32p, _i32p, _i32p, _i32p]
        L.cs3_csc_lsolve.argtypes = [I64, _i32p, _i32p, _f64p, _f64p, I64]
        L.cs3_csc_usolve.argtypes = [I64, _i32p, _i32p, _f64p, _f64p, I64]
        L.cs3_csc_matvec.argtypes = [I64, I64, _i32p, _i32p, _f64p, _f64p, _f64p, I64]
        L.cs3_csc_stack_4_by_4.argtypes = [I64, I64, _i32p, _i32p, _f64p] * 4 + [_i32p, _i32p, _f64p]
        L.cs3_csc_stack_4_by_4_dev.argtypes = [I64, I64, I64, vp, vp, vp] * 4 + [vp, vp, vp, vp, vp]
        L.cs3_restack_values_dev.argtypes = [I64, vp, I64, I64, I64, vp, vp, vp, vp, vp, vp]
        L.cs3_csc_transpose.argtypes = [I64, I64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.cs3_coo_to_csc.argtypes = [I64, I64, I64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p]
        L.cs3_csc_norm.argtypes = [I64, _i32p, _f64p, C.POINTER(C.c_double)]
        L.cs3_csc_add.argtypes = [I64, I64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, C.c_double, C.c_double, _i32p, _i32p, _f64p]
        L.cs3_csc_sub_matrix.argtypes = [I64, _i32p, _i32p, _f64p, _i32p, I64, _i32p, I64, _i32p, _i32p, _f64p, I64]
        L.cs3_find_islands.argtypes = [I64, _i32p, _i32p, _i32p]
        _lib = L
    return _lib


def _check(rc):
    if rc == 0:
        return
    msg = lib().cs3_last_error().decode("utf-8", "replace")
    if rc == -4:
        raise SingularMatrix(rc, msg)
    if rc == -5:
        raise NotPositiveDefinite(rc, msg)
    raise Cs3Error(rc, msg)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def _pf(a):
    return None if a is None else a.ctypes.data_as(_f64p)


def device_count():
    return int(lib().cs3_device_count())


# ------------------------------------------------------ ordering / symbolic --

def csc_amd_f(order, m, n, Ap, Ai):
    """q = amd(A + A') (order 1) or the natural order (order 0)."""
    Ap, Ai = _i32(Ap), _i32(Ai)
    q = np.empty(n, dtype=np.int32)
    _check(lib().cs3_amd(order, m, n, _pi(Ap), _pi(Ai), _pi(q)))
    return q


def csc_etree_f(n, Ap, Ai):
    """Elimination tree of the symmetric matrix whose UPPER triangle is (Ap, Ai)."""
    Ap, Ai = _i32(Ap), _i32(Ai)
    parent = np.empty(n, dtype=np.int32)
    _check(lib().cs3_etree(n, _pi(Ap), _pi(Ai), _pi(parent)))
    return parent


def csc_post_f(n, parent):
    parent = _i32(parent)
    post = np.empty(n, dtype=np.int32)
    _check(lib().cs3_post(n, _pi(parent), _pi(post)))
    return post


def csc_counts_f(n, Ap, Ai, parent, post):
    Ap, Ai, parent, post = _i32(Ap), _i32(Ai), _i32(parent), _i32(post)
    cc = np.empty(n, dtype=np.int32)
    _check(lib().cs3_counts(n, _pi(Ap), _pi(Ai), _pi(parent), _pi(post), _pi(cc)))
    return cc


# ------------------------------------------------------------------ handle --

class Factorization:
    """Device-resident factorisation handle: analyze once, (re)factor, solve.

    kind: CS3_LU or CS3_CHOLESKY.  order: ORDER_NATURAL / ORDER_AMD, or pass q.
    batch: number of matrices sharing the pattern (values [batch, nnz]).
    """

    def __init__(self, m, n, Ap, Ai, kind=CS3_LU, order=ORDER_AMD, q=None, batch=1):
        assert m == n, "square matrix required"
        self._h = C.c_void_p()
        self.kind = kind
        self.n = int(n)
        self.batch = int(batch)
        Ap, Ai = _i32(Ap), _i32(Ai)
        self.nnz = int(Ap[n])
        qa = None
        if q is not None:
            qa = _i32(q)
            order = ORDER_GIVEN
        _check(lib().cs3_analyze(kind, order, n, _pi(Ap), _pi(Ai), _pi(qa), batch, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().cs3_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def info(self):
        out = Cs3Info()
        _check(lib().cs3_get_info(self._h, C.byref(out)))
        return out

    def ordering(self):
        """-> dict(q_amd, parent, post, colcount, q, pinv), all int32[n]."""
        names = ("q_amd", "parent", "post", "colcount", "q", "pinv")
        arrs = [np.empty(self.n, dtype=np.int32) for _ in names]
        _check(lib().cs3_get_ordering(self._h, *[_pi(a) for a in arrs]))
        return dict(zip(names, arrs))

    def supernodes(self):
        ns = int(self.info.nsuper)
        sn_ptr = np.empty(ns + 1, dtype=np.int32)
        sn_parent = np.empty(ns, dtype=np.int32)
        sn_level = np.empty(ns, dtype=np.int32)
        _check(lib().cs3_get_supernodes(self._h, _pi(sn_ptr), _pi(sn_parent), _pi(sn_level)))
        return sn_ptr, sn_parent, sn_level

    # -- numeric, host arrays
    def factor(self, Ax, tol=0.0):
        Ax = _f64(Ax)
        assert Ax.size >= self.batch * self.nnz
        _check(lib().cs3_factor(self._h, _pf(Ax), tol))
        return self

    def solve(self, b):
        """Solve A x = b.  b: [n], [n, k] or [batch, n, k]; returns a new array."""
        x = np.array(b, dtype=np.float64, order="C", copy=True)
        per = self.batch * self.n
        assert x.size % per == 0, "right-hand side does not match [batch,] n [, k]"
        _check(lib().cs3_solve(self._h, _pf(x), x.size // per))
        return x

    def _sweep(self, fn, x):
        x = np.array(x, dtype=np.float64, order="C", copy=True)
        per = self.batch * self.n
        assert x.size % per == 0
        _check(fn(self._h, _pf(x), x.size // per))
        return x

    def lsolve(self, x):
        """x = L \\ x in pivot order (cs_lsolve on this factorisation's L)."""
        return self._sweep(lib().cs3_lsolve, x)

    def usolve(self, x):
        """x = U \\ x in pivot order (cs_usolve; L' for Cholesky, i.e. cs_ltsolve)."""
        return self._sweep(lib().cs3_usolve, x)

    # -- numeric, device pointers (e.g. torch.Tensor.data_ptr()) on a HIP stream
    def factor_dev(self, ax_ptr, tol=0.0, stream=0):
        _check(lib().cs3_factor_dev(self._h, C.c_void_p(ax_ptr), tol, C.c_void_p(stream)))

    def factor_solve_dev(self, ax_ptr, x_ptr, k=1, tol=0.0, stream=0):
        """(Re)factorise and solve in one call (cs_lusol on resident data); X is overwritten."""
        _check(lib().cs3_factor_solve_dev(self._h, C.c_void_p(ax_ptr), tol, C.c_void_p(x_ptr), k, C.c_void_p(stream)))

    def factor_solve_bx_dev(self, ax_ptr, b_ptr, x_ptr, k=1, tol=0.0, stream=0):
        """The same out of place: right-hand sides at b_ptr stay as they are, solutions go to x_ptr."""
        _check(lib().cs3_factor_solve_bx_dev(self._h, C.c_void_p(ax_ptr), tol, C.c_void_p(b_ptr), C.c_void_p(x_ptr), k, C.c_void_p(stream)))

    def factor_status(self, stream=0):
        _check(lib().cs3_factor_status(self._h, C.c_void_p(stream)))

    def solve_dev(self, x_ptr, k=1, stream=0):
        _check(lib().cs3_solve_dev(self._h, C.c_void_p(x_ptr), k, C.c_void_p(stream)))

    def lsolve_dev(self, x_ptr, k=1, stream=0):
        _check(lib().cs3_lsolve_dev(self._h, C.c_void_p(x_ptr), k, C.c_void_p(stream)))

    def usolve_dev(self, x_ptr, k=1, stream=0):
        _check(lib().cs3_usolve_dev(self._h, C.c_void_p(x_ptr), k, C.c_void_p(stream)))

    def residual_dev(self, ax_ptr, b_ptr, x_ptr, r_ptr, k=1, stream=0):
        """R = B - A X on resident data (A's values at ax_ptr, the analysed pattern); csc_mat_vec_ff's summation order."""
        _check(lib().cs3_residual_dev(self._h, C.c_void_p(ax_ptr), C.c_void_p(b_ptr), C.c_void_p(x_ptr), C.c_void_p(r_ptr), k,
                                      C.c_void_p(stream)))

    def matvec_dev(self, ax_ptr, x_ptr, y_ptr, k=1, stream=0):
        """Y = A X on resident data (the analysed pattern, values at ax_ptr), summed as csc_mat_vec_ff sums."""
        _check(lib().cs3_matvec_dev(self._h, C.c_void_p(ax_ptr), C.c_void_p(x_ptr), C.c_void_p(y_ptr), k, C.c_void_p(stream)))

    def refine_dev(self, ax_ptr, b_ptr, x_ptr, k=1, steps=1, stream=0, want_correction=True):
        """`steps` rounds of x += A \\ (b - A x) with the factors at hand; returns max |dx| of the last round."""
        out = C.c_double(0.0)
        _check(lib().cs3_refine_dev(self._h, C.c_void_p(ax_ptr), C.c_void_p(b_ptr), C.c_void_p(x_ptr), k, steps,
                                    C.byref(out) if want_correction else None, C.c_void_p(stream)))
        return float(out.value)

    def export_factor_dev(self, dst_ptr, stream=0):
        """Copy the factor panels (info.factor_bytes per matrix) into an HBM buffer."""
        _check(lib().cs3_export_factor_dev(self._h, C.c_void_p(dst_ptr), C.c_void_p(stream)))

    def import_factor_dev(self, src_ptr, stream=0):
        """Install factor panels produced by another handle with the same analysis."""
        _check(lib().cs3_import_factor_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(stream)))

    def factors(self, b=0, values=True):
        """-> (Lp, Li, Lx, Up, Ui, Ux) in CSparse form (U parts None for Cholesky)."""
        inf = self.info
        n = self.n
        Lp = np.empty(n + 1, dtype=np.int32)
        Li = np.empty(inf.nnz_l, dtype=np.int32)
        Lx = np.empty(inf.nnz_l, dtype=np.float64) if values else None
        if self.kind == CS3_LU:
            Up = np.empty(n + 1, dtype=np.int32)
            Ui = np.empty(inf.nnz_u, dtype=np.int32)
            Ux = np.empty(inf.nnz_u, dtype=np.float64) if values else None
        else:
            Up = Ui = Ux = None
        _check(lib().cs3_get_factors(self._h, b, _pi(Lp), _pi(Li), _pf(Lx), _pi(Up), _pi(Ui), _pf(Ux)))
        return Lp, Li, Lx, Up, Ui, Ux


# ---------------------------------------------- flat functions, reference style --

def csc_lu_f(m, n, Ap, Ai, Ax, tol=0.0, order=ORDER_AMD, q=None):
    """LU with static diagonal pivots in a fill-reducing order: P A Q = L U.

    -> (Lp, Li, Lx, Up, Ui, Ux, pinv, q).  L unit lower with the diagonal
    first in each column, U upper with the diagonal last (cs_lu's layout).
    """
    with Factorization(m, n, Ap, Ai, CS3_LU, order, q) as F:
        F.factor(Ax, tol)
        Lp, Li, Lx, Up, Ui, Ux = F.factors()
        o = F.ordering()
    return Lp, Li, Lx, Up, Ui, Ux, o["pinv"], o["q"]


def csc_chol_f(m, n, Ap, Ai, Ax, order=ORDER_AMD, q=None):
    """Cholesky P A P' = L L'.  -> (Lp, Li, Lx, pinv)."""
    with Factorization(m, n, Ap, Ai, CS3_CHOLESKY, order, q) as F:
        F.factor(Ax)
        Lp, Li, Lx, _, _, _ = F.factors()
        o = F.ordering()
    return Lp, Li, Lx, o["pinv"]


def _tri(fn, n, Gp, Gi, Gx, x):
    Gp, Gi, Gx = _i32(Gp), _i32(Gi), _f64(Gx)
    assert isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous
    assert x.shape[0] == n
    k = 1 if x.ndim == 1 else x.shape[1]
    _check(fn(n, _pi(Gp), _pi(Gi), _pf(Gx), _pf(x), k))


def csc_lsolve_f(n, Lp, Li, Lx, x):
    """x = L \\ x in place; L lower triangular CSC, diagonal first per column."""
    _tri(lib().cs3_csc_lsolve, n, Lp, Li, Lx, x)


def csc_usolve_f(n, Up, Ui, Ux, x):
    """x = U \\ x in place; U upper triangular CSC, diagonal last per column."""
    _tri(lib().cs3_csc_usolve, n, Up, Ui, Ux, x)


def csc_lusol_f(order, m, n, Ap, Ai, Ax, b, tol=0.0):
    """x = A \\ b by LU (cs_lusol)."""
    with Factorization(m, n, Ap, Ai, CS3_LU, order) as F:
        return F.factor(Ax, tol).solve(b)


def csc_cholsol_f(order, m, n, Ap, Ai, Ax, b):
    """x = A \\ b by Cholesky (cs_cholsol)."""
    with Factorization(m, n, Ap, Ai, CS3_CHOLESKY, order) as F:
        return F.factor(Ax).solve(b)


def csc_mat_vec_ff(m, n, Ap, Ai, Ax, x):
    """y = A x on the device (csc_numba.py:309-328); x [n] or [n, k] row-major."""
    Ap, Ai, Ax, x = _i32(Ap), _i32(Ai), _f64(Ax), _f64(x)
    assert x.shape[0] == n
    k = 1 if x.ndim == 1 else x.shape[1]
    y = np.empty((m,) if x.ndim == 1 else (m, k), dtype=np.float64)
    _check(lib().cs3_csc_matvec(m, n, _pi(Ap), _pi(Ai), _pf(Ax), _pf(x), _pf(y), k))
    return y


def csc_stack_4_by_4_ff(am, an, Ai, Ap, Ax, bm, bn, Bi, Bp, Bx, cm, cn, Ci, Cp, Cx, dm, dn, Di, Dp, Dx):
    """[[A, B], [C, D]] on the device; argument and return order (m, n, indices, indptr, data) as the
    reference's csc_stack_4_by_4_ff (csc_numba.py:640-720)."""
    assert am == bm and cm == dm and an == cn and bn == dn          # csc_numba.py:679-682
    a = [_i32(Ai), _i32(Ap), _f64(Ax), _i32(Bi), _i32(Bp), _f64(Bx), _i32(Ci), _i32(Cp), _f64(Cx), _i32(Di), _i32(Dp), _f64(Dx)]
    nnz = int(a[1][an]) + int(a[4][bn]) + int(a[7][cn]) + int(a[10][dn])
    Pi = np.empty(nnz, dtype=np.int32); Pp = np.empty(an + bn + 1, dtype=np.int32); Px = np.empty(nnz, dtype=np.float64)
    _check(lib().cs3_csc_stack_4_by_4(am, an, _pi(a[0]), _pi(a[1]), _pf(a[2]), bm, bn, _pi(a[3]), _pi(a[4]), _pf(a[5]),
                                      cm, cn, _pi(a[6]), _pi(a[7]), _pf(a[8]), dm, dn, _pi(a[9]), _pi(a[10]), _pf(a[11]),
                                      _pi(Pi), _pi(Pp), _pf(Px)))
    return am + cm, an + bn, Pi, Pp, Px


def csc_stack_4_by_4_dev(blocks, Pi_ptr, Pp_ptr, Px_ptr, map_ptr=0, stream=0):
    """csc_stack_4_by_4_ff on arrays that already live in HBM.  blocks = [(m, n, nnz, indices_ptr, indptr_ptr, data_ptr)] * 4
    for A, B, C, D (device addresses, e.g. torch.Tensor.data_ptr()); outputs are device addresses too.  Asynchronous on
    `stream`; nothing crosses PCIe.  map_ptr (optional, int32[nnz]) receives the restack map for restack_values_dev."""
    args = []
    for (m, n, nnz, pi, pp, px) in blocks:
        args += [m, n, nnz, C.c_void_p(pi), C.c_void_p(pp), C.c_void_p(px)]
    _check(lib().cs3_csc_stack_4_by_4_dev(*args, C.c_void_p(Pi_ptr), C.c_void_p(Pp_ptr), C.c_void_p(Px_ptr),
                                          C.c_void_p(map_ptr), C.c_void_p(stream)))


def restack_values_dev(nnz, map_ptr, nnz_a, nnz_b, nnz_c, ax_ptr, bx_ptr, cx_ptr, dx_ptr, px_ptr, stream=0):
    """Px[p] = (A | B | C | D)[map[p]]: the values-only restack of a Newton iteration (patterns unchanged)."""
    _check(lib().cs3_restack_values_dev(nnz, C.c_void_p(map_ptr), nnz_a, nnz_b, nnz_c, C.c_void_p(ax_ptr), C.c_void_p(bx_ptr),
                                        C.c_void_p(cx_ptr), C.c_void_p(dx_ptr), C.c_void_p(px_ptr), C.c_void_p(stream)))


# ---- format conversions and utilities on the device ------------------------
# Names, argument order and return shapes of the reference's own kernels (csc_numba.py); outputs are
# bit-identical with them, output order included.

def csc_transpose(m, n, Ap, Ai, Ax):
    """C = A' -> (n, m, Cp, Ci, Cx) as csc_transpose (csc_numba.py:400-436): m and n swapped in the result."""
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    nnz = int(Ap[n])
    Cp = np.empty(m + 1, dtype=np.int32); Ci = np.empty(nnz, dtype=np.int32); Cx = np.empty(nnz, dtype=np.float64)
    _check(lib().cs3_csc_transpose(m, n, _pi(Ap), _pi(Ai), _pf(Ax), _pi(Cp), _pi(Ci), _pf(Cx)))
    return n, m, Cp, Ci, Cx


def csc_to_csr(m, n, Ap, Ai, Ax, Bp, Bi, Bx):
    """CSR arrays of A written into the caller's Bp[m + 1], Bi[nnz], Bx[nnz], as csc_to_csr (csc_numba.py:360-397)."""
    Ap, Ai, Ax = _i32(Ap), _i32(Ai), _f64(Ax)
    for a, t in ((Bp, np.int32), (Bi, np.int32), (Bx, np.float64)):
        assert isinstance(a, np.ndarray) and a.dtype == t and a.flags.c_contiguous
    _check(lib().cs3_csc_transpose(m, n, _pi(Ap), _pi(Ai), _pf(Ax), _pi(Bp), _pi(Bi), _pf(Bx)))


def coo_to_csc(m, n, Ti, Tj, Tx, nz):
    """Triplets -> (m, n, Cp, Ci, Cx); duplicates kept, triplet order inside a column (csc_numba.py:331-357)."""
    Ti, Tj, Tx = _i32(Ti), _i32(Tj), _f64(Tx)
    Cp = np.empty(n + 1, dtype=np.int32); Ci = np.empty(nz, dtype=np.int32); Cx = np.empty(nz, dtype=np.float64)
    _check(lib().cs3_coo_to_csc(m, n, nz, _pi(Ti), _pi(Tj), _pf(Tx), _pi(Cp), _pi(Ci), _pf(Cx)))
    return m, n, Cp, Ci, Cx


def csc_norm(n, Ap, Ax):
    """1-norm (csc_numba.py:723-739)."""
    Ap, Ax = _i32(Ap), _f64(Ax)
    out = C.c_double(0.0)
    _check(lib().cs3_csc_norm(n, _pi(Ap), _pf(Ax), C.byref(out)))
    return float(out.value)


def csc_add_ff(Am, An, Ap, Ai, Ax, Bm, Bn, Bp, Bi, Bx, alpha, beta):
    """C = alpha A + beta B -> (m, n, Cp, Ci, Cx) as csc_add_ff (csc_numba.py:183-219)."""
    assert Am == Bm and An == Bn
    Ap, Ai, Ax, Bp, Bi, Bx = _i32(Ap), _i32(Ai), _f64(Ax), _i32(Bp), _i32(Bi), _f64(Bx)
    cap = int(Ap[An]) + int(Bp[Bn])
    Cp = np.empty(An + 1, dtype=np.int32); Ci = np.empty(max(cap, 1), dtype=np.int32); Cx = np.empty(max(cap, 1), dtype=np.float64)
    _check(lib().cs3_csc_add(Am, An, _pi(Ap), _pi(Ai), _pf(Ax), _pi(Bp), _pi(Bi), _pf(Bx), alpha, beta, _pi(Cp), _pi(Ci), _pf(Cx)))
    nz = int(Cp[An])
    return Am, An, Cp, Ci[:nz].copy(), Cx[:nz].copy()


def csc_sub_matrix(Am, Anz, Ap, Ai, Ax, rows, cols):
    """A[rows, cols] -> (nnz, Bp, Bi, Bx) exactly as csc_sub_matrix computes it (csc_numba.py:464-502): the new row
    index is that function's running match counter, not the position of the row in `rows`."""
    Ap, Ai, Ax, rows, cols = _i32(Ap), _i32(Ai), _f64(Ax), _i32(rows), _i32(cols)
    n = len(Ap) - 1
    cap = max(Anz, 1)                                    # what the reference allocates (csc_numba.py:476-478)
    Bp = np.empty(len(cols) + 1, dtype=np.int32); Bi = np.empty(cap, dtype=np.int32); Bx = np.zeros(cap, dtype=np.float64)
    _check(lib().cs3_csc_sub_matrix(n, _pi(Ap), _pi(Ai), _pf(Ax), _pi(rows), len(rows), _pi(cols), len(cols), _pi(Bp), _pi(Bi), _pf(Bx), cap))
    nz = int(Bp[len(cols)])
    return nz, Bp, Bi[:nz].copy(), Bx[:nz].copy()


def find_islands(node_number, indptr, indices):
    """Islands of the pattern as find_islands / CscMat.islands give them (csc_numba.py:744-808, csc.py:515-521):
    a list of islands ordered by their smallest node, each a sorted int32 array."""
    Ap, Ai = _i32(indptr), _i32(indices)
    label = np.empty(node_number, dtype=np.int32)
    _check(lib().cs3_find_islands(node_number, _pi(Ap), _pi(Ai), _pi(label)))
    order = np.argsort(label, kind="stable")
    cuts = np.flatnonzero(np.diff(label[order])) + 1
    return [g.astype(np.int32) for g in np.split(order, cuts)] if node_number else []

