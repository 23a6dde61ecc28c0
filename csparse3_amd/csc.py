"""Host-side mirror of the reference's matrix class for the factor/solve path.

`CscMat` keeps the reference's constructor, attributes and accessors
(/root/reference/src/CSparse3/csc.py:44-141, 459-538: m, n, indptr, indices,
data, nzmax, get_nnz, shape, copy, todense, t) and adds the entry points the
reference lacks at this snapshot: lu / chol / solve on the
object, lusol / cholsol / lsolve / usolve as module functions.  Kernel names
are star-imported from the backend module exactly as csc.py:34-41 does, so
`from csparse3_amd.csc import *` exposes csc_lu_f, csc_lsolve_f, ... next to
the class.
"""
import hashlib

import numpy as np

from csparse3_amd import __config__

if __config__.BACKEND != "hip":
    raise ImportError("csparse3_amd has one numeric backend, 'hip' (got %r); there is no CPU fallback"
                      % __config__.BACKEND)
from csparse3_amd.csc_hip import *                       # noqa: F401,F403  (kernel names, reference style)
from csparse3_amd import csc_hip as _k


class CscMat:
    """Matrix in compressed-column form (same fields as the reference's CscMat)."""

    def __init__(self, m=0, n=0, nz_max=0, indptr=None, indices=None, data=None, zeros=False):
        self.m, self.n = m, n
        if indptr is None:
            self.nzmax = max(nz_max, 1)
            alloc = np.zeros if zeros else np.empty
            self.indptr = alloc(n + 1, dtype=np.int32)
            self.indices = alloc(nz_max, dtype=np.int32)
            self.data = alloc(nz_max, dtype=np.float64)
        else:
            self.indptr, self.indices, self.data = indptr, indices, data
            self.nzmax = len(self.data)
        self._factorization = None

    # ---- what the reference class already offers, kept for drop-in use
    def get_nnz(self):
        return self.indptr[self.n]                       # csc.py:480: nnz is indptr[n], not len(data)

    @property
    def shape(self):
        return self.m, self.n

    def copy(self):
        return CscMat(self.m, self.n, indptr=self.indptr.copy(), indices=self.indices.copy(),
                      data=self.data.copy())

    def todense(self):
        val = np.zeros((self.m, self.n), dtype=np.float64)
        for j in range(self.n):
            for p in range(self.indptr[j], self.indptr[j + 1]):
                val[self.indices[p], j] = self.data[p]
        return val

    # ---- operators of the reference class whose kernels live on the device (csc.py:143-346, 425-457)
    def __getitem__(self, key):
        """The eight slicing forms of the reference (csc.py:143-283; src/test/test2_slicing.py:6-79), same kernels, same
        results: A[a, :], A[:, b], A[:, :], A[a, list], A[list, b], A[:, list], A[list, :], A[list, list].  The row
        selections run cs3_csc_sub_matrix on the device and keep the reference's row numbering (a running match counter:
        csc_numba.py:464-502, 541-578 -- csc_sub_matrix_rows is csc_sub_matrix over all columns, loop for loop); a column
        selection copies whole columns with their original row indices (csc_sub_matrix_cols, csc_numba.py:505-538)."""
        from collections.abc import Iterable
        if not isinstance(key, tuple):
            raise Exception('The indices must be a tuple :/')
        a, b = key
        is_int = lambda v: isinstance(v, (int, np.integer))
        as_idx = lambda v: np.asarray([v] if is_int(v) else v, dtype=np.int32)
        if is_int(a) and is_int(b):
            raise NotImplementedError('Single value extraction not implemented')       # csc.py:150
        if isinstance(a, slice) and isinstance(b, slice):
            return self                                                                # csc.py:180-182
        if isinstance(a, slice):                         # (:, b) / (:, list_b): whole columns
            cols = as_idx(b)
            Bp, Bi, Bx = _sub_matrix_cols(self.indptr, self.indices, self.data, cols)
            return CscMat(self.m, len(cols), indptr=Bp, indices=Bi, data=Bx)
        if not (is_int(a) or isinstance(a, Iterable)) or not (is_int(b) or isinstance(b, slice) or isinstance(b, Iterable)):
            raise Exception('The indices must be a tuple :/')
        rows = as_idx(a)
        cols = np.arange(self.n, dtype=np.int32) if isinstance(b, slice) else as_idx(b)      # (a, :) / (list_a, :): every column
        _, Bp, Bi, Bx = _k.csc_sub_matrix(self.m, self.nzmax, self.indptr, self.indices, self.data, rows, cols)
        return CscMat(len(rows), len(cols), indptr=Bp, indices=Bi, data=Bx)

    def __setitem__(self, key, value):
        raise Exception('Setting values is not allowed in a CSC Matrix, use a Lil Matrix instead and convert it to CSC')

    def _add(self, other, beta):
        if isinstance(other, CscMat):
            assert other.m == self.m                     # csc.py:309-310
            assert other.n == self.n
            m, n, Cp, Ci, Cx = _k.csc_add_ff(self.m, self.n, self.indptr, self.indices, self.data,
                                             other.m, other.n, other.indptr, other.indices, other.data, 1.0, beta)
            return CscMat(m, n, indptr=Cp, indices=Ci, data=Cx)
        if isinstance(other, (float, int)):
            raise NotImplementedError('Adding a nonzero scalar to a sparse matrix would make it a dense matrix.')
        raise NotImplementedError('Type not supported')

    def __add__(self, other):
        """A + B on the device through the package's own kernel csc_add_ff (csc_numba.py:183-219: entries in its order,
        explicit zeros kept); the reference's operator goes through SciPy's csc_plus_csc (csc.py:301-322) -- the dense
        results are equal, which is what its test compares (src/test/test1_operations.py:12-72)."""
        return self._add(other, 1.0)

    def __sub__(self, other):
        return self._add(other, -1.0)

    def __neg__(self):
        return self.__mul__(-1.0)                        # csc.py:425-430

    def __eq__(self, other):
        """Same shape and the same three arrays, entry by entry (csc.py:432-457)."""
        if not isinstance(other, CscMat) or self.shape != other.shape:
            return False
        same = lambda x, y: all(p == q for p, q in zip(x, y))
        return bool(same(self.indices, other.indices) and same(self.indptr, other.indptr) and same(self.data, other.data))

    __hash__ = None

    def __mul__(self, other):
        """A * x for a vector or an [n, k] block, on the device (csc.py:372-415 semantics)."""
        if isinstance(other, np.ndarray):
            return _k.csc_mat_vec_ff(self.m, self.n, self.indptr, self.indices, self.data, other)
        if isinstance(other, (int, float)):
            C = self.copy()
            C.data *= other
            return C
        raise Exception("Type not supported")

    def to_csr(self):
        """CSR arrays (Bp, Bi, Bx) of this matrix, on the device (csc.py:466-478 -> csc_to_csr)."""
        nnz = int(self.indptr[self.n])
        Bp = np.zeros(self.m + 1, dtype=np.int32)
        Bi = np.empty(nnz, dtype=np.int32)
        Bx = np.empty(nnz, dtype=np.float64)
        _k.csc_to_csr(self.m, self.n, self.indptr, self.indices, self.data, Bp, Bi, Bx)
        return Bp, Bi, Bx

    def t(self):
        """Transpose, on the device (csc.py:502-513 -> csc_transpose)."""
        C = CscMat()
        C.m, C.n, C.indptr, C.indices, C.data = _k.csc_transpose(self.m, self.n, self.indptr, self.indices, self.data)
        C.nzmax = len(C.data)
        return C

    def islands(self):
        """Islands of the (structurally symmetric) pattern, each sorted (csc.py:515-521 -> find_islands)."""
        return _k.find_islands(self.n, self.indptr, self.indices)

    def norm(self):
        """1-norm (csc_norm, csc_numba.py:723-739)."""
        return _k.csc_norm(self.n, self.indptr, self.data)

    # ---- new: factor / solve
    def _pattern_fingerprint(self):
        """(m, n, nnz, digest of indptr / indices): the symbolic analysis is valid for exactly this pattern.
        indptr / indices may be changed in place or reassigned between two lu() calls; comparing the digest
        (a few ms at 500k entries) is what makes "mutate, then refactor" safe."""
        n = self.n
        indptr = np.ascontiguousarray(self.indptr[:n + 1], dtype=np.int32)
        nnz = int(indptr[n]) if n > 0 else 0
        h = hashlib.blake2b(digest_size=16)
        h.update(indptr.tobytes())
        h.update(np.ascontiguousarray(self.indices[:nnz], dtype=np.int32).tobytes())
        return self.m, n, nnz, h.digest()

    def _analysis(self, kind, order, q):
        key = (kind, order, None if q is None else bytes(np.asarray(q, dtype=np.int32)), self._pattern_fingerprint())
        f = self._factorization
        if f is None or f[0] != key:
            if f is not None:
                f[1].close()
            fac = _k.Factorization(self.m, self.n, self.indptr, self.indices, kind=kind, order=order, q=q)
            self._factorization = f = (key, fac)
        return f[1]

    def lu(self, tol=0.0, order=_k.ORDER_AMD, q=None):
        """Numeric LU on the device; the symbolic analysis is cached on the object, so calling
        lu() again after changing .data is a refactorisation with the pattern reused."""
        F = self._analysis(_k.CS3_LU, order, q)
        F.factor(self.data, tol)
        return F

    def chol(self, order=_k.ORDER_AMD, q=None):
        F = self._analysis(_k.CS3_CHOLESKY, order, q)
        F.factor(self.data)
        return F

    def solve(self, b, tol=0.0):
        """x = A \\ b by LU (factorises if needed)."""
        return self.lu(tol).solve(b)


def _sub_matrix_cols(Ap, Ai, Ax, cols):
    """Whole columns `cols` of a CSC matrix with their original row indices (csc_sub_matrix_cols, csc_numba.py:505-538):
    a copy of contiguous slices, done on the host where the arrays live."""
    counts = np.asarray([Ap[j + 1] - Ap[j] for j in cols], dtype=np.int64)
    Bp = np.zeros(len(cols) + 1, dtype=np.int32)
    Bp[1:] = np.cumsum(counts)
    if len(cols):
        take = np.concatenate([np.arange(Ap[j], Ap[j + 1]) for j in cols]) if counts.sum() else np.zeros(0, dtype=np.int64)
    else:
        take = np.zeros(0, dtype=np.int64)
    return Bp, np.asarray(Ai)[take].astype(np.int32), np.asarray(Ax)[take].astype(np.float64)


def scipy_to_mat(scipy_mat):
    """Alias SciPy's CSC arrays without copying (csc.py:541-553)."""
    m, n = scipy_mat.shape
    return CscMat(m, n, indptr=scipy_mat.indptr, indices=scipy_mat.indices, data=scipy_mat.data)


def lusol(A, b, order=1, tol=0.0):
    """x = A \\ b (cs_lusol)."""
    return _k.csc_lusol_f(order, A.m, A.n, A.indptr, A.indices, A.data, b, tol)


def cholsol(A, b, order=1):
    """x = A \\ b for symmetric positive definite A (cs_cholsol)."""
    return _k.csc_cholsol_f(order, A.m, A.n, A.indptr, A.indices, A.data, b)


def lsolve(L, x):
    """x = L \\ x in place, L a lower-triangular CscMat with the diagonal first in each column."""
    _k.csc_lsolve_f(L.n, L.indptr, L.indices, L.data, x)
    return x


def usolve(U, x):
    """x = U \\ x in place, U an upper-triangular CscMat with the diagonal last in each column."""
    _k.csc_usolve_f(U.n, U.indptr, U.indices, U.data, x)
    return x


def pack_4_by_4(A11, A12, A21, A22):
    """[[A11, A12], [A21, A22]] assembled on the device (csc.py:588-606: the power-flow Jacobian layout)."""
    m, n, Pi, Pp, Px = _k.csc_stack_4_by_4_ff(A11.m, A11.n, A11.indices, A11.indptr, A11.data,
                                              A12.m, A12.n, A12.indices, A12.indptr, A12.data,
                                              A21.m, A21.n, A21.indices, A21.indptr, A21.data,
                                              A22.m, A22.n, A22.indices, A22.indptr, A22.data)
    return CscMat(m, n, indptr=Pp, indices=Pi, data=Px)
