/*
 * csparse3_amd.h -- C ABI of the MI355X (gfx950) sparse direct-solve backend.
 *
 * This is the drop-in boundary for the factor/solve path of SanPen/CSparse3.
 * The reference selects its kernel module by name at import
 * (/root/reference/src/CSparse3/csc.py:29-41, __config__.NATIVE) and calls
 * kernels with loose flat arrays: scalars int64, index arrays int32, values
 * float64, matrices as (m, n, Ap, Ai, Ax)
 * (/root/reference/src/CSparse3/csc_numba.py:183,331,400 signature strings).
 * Every entry point below keeps that convention: plain pointers and sizes,
 * no torch / numpy types.  The reference has no factor/solve kernel at this
 * snapshot, so each function names the CSparse-lineage
 * kernel it stands for and the reference convention it follows, not a line it
 * replaces.  The ctypes binding a maintainer would add is in INTEGRATION.md.
 *
 * Return value: 0 on success, negative cs3_status otherwise;
 * cs3_last_error() gives the message for the calling thread.
 * Pointers named *_dev are device (HBM) addresses, everything else is host.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).
 * A handle may be used by one host thread at a time.
 */
#ifndef CSPARSE3_AMD_H
#define CSPARSE3_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs3_handle_s *cs3_handle;

enum cs3_status {
    CS3_OK = 0,
    CS3_ERR_ARG = -1,        /* bad argument / shape (the reference asserts, csc_numba.py:240,322) */
    CS3_ERR_ALLOC = -2,
    CS3_ERR_HIP = -3,        /* HIP runtime error, no device, launch failure */
    CS3_ERR_PIVOT = -4,      /* zero / rejected static pivot; column in cs3_info.fail_col */
    CS3_ERR_NOT_SPD = -5,    /* non-positive Cholesky pivot; column in cs3_info.fail_col */
    CS3_ERR_STATE = -6       /* call out of order (solve before factor ...) */
};

enum cs3_kind { CS3_LU = 0, CS3_CHOLESKY = 1 };
enum cs3_order { CS3_ORDER_NATURAL = 0, CS3_ORDER_AMD = 1, CS3_ORDER_GIVEN = 2 };

typedef struct cs3_info {
    int64_t n, nnz_a;
    int64_t nnz_l, nnz_u;          /* entries of L (incl. diagonal) and U (incl. diagonal) */
    int64_t nsuper, nlevels;       /* supernodes, levels of the supernodal tree */
    int64_t max_front, max_width;  /* largest front order r and supernode width w */
    int64_t factor_bytes;          /* dense panel storage in HBM, bytes per matrix */
    int64_t update_bytes;          /* contribution-block pool in HBM, bytes per matrix */
    int64_t batch;                 /* matrices factorised together (same pattern) */
    int64_t fail_col;              /* first failing pivot column (permuted index), -1 if none */
    double  flops_factor;          /* floating-point operations of the numeric phase */
    double  t_order_s, t_symbolic_s; /* host seconds spent in ordering / symbolic analysis */
} cs3_info;

const char *cs3_last_error(void);
int cs3_version(void);
int cs3_device_count(void);        /* 0 when no GPU is visible */

/* ---- ordering and symbolic kernels (host side of the library) ----------
 * Flat-array functions in the style of csc_numba.py; integer outputs are
 * bit-exact with the oracle.  They need no GPU. */

/* cs_amd lineage.  order 0: natural, 1: amd(A+A').  q[n] out. */
int cs3_amd(int64_t order, int64_t m, int64_t n, const int32_t *Ap,
            const int32_t *Ai, int32_t *q);
/* cs_etree lineage: (Ap, Ai) is the upper triangle of a symmetric pattern. */
int cs3_etree(int64_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent);
/* cs_post lineage. */
int cs3_post(int64_t n, const int32_t *parent, int32_t *post);
/* cs_counts lineage: column counts of the Cholesky factor of that pattern. */
int cs3_counts(int64_t n, const int32_t *Ap, const int32_t *Ai,
               const int32_t *parent, const int32_t *post, int32_t *colcount);

/* ---- analysis: ordering + symbolic + supernodes + schedule --------------
 * cs_sqr / cs_schol lineage.  Pattern only; values come with cs3_factor*.
 * kind: cs3_kind.  order: cs3_order; with CS3_ORDER_GIVEN q_given[n] is the
 * fill-reducing order to use.  For CS3_CHOLESKY (Ap, Ai) may hold the full
 * symmetric pattern or only one triangle.  Rows inside a column need not be
 * sorted (csc_numba.py:334-335); nnz is Ap[n], not the array length
 * (csc.py:138,480).  batch >= 1 matrices share this pattern. */
int cs3_analyze(int64_t kind, int64_t order, int64_t n, const int32_t *Ap,
                const int32_t *Ai, const int32_t *q_given, int64_t batch,
                cs3_handle *out);
int cs3_free(cs3_handle h);
int cs3_get_info(cs3_handle h, cs3_info *info);
/* Ordering results.  Any pointer may be NULL.
 * q_amd[n]:   the fill-reducing order before postordering (what cs_amd returns)
 * parent[n], post[n], colcount[n]: etree / postorder / column counts of
 *             A(q_amd,q_amd) + its transpose, as cs_schol computes them
 * q[n]:       the pivot order actually used = q_amd[post[.]]
 * pinv[n]:    row permutation of the factorisation, pinv[q[k]] = k */
int cs3_get_ordering(cs3_handle h, int32_t *q_amd, int32_t *parent, int32_t *post,
                     int32_t *colcount, int32_t *q, int32_t *pinv);
/* Supernode partition: sn_ptr[nsuper+1] first pivot column of each supernode,
 * sn_parent[nsuper], sn_level[nsuper]. */
int cs3_get_supernodes(cs3_handle h, int32_t *sn_ptr, int32_t *sn_parent, int32_t *sn_level);

/* ---- numeric factorisation (cs_lu / cs_chol lineage) --------------------
 * P A Q = L U with P = Q' (static diagonal pivots in AMD order), L unit lower,
 * U upper; or P A P' = L L'.  Ax[batch][nnz_a] in the entry order of the
 * analysed (Ap, Ai).  tol as in cs_lu: a diagonal pivot is accepted when
 * |pivot| >= tol * max|column below|; rejected pivots give CS3_ERR_PIVOT
 * (there is no CPU fallback).  tol <= 0 disables the test. */
int cs3_factor(cs3_handle h, const double *Ax, double tol);
int cs3_factor_dev(cs3_handle h, const double *Ax_dev, double tol, void *stream);
/* cs_lusol / cs_cholsol as one call on resident data: numeric (re)factorisation of Ax AND the full
 * solve of X [batch][n, k] in place.  Same results as cs3_factor_dev followed by cs3_solve_dev, bit
 * for bit; the forward sweep of a tree level runs beside the factorisation of the next level, so
 * the call is shorter than the two in sequence.  Status via cs3_factor_status. */
int cs3_factor_solve_dev(cs3_handle h, const double *Ax_dev, double tol, double *X_dev, int64_t k, void *stream);
/* The same, out of place: right-hand sides read from B_dev, solutions written to X_dev (x = A^-1 b without
 * the copy of b that the in-place form needs when b is kept). */
int cs3_factor_solve_bx_dev(cs3_handle h, const double *Ax_dev, double tol, const double *B_dev, double *X_dev, int64_t k, void *stream);
/* Deferred status of the last cs3_factor_dev / cs3_factor_solve_dev (synchronises the stream). */
int cs3_factor_status(cs3_handle h, void *stream);

/* ---- solves (cs_lsolve / cs_usolve / cs_ltsolve / cs_lusol / cs_cholsol) -
 * X is [n, k] row-major (the reference's multi-vector layout, csc.py:409-414),
 * overwritten in place.  With batch > 1, X is [batch, n, k].
 * cs3_solve:   full solve A x = b including both permutations
 * cs3_lsolve:  x = L \ x     in the permuted (pivot-order) space
 * cs3_usolve:  x = U \ x     (for Cholesky: x = L' \ x) */
int cs3_solve(cs3_handle h, double *X, int64_t k);
int cs3_lsolve(cs3_handle h, double *X, int64_t k);
int cs3_usolve(cs3_handle h, double *X, int64_t k);
int cs3_solve_dev(cs3_handle h, double *X_dev, int64_t k, void *stream);
int cs3_lsolve_dev(cs3_handle h, double *X_dev, int64_t k, void *stream);
int cs3_usolve_dev(cs3_handle h, double *X_dev, int64_t k, void *stream);

/* ---- residual and iterative refinement on resident data ----
 * R = B - A X with the handle's analysed pattern and the values Ax_dev [batch][nnz]; X, B, R [batch][n, k] row-major.
 * Every row of A X is summed as csc_mat_vec_ff sums it (csc_numba.py:309-328: ascending column, product rounded before
 * the add), so results are reproducible and A X equals the reference's matvec bit for bit. */
int cs3_residual_dev(cs3_handle h, const double *Ax_dev, const double *B_dev, const double *X_dev, double *R_dev,
                     int64_t k, void *stream);
/* Y = A X alone, same summation (the device-resident csc_mat_vec_ff; cs3_csc_matvec is the host-pointer form). */
int cs3_matvec_dev(cs3_handle h, const double *Ax_dev, const double *X_dev, double *Y_dev, int64_t k, void *stream);
/* `steps` rounds of  x += A \ (b - A x)  with the factors the handle holds (e.g. factors of an earlier Newton iterate
 * refining the solution for the current values Ax_dev).  last_correction (optional): max |dx| of the last round
 * (reading it synchronises the stream). */
int cs3_refine_dev(cs3_handle h, const double *Ax_dev, const double *B_dev, double *X_dev, int64_t k, int64_t steps,
                   double *last_correction, void *stream);

/* ---- factors back to the host in CSparse's CSC form ---------------------
 * L: diagonal FIRST in each column (unit for LU); U: diagonal LAST; row
 * indices sorted otherwise.  Sizes from cs3_info.nnz_l / nnz_u.  NumPy-style
 * ownership: the caller allocates, the library fills.  b = matrix index in
 * the batch.  For Cholesky Up/Ui/Ux must be NULL. */
int cs3_get_factors(cs3_handle h, int64_t b, int32_t *Lp, int32_t *Li, double *Lx,
                    int32_t *Up, int32_t *Ui, double *Ux);

/* ---- moving a factorisation between GPUs (BASELINE config 4) -----------
 * The numeric state of a handle is its factor panels: cs3_info.factor_bytes
 * per matrix, batch matrices back to back.  Export copies them into a caller
 * buffer in HBM (which RCCL then broadcasts); import installs such a buffer in
 * a handle that analysed the SAME pattern with the SAME ordering, after which
 * it solves as if it had factorised itself. */
int cs3_export_factor_dev(cs3_handle h, double *dst_dev, void *stream);
int cs3_import_factor_dev(cs3_handle h, const double *src_dev, void *stream);

/* ---- diagnostics --------------------------------------------------------
 * With CS3_PROFILE=1 in the environment every LDS-resident front records six
 * shader-clock stamps (descriptor read, zeroed, assembled, eliminated, staged,
 * stored) relative to its start; out[nsuper][8] in schedule order.  Not part
 * of the reference-facing surface. */
int cs3_debug_front_stamps(cs3_handle h, int64_t *out);
/* Fills the LDS of every CU with NaN bit patterns (the LDS keeps its contents between kernels): the parity tests call it
 * before the numeric entry points so that a product of a masked zero and an unwritten LDS word cannot hide. */
int cs3_debug_poison_lds(void *stream);
/* on != 0: the producers of the LDS hand-overs between waves (shared eliminations) keep their counters back, so every
 * consumer runs into its bounded wait and gives up: the step must then be reported as failed by cs3_factor_status
 * (CS3_ERR_STATE), never pass silently.  Process-wide; on = 0 restores the normal path.  For the test of that path. */
int cs3_debug_withhold_handover(int on);
/* Factorisation schedule: supernode id, front order r and width w per schedule slot. */
int cs3_debug_schedule(cs3_handle h, int32_t *sched, int32_t *front_r, int32_t *front_w);
/* The bottom forest (subtrees of small fronts walked by one workgroup each): returns the number of forest
 * fronts (0: no forest) and, for arrays that are not null, per forest front in task order its supernode, its task, its local
 * level inside the task and its tier (= launch).  Diagnostics and tests. */
int64_t cs3_debug_forest(cs3_handle h, int32_t *supernode, int32_t *task, int32_t *level, int32_t *tier);

/* ---- general triangular solves on caller-supplied CSC factors -----------
 * cs_lsolve / cs_usolve lineage, the csc_lsolve_f(n, Lp, Li, Lx, x) shape of
 * x[n, k] row-major, in place; diagonal first (L) /
 * last (U) in each column.  Level-scheduled on the device. */
int cs3_csc_lsolve(int64_t n, const int32_t *Lp, const int32_t *Li, const double *Lx,
                   double *x, int64_t k);
int cs3_csc_usolve(int64_t n, const int32_t *Up, const int32_t *Ui, const double *Ux,
                   double *x, int64_t k);

/* ---- neighbours of the path -----------------------
 * y = A x on the device, csc_mat_vec_ff (csc_numba.py:309-328) semantics;
 * X [n, k] and Y [m, k] row-major as csc_matvecs (sparsetools/csc.h:68-84). */
int cs3_csc_matvec(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai,
                   const double *Ax, const double *X, double *Y, int64_t k);

/* [[A, B], [C, D]] in CSC on the device: csc_stack_4_by_4_ff / pack_4_by_4 (csc_numba.py:640-720,
 * csc.py:588-606), the power-flow Jacobian assembly.  Argument order (m, n, indices, indptr, data)
 * as in the reference; outputs Pi[nnz], Pp[an+bn+1], Px[nnz] with nnz = the four blocks' nnz,
 * caller-allocated.  Incompatible block shapes (the reference asserts) give CS3_ERR_ARG. */
int cs3_csc_stack_4_by_4(int64_t am, int64_t an, const int32_t *Ai, const int32_t *Ap, const double *Ax,
                         int64_t bm, int64_t bn, const int32_t *Bi, const int32_t *Bp, const double *Bx,
                         int64_t cm, int64_t cn, const int32_t *Ci, const int32_t *Cp, const double *Cx,
                         int64_t dm, int64_t dn, const int32_t *Di, const int32_t *Dp, const double *Dx,
                         int32_t *Pi, int32_t *Pp, double *Px);

/* The same with every array already in HBM (device pointers in and out, asynchronous on `stream`): the Jacobian is
 * assembled where the factorisation reads it, so  stack -> cs3_factor_solve_dev  runs without a host copy.  nnz_* are
 * the blocks' entry counts (Ap[an] ...), passed by the caller so that nothing has to come back to the host.
 * map (optional, [nnz]): position of every output entry in the concatenation A | B | C | D of the value arrays. */
int cs3_csc_stack_4_by_4_dev(int64_t am, int64_t an, int64_t nnz_a, const int32_t *Ai_dev, const int32_t *Ap_dev, const double *Ax_dev,
                             int64_t bm, int64_t bn, int64_t nnz_b, const int32_t *Bi_dev, const int32_t *Bp_dev, const double *Bx_dev,
                             int64_t cm, int64_t cn, int64_t nnz_c, const int32_t *Ci_dev, const int32_t *Cp_dev, const double *Cx_dev,
                             int64_t dm, int64_t dn, int64_t nnz_d, const int32_t *Di_dev, const int32_t *Dp_dev, const double *Dx_dev,
                             int32_t *Pi_dev, int32_t *Pp_dev, double *Px_dev, int32_t *map_dev, void *stream);
/* Newton-loop restack: the blocks' patterns have not changed, only their values -- Px[p] = (A | B | C | D)[map[p]] with the
 * map of the first stacking.  One gather kernel; its output is what cs3_factor_solve_dev takes as Ax_dev. */
int cs3_restack_values_dev(int64_t nnz, const int32_t *map_dev, int64_t nnz_a, int64_t nnz_b, int64_t nnz_c,
                           const double *Ax_dev, const double *Bx_dev, const double *Cx_dev, const double *Dx_dev,
                           double *Px_dev, void *stream);

/* ---- format conversions and utilities on the device ----
 * Same outputs as the reference's Python kernels, bit for bit (tests/golden/): output ORDER included.
 * Host pointers in and out; results are caller-allocated. */
/* C = A' (csc_transpose, csc_numba.py:400-436); the same three arrays are A in CSR form
 * (csc_to_csr, :360-397).  Cp[m + 1], Ci / Cx[nnz]. */
int cs3_csc_transpose(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                      int32_t *Cp, int32_t *Ci, double *Cx);
/* Triplets to CSC, duplicates kept, triplet order inside a column (coo_to_csc, csc_numba.py:331-357). */
int cs3_coo_to_csc(int64_t m, int64_t n, int64_t nz, const int32_t *Ti, const int32_t *Tj, const double *Tx,
                   int32_t *Cp, int32_t *Ci, double *Cx);
/* 1-norm: max column sum of |x| (csc_norm, csc_numba.py:723-739). */
int cs3_csc_norm(int64_t n, const int32_t *Ap, const double *Ax, double *norm);
/* C = alpha A + beta B (csc_add_ff, csc_numba.py:183-219).  Ci / Cx: room for nnz(A) + nnz(B); used: Cp[n]. */
int cs3_csc_add(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                const int32_t *Bp, const int32_t *Bi, const double *Bx, double alpha, double beta,
                int32_t *Cp, int32_t *Ci, double *Cx);
/* B = A[rows, cols] exactly as csc_sub_matrix computes it (csc_numba.py:464-502), its running row counter
 * included.  Bp[ncols + 1]; Bi / Bx: room for b_cap entries (the reference allocates nnz(A)); used: Bp[ncols].
 * Repeated rows / columns can need more than nnz(A): then Bp is filled, nothing else is written and the call
 * returns CS3_ERR_ARG (the reference runs off its arrays there). */
int cs3_csc_sub_matrix(int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                       const int32_t *rows, int64_t nrows, const int32_t *cols, int64_t ncols,
                       int32_t *Bp, int32_t *Bi, double *Bx, int64_t b_cap);
/* label[i] = the node at which find_islands (csc_numba.py:744-808) opens the island that receives node i = the smallest
 * node that reaches i along column -> row edges.  On a structurally symmetric pattern that is the smallest node of i's
 * connected component; on an unsymmetric one it follows the reference's directed search exactly.  find_islands lists
 * islands by ascending start node and CscMat.islands (csc.py:515-521) sorts each -- both follow from the labels. */
int cs3_find_islands(int64_t n, const int32_t *Ap, const int32_t *Ai, int32_t *label);

#ifdef __cplusplus
}
#endif
#endif
