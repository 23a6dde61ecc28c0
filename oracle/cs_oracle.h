/*
 * cs_oracle.h -- CPU oracle for the sparse direct-solve hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under csparse3_amd/ may include, link,
 * import or execute this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * PARITY UNPINNED for ordering / etree / LU / Cholesky / lsolve / usolve:
 * the reference snapshot (SanPen/CSparse3 @ v1) contains none of these
 * functions, no test for them and no golden vector.
 * They are restated here from the published algorithms the reference credits
 * (T. A. Davis, "Direct Methods for Sparse Linear Systems", SIAM 2006 --
 * chapters cited per function; credited at
 * /root/reference/src/CSparse3/csc_numba.py:1-4,24-26), following the
 * reference's data conventions: int32 indptr/indices, float64 data, loose
 * (m, n, Ap, Ai, Ax) arguments, rows inside a column not assumed sorted,
 * nnz = Ap[n].
 *
 * PINNED (against the reference's own golden vector and against outputs of
 * the reference's Python run in the build container, tests/golden/): the
 * substrate functions that DO exist in the reference -- cumsum, scatter,
 * add, transpose, to_csr, mat_vec, norm, coo_to_csc, stack_4_by_4.  Each
 * cites the csc_numba.py lines it follows.
 */
#ifndef CS_ORACLE_H
#define CS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* heap-allocated CSC matrix returned by the factorisation routines */
typedef struct orc_csc {
    int64_t m, n, nzmax;
    int32_t *p;   /* [n+1] */
    int32_t *i;   /* [nzmax] */
    double  *x;   /* [nzmax] */
} orc_csc;

void orc_csc_free(orc_csc *A);
void orc_free(void *ptr);

/* ---- substrate (exists in the reference; pinned) ---------------------- */
int64_t orc_cumsum(int32_t *p, int32_t *c, int64_t n);
int64_t orc_scatter(const int32_t *Ap, const int32_t *Ai, const double *Ax,
                    int64_t j, double beta, int32_t *w, double *x,
                    int64_t mark, int32_t *Ci, int64_t nz);
orc_csc *orc_add(int64_t Am, int64_t An, const int32_t *Ap, const int32_t *Ai,
                 const double *Ax, int64_t Bm, int64_t Bn, const int32_t *Bp,
                 const int32_t *Bi, const double *Bx, double alpha, double beta);
orc_csc *orc_transpose(int64_t m, int64_t n, const int32_t *Ap,
                       const int32_t *Ai, const double *Ax);
void orc_to_csr(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai,
                const double *Ax, int32_t *Bp, int32_t *Bi, double *Bx);
void orc_mat_vec(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai,
                 const double *Ax, const double *x, double *y);
void orc_mat_vecs(int64_t m, int64_t n, int64_t k, const int32_t *Ap,
                  const int32_t *Ai, const double *Ax, const double *X, double *Y);
double orc_norm(int64_t n, const int32_t *Ap, const double *Ax);
orc_csc *orc_sub_matrix(int64_t An, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                        const int32_t *rows, int64_t nrows, const int32_t *cols, int64_t ncols);
int64_t orc_find_islands(int64_t n, const int32_t *Ap, const int32_t *Ai, int32_t *island_of);
orc_csc *orc_coo_to_csc(int64_t m, int64_t n, const int32_t *Ti,
                        const int32_t *Tj, const double *Tx, int64_t nz);
orc_csc *orc_stack_4_by_4(int64_t am, int64_t an, const int32_t *Ai, const int32_t *Ap, const double *Ax,
                          int64_t bm, int64_t bn, const int32_t *Bi, const int32_t *Bp, const double *Bx,
                          int64_t cm, int64_t cn, const int32_t *Ci, const int32_t *Cp, const double *Cx,
                          int64_t dm, int64_t dn, const int32_t *Di, const int32_t *Dp, const double *Dx);

/* ---- ordering / symbolic (absent from the reference; unpinned) -------- */
/* order: 0 natural, 1 amd(A+A').  q has n entries.  returns 0 on success. */
int orc_amd(int64_t order, int64_t m, int64_t n, const int32_t *Ap,
            const int32_t *Ai, int32_t *q);
/* pattern of triu(P A P') for pinv; values optional (Ax may be NULL) */
orc_csc *orc_symperm(int64_t n, const int32_t *Ap, const int32_t *Ai,
                     const double *Ax, const int32_t *pinv);
/* C = A(p,q): row i of A becomes row pinv[i], column k of C is column q[k] */
orc_csc *orc_permute(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai,
                     const double *Ax, const int32_t *pinv, const int32_t *q);
void orc_pinv(const int32_t *p, int32_t *pinv, int64_t n);
/* etree of a matrix whose upper triangle is given (ata = 0 only) */
void orc_etree(int64_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent);
void orc_post(int64_t n, const int32_t *parent, int32_t *post);
void orc_counts(int64_t n, const int32_t *Ap, const int32_t *Ai,
                const int32_t *parent, const int32_t *post, int32_t *colcount);

/* ---- numeric (absent from the reference; unpinned) -------------------- */
/* left-looking LU with threshold partial pivoting.  q may be NULL.
 * returns 0 ok, -(k+1) if no pivot found at step k. */
int orc_lu(int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
           const int32_t *q, double tol, orc_csc **L, orc_csc **U, int32_t *pinv);
/* up-looking Cholesky of triu(P A P'), given parent and column pointers cp.
 * returns 0 ok, -(k+1) if not positive definite at step k. */
int orc_chol(int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
             const int32_t *pinv, const int32_t *parent, const int32_t *cp,
             orc_csc **L);
void orc_lsolve(int64_t n, const int32_t *Lp, const int32_t *Li, const double *Lx, double *x);
void orc_usolve(int64_t n, const int32_t *Up, const int32_t *Ui, const double *Ux, double *x);
void orc_ltsolve(int64_t n, const int32_t *Lp, const int32_t *Li, const double *Lx, double *x);
void orc_utsolve(int64_t n, const int32_t *Up, const int32_t *Ui, const double *Ux, double *x);
void orc_ipvec(const int32_t *p, const double *b, double *x, int64_t n); /* x[p[k]] = b[k] */
void orc_pvec(const int32_t *p, const double *b, double *x, int64_t n);  /* x[k] = b[p[k]] */
/* b is overwritten with the solution.  order as in orc_amd. */
int orc_lusol(int64_t order, int64_t n, const int32_t *Ap, const int32_t *Ai,
              const double *Ax, double *b, double tol);
int orc_cholsol(int64_t order, int64_t n, const int32_t *Ap, const int32_t *Ai,
                const double *Ax, double *b);

#ifdef __cplusplus
}
#endif
#endif
