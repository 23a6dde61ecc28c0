/*
 * cs_oracle.c -- CPU oracle (see cs_oracle.h: TEST INFRASTRUCTURE ONLY,
 * "parity unpinned" for every function the reference lacks).
 *
 * Plain C99, single thread, no dependencies.  Build: make -C oracle
 */
#include "cs_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int32_t i32;
typedef int64_t i64;

#define FLIP(i)   (-(i) - 2)
#define MAX2(a,b) ((a) > (b) ? (a) : (b))
#define MIN2(a,b) ((a) < (b) ? (a) : (b))

static void *xmalloc(i64 count, size_t size)
{
    return malloc((size_t) MAX2(count, 1) * size);
}
static void *xcalloc(i64 count, size_t size)
{
    return calloc((size_t) MAX2(count, 1), size);
}

void orc_free(void *ptr) { free(ptr); }

void orc_csc_free(orc_csc *A)
{
    if (!A) return;
    free(A->p); free(A->i); free(A->x); free(A);
}

/* csc_spalloc_f, csc_numba.py:46-60: nzmax = max(nzmax, 1), zero-filled */
static orc_csc *csc_alloc(i64 m, i64 n, i64 nzmax, int values)
{
    orc_csc *A = (orc_csc *) xcalloc(1, sizeof(orc_csc));
    if (!A) return NULL;
    A->m = m; A->n = n; A->nzmax = MAX2(nzmax, 1);
    A->p = (i32 *) xcalloc(n + 1, sizeof(i32));
    A->i = (i32 *) xcalloc(A->nzmax, sizeof(i32));
    A->x = values ? (double *) xcalloc(A->nzmax, sizeof(double)) : NULL;
    if (!A->p || !A->i || (values && !A->x)) { orc_csc_free(A); return NULL; }
    return A;
}

/* csc_sprealloc_f, csc_numba.py:97-122: nzmax <= 0 trims to Ap[n] */
static int csc_resize(orc_csc *A, i64 nzmax)
{
    if (nzmax <= 0) nzmax = A->p[A->n];
    nzmax = MAX2(nzmax, 1);
    i32 *ni = (i32 *) realloc(A->i, (size_t) nzmax * sizeof(i32));
    if (!ni) return 0;
    A->i = ni;
    if (A->x) {
        double *nx = (double *) realloc(A->x, (size_t) nzmax * sizeof(double));
        if (!nx) return 0;
        A->x = nx;
    }
    A->nzmax = nzmax;
    return 1;
}

/* ======================================================================
 * substrate -- functions that exist in the reference
 * ====================================================================== */

/* csc_cumsum_i, csc_numba.py:75-94 */
i64 orc_cumsum(i32 *p, i32 *c, i64 n)
{
    i64 nz = 0;
    for (i64 i = 0; i < n; i++) {
        p[i] = (i32) nz;
        nz += c[i];
        c[i] = p[i];
    }
    p[n] = (i32) nz;
    return nz;
}

/* csc_scatter_f / csc_scatter_ff, csc_numba.py:125-151 / 154-180 */
i64 orc_scatter(const i32 *Ap, const i32 *Ai, const double *Ax, i64 j,
                double beta, i32 *w, double *x, i64 mark, i32 *Ci, i64 nz)
{
    for (i32 p = Ap[j]; p < Ap[j + 1]; p++) {
        i32 i = Ai[p];
        if (w[i] < mark) {
            w[i] = (i32) mark;
            Ci[nz++] = i;
            if (x) x[i] = beta * Ax[p];
        } else if (x) {
            x[i] += beta * Ax[p];
        }
    }
    return nz;
}

/* csc_add_ff, csc_numba.py:183-219.  Ax/Bx may be NULL: pattern only. */
orc_csc *orc_add(i64 Am, i64 An, const i32 *Ap, const i32 *Ai, const double *Ax,
                 i64 Bm, i64 Bn, const i32 *Bp, const i32 *Bi, const double *Bx,
                 double alpha, double beta)
{
    (void) Bm;
    int values = (Ax != NULL && Bx != NULL);
    i64 m = Am, n = Bn, nz = 0;
    i64 anz = Ap[An], bnz = Bp[n];
    i32 *w = (i32 *) xcalloc(m, sizeof(i32));
    double *x = values ? (double *) xcalloc(m, sizeof(double)) : NULL;
    orc_csc *C = csc_alloc(m, n, anz + bnz, values);
    if (!w || !C || (values && !x)) { free(w); free(x); orc_csc_free(C); return NULL; }
    for (i64 j = 0; j < n; j++) {
        C->p[j] = (i32) nz;
        nz = orc_scatter(Ap, Ai, Ax, j, alpha, w, x, j + 1, C->i, nz);
        nz = orc_scatter(Bp, Bi, Bx, j, beta, w, x, j + 1, C->i, nz);
        if (values)
            for (i64 p = C->p[j]; p < nz; p++) C->x[p] = x[C->i[p]];
    }
    C->p[n] = (i32) nz;
    free(w); free(x);
    return C;
}

/* csc_transpose, csc_numba.py:400-436.  Ax may be NULL: pattern only. */
orc_csc *orc_transpose(i64 m, i64 n, const i32 *Ap, const i32 *Ai, const double *Ax)
{
    orc_csc *C = csc_alloc(n, m, Ap[n], Ax != NULL);
    i32 *w = (i32 *) xcalloc(m, sizeof(i32));
    if (!C || !w) { orc_csc_free(C); free(w); return NULL; }
    for (i32 p = 0; p < Ap[n]; p++) w[Ai[p]]++;
    orc_cumsum(C->p, w, m);
    for (i64 j = 0; j < n; j++) {
        for (i32 p = Ap[j]; p < Ap[j + 1]; p++) {
            i32 q = w[Ai[p]]++;
            C->i[q] = (i32) j;
            if (Ax) C->x[q] = Ax[p];
        }
    }
    free(w);
    return C;
}

/* csc_to_csr, csc_numba.py:360-397 (Bp must arrive zeroed, as there) */
void orc_to_csr(i64 m, i64 n, const i32 *Ap, const i32 *Ai, const double *Ax,
                i32 *Bp, i32 *Bi, double *Bx)
{
    i32 nnz = Ap[n];
    for (i32 k = 0; k < nnz; k++) Bp[Ai[k]]++;
    i32 run = 0;
    for (i64 r = 0; r < m; r++) { i32 t = Bp[r]; Bp[r] = run; run += t; }
    Bp[m] = nnz;
    for (i64 j = 0; j < n; j++) {
        for (i32 p = Ap[j]; p < Ap[j + 1]; p++) {
            i32 r = Ai[p], dest = Bp[r];
            Bi[dest] = (i32) j;
            Bx[dest] = Ax[p];
            Bp[r]++;
        }
    }
    i32 last = 0;
    for (i64 r = 0; r < m; r++) { i32 t = Bp[r]; Bp[r] = last; last = t; }
}

/* csc_mat_vec_ff, csc_numba.py:309-328 */
void orc_mat_vec(i64 m, i64 n, const i32 *Ap, const i32 *Ai, const double *Ax,
                 const double *x, double *y)
{
    for (i64 i = 0; i < m; i++) y[i] = 0.0;
    for (i64 j = 0; j < n; j++)
        for (i32 p = Ap[j]; p < Ap[j + 1]; p++)
            y[Ai[p]] += Ax[p] * x[j];
}

/* multi-vector form used by CscMat.__mul__, csc.py:409-414; semantics of
 * csc_matvecs, /root/reference/src/sparsetools/csc.h:68-84: X is [n,k] and
 * Y is [m,k], both row-major; Y += A X with Y zeroed by the caller there,
 * zeroed here. */
void orc_mat_vecs(i64 m, i64 n, i64 k, const i32 *Ap, const i32 *Ai,
                  const double *Ax, const double *X, double *Y)
{
    for (i64 t = 0; t < m * k; t++) Y[t] = 0.0;
    for (i64 j = 0; j < n; j++) {
        for (i32 p = Ap[j]; p < Ap[j + 1]; p++) {
            double a = Ax[p];
            double *y = Y + (i64) Ai[p] * k;
            const double *x = X + j * k;
            for (i64 t = 0; t < k; t++) y[t] += a * x[t];
        }
    }
}

/* csc_norm, csc_numba.py:723-739 */
double orc_norm(i64 n, const i32 *Ap, const double *Ax)
{
    double norm = 0.0;
    for (i64 j = 0; j < n; j++) {
        double s = 0.0;
        for (i32 p = Ap[j]; p < Ap[j + 1]; p++) s += fabs(Ax[p]);
        norm = MAX2(norm, s);
    }
    return norm;
}

/* coo_to_csc, csc_numba.py:331-357 (unsorted, duplicates kept) */
orc_csc *orc_coo_to_csc(i64 m, i64 n, const i32 *Ti, const i32 *Tj,
                        const double *Tx, i64 nz)
{
    orc_csc *C = csc_alloc(m, n, nz, 1);
    i32 *w = (i32 *) xcalloc(n, sizeof(i32));
    if (!C || !w) { orc_csc_free(C); free(w); return NULL; }
    for (i64 k = 0; k < nz; k++) w[Tj[k]]++;
    orc_cumsum(C->p, w, n);
    for (i64 k = 0; k < nz; k++) {
        i32 p = w[Tj[k]]++;
        C->i[p] = Ti[k];
        C->x[p] = Tx[k];
    }
    free(w);
    return C;
}

/* csc_sub_matrix, csc_numba.py:464-502, statement by statement -- including its
 * row numbering: `i` is a running counter that advances on every match and is
 * bumped from 0 to 1 after a selected row without one; it is NOT the position of
 * the row in `rows`.  Returns a CSC with n = ncols columns (m is not defined by
 * the reference; the number of selected rows is recorded). */
orc_csc *orc_sub_matrix(i64 An, const i32 *Ap, const i32 *Ai, const double *Ax,
                        const i32 *rows, i64 nrows, const i32 *cols, i64 ncols)
{
    (void) An;
    i64 cap = 0;
    for (i64 c = 0; c < ncols; c++) cap += (i64) (Ap[cols[c] + 1] - Ap[cols[c]]) * (nrows > 0 ? 1 : 0);
    /* a row listed twice matches twice: size for the worst case */
    cap = cap * (nrows > 0 ? nrows : 1);
    if (cap > ((i64) 1 << 28)) cap = (i64) 1 << 28;
    orc_csc *B = csc_alloc(nrows, ncols, cap > 0 ? cap : 1, 1);
    if (!B) return NULL;
    i64 n = 0;
    B->p[0] = 0;
    for (i64 c = 0; c < ncols; c++) {
        i32 j = cols[c];
        i32 i = 0;
        for (i64 rr = 0; rr < nrows; rr++) {
            i32 r = rows[rr];
            for (i32 k = Ap[j]; k < Ap[j + 1]; k++) {
                if (Ai[k] == r) {
                    if (n >= cap) { orc_csc_free(B); return NULL; }
                    B->x[n] = Ax[k];
                    B->i[n] = i;
                    i++;
                    n++;
                }
            }
            if (i == 0) i++;
        }
        B->p[c + 1] = (i32) n;
    }
    return B;
}

/* find_islands, csc_numba.py:744-808: walk from every unvisited node in ascending
 * order, first-in first-out (the reference pops the front of its "stack"),
 * following the entries of column v to their row indices.  island_of[v] receives
 * the index of v's island; returns the number of islands.  The sort of every
 * island done by CscMat.islands (csc.py:515-521) is left to the caller. */
int64_t orc_find_islands(i64 n, const i32 *Ap, const i32 *Ai, i32 *island_of)
{
    unsigned char *visited = (unsigned char *) xcalloc(n > 0 ? n : 1, 1);
    i64 cap = (n > 0 ? n : 1) + (Ap ? (i64) Ap[n] : 0) + 1;
    i32 *queue = (i32 *) xcalloc(cap, sizeof(i32));
    if (!visited || !queue) { free(visited); free(queue); return -1; }
    i64 count = 0;
    for (i64 node = 0; node < n; node++) {
        if (visited[node]) continue;
        i64 head = 0, tail = 0;
        queue[tail++] = (i32) node;
        while (head < tail) {
            i32 v = queue[head++];
            if (visited[v]) continue;
            visited[v] = 1;
            island_of[v] = (i32) count;
            for (i32 p = Ap[v]; p < Ap[v + 1]; p++) {
                i32 k = Ai[p];
                if (!visited[k] && tail < cap) queue[tail++] = k;
            }
        }
        count++;
    }
    free(visited); free(queue);
    return count;
}

/* csc_stack_4_by_4_ff, csc_numba.py:640-720 -- argument order (m, n,
 * indices, indptr, data) as in the reference.  [[A, B], [C, D]]. */
orc_csc *orc_stack_4_by_4(i64 am, i64 an, const i32 *Ai, const i32 *Ap, const double *Ax,
                          i64 bm, i64 bn, const i32 *Bi, const i32 *Bp, const double *Bx,
                          i64 cm, i64 cn, const i32 *Ci, const i32 *Cp, const double *Cx,
                          i64 dm, i64 dn, const i32 *Di, const i32 *Dp, const double *Dx)
{
    if (am != bm || cm != dm || an != cn || bn != dn) return NULL;
    i64 nnz = (i64) Ap[an] + Bp[bn] + Cp[cn] + Dp[dn];
    orc_csc *P = csc_alloc(am + cm, an + bn, nnz, 1);
    if (!P) return NULL;
    i64 cnt = 0;
    for (i64 j = 0; j < an; j++) {
        for (i32 k = Ap[j]; k < Ap[j + 1]; k++) { P->i[cnt] = Ai[k]; P->x[cnt++] = Ax[k]; }
        for (i32 k = Cp[j]; k < Cp[j + 1]; k++) { P->i[cnt] = Ci[k] + (i32) am; P->x[cnt++] = Cx[k]; }
        P->p[j + 1] = (i32) cnt;
    }
    for (i64 j = 0; j < bn; j++) {
        for (i32 k = Bp[j]; k < Bp[j + 1]; k++) { P->i[cnt] = Bi[k]; P->x[cnt++] = Bx[k]; }
        for (i32 k = Dp[j]; k < Dp[j + 1]; k++) { P->i[cnt] = Di[k] + (i32) bm; P->x[cnt++] = Dx[k]; }
        P->p[an + j + 1] = (i32) cnt;
    }
    return P;
}

/* ======================================================================
 * permutations
 * ====================================================================== */

void orc_pinv(const i32 *p, i32 *pinv, i64 n)
{
    for (i64 k = 0; k < n; k++) pinv[p[k]] = (i32) k;
}

void orc_ipvec(const i32 *p, const double *b, double *x, i64 n)
{
    for (i64 k = 0; k < n; k++) x[p ? p[k] : k] = b[k];
}

void orc_pvec(const i32 *p, const double *b, double *x, i64 n)
{
    for (i64 k = 0; k < n; k++) x[k] = b[p ? p[k] : k];
}

/* Davis section 2.11: upper triangle of the symmetric permutation */
orc_csc *orc_symperm(i64 n, const i32 *Ap, const i32 *Ai, const double *Ax,
                     const i32 *pinv)
{
    orc_csc *C = csc_alloc(n, n, Ap[n], Ax != NULL);
    i32 *w = (i32 *) xcalloc(n, sizeof(i32));
    if (!C || !w) { orc_csc_free(C); free(w); return NULL; }
    for (i64 j = 0; j < n; j++) {
        i32 j2 = pinv ? pinv[j] : (i32) j;
        for (i32 p = Ap[j]; p < Ap[j + 1]; p++) {
            i32 i = Ai[p];
            if (i > j) continue;
            i32 i2 = pinv ? pinv[i] : i;
            w[MAX2(i2, j2)]++;
        }
    }
    orc_cumsum(C->p, w, n);
    for (i64 j = 0; j < n; j++) {
        i32 j2 = pinv ? pinv[j] : (i32) j;
        for (i32 p = Ap[j]; p < Ap[j + 1]; p++) {
            i32 i = Ai[p];
            if (i > j) continue;
            i32 i2 = pinv ? pinv[i] : i;
            i32 q = w[MAX2(i2, j2)]++;
            C->i[q] = MIN2(i2, j2);
            if (Ax) C->x[q] = Ax[p];
        }
    }
    free(w);
    return C;
}

/* Davis section 2.9 */
orc_csc *orc_permute(i64 m, i64 n, const i32 *Ap, const i32 *Ai, const double *Ax,
                     const i32 *pinv, const i32 *q)
{
    orc_csc *C = csc_alloc(m, n, Ap[n], Ax != NULL);
    if (!C) return NULL;
    i64 nz = 0;
    for (i64 k = 0; k < n; k++) {
        C->p[k] = (i32) nz;
        i32 j = q ? q[k] : (i32) k;
        for (i32 t = Ap[j]; t < Ap[j + 1]; t++) {
            if (Ax) C->x[nz] = Ax[t];
            C->i[nz++] = pinv ? pinv[Ai[t]] : Ai[t];
        }
    }
    C->p[n] = (i32) nz;
    return C;
}

/* ======================================================================
 * elimination tree, postorder, column counts  (Davis chapter 4)
 * ====================================================================== */

/* section 4.1: etree of a symmetric matrix whose upper triangle is stored */
void orc_etree(i64 n, const i32 *Ap, const i32 *Ai, i32 *parent)
{
    i32 *ancestor = (i32 *) xmalloc(n, sizeof(i32));
    for (i64 k = 0; k < n; k++) {
        parent[k] = -1;
        ancestor[k] = -1;
        for (i32 p = Ap[k]; p < Ap[k + 1]; p++) {
            i32 i = Ai[p];
            while (i != -1 && i < k) {
                i32 up = ancestor[i];
                ancestor[i] = (i32) k;        /* path compression */
                if (up == -1) parent[i] = (i32) k;
                i = up;
            }
        }
    }
    free(ancestor);
}

/* non-recursive DFS of one tree; children lists in head/next */
static i64 tree_dfs(i64 root, i64 k, i64 *head, const i64 *next, i32 *post, i64 *stack)
{
    i64 top = 0;
    stack[0] = root;
    while (top >= 0) {
        i64 p = stack[top];
        i64 child = head[p];
        if (child == -1) {
            top--;
            post[k++] = (i32) p;
        } else {
            head[p] = next[child];
            stack[++top] = child;
        }
    }
    return k;
}

/* section 4.3 */
void orc_post(i64 n, const i32 *parent, i32 *post)
{
    i64 *w = (i64 *) xmalloc(3 * n, sizeof(i64));
    i64 *head = w, *next = w + n, *stack = w + 2 * n;
    for (i64 j = 0; j < n; j++) head[j] = -1;
    for (i64 j = n - 1; j >= 0; j--) {       /* reverse, so lists are ascending */
        if (parent[j] == -1) continue;
        next[j] = head[parent[j]];
        head[parent[j]] = j;
    }
    i64 k = 0;
    for (i64 j = 0; j < n; j++)
        if (parent[j] == -1) k = tree_dfs(j, k, head, next, post, stack);
    free(w);
}

/* section 4.4: is j a leaf of the i-th row subtree; returns lca */
static i32 row_leaf(i32 i, i32 j, const i32 *first, i32 *maxfirst, i32 *prevleaf,
                    i32 *ancestor, int *jleaf)
{
    *jleaf = 0;
    if (i <= j || first[j] <= maxfirst[i]) return -1;
    maxfirst[i] = first[j];
    i32 jprev = prevleaf[i];
    prevleaf[i] = j;
    *jleaf = (jprev == -1) ? 1 : 2;
    if (*jleaf == 1) return i;
    i32 q = jprev;
    while (q != ancestor[q]) q = ancestor[q];
    for (i32 s = jprev; s != q; ) {
        i32 sp = ancestor[s];
        ancestor[s] = q;
        s = sp;
    }
    return q;
}

/* section 4.5: column counts of chol(C), C's upper triangle stored in A */
void orc_counts(i64 n, const i32 *Ap, const i32 *Ai, const i32 *parent,
                const i32 *post, i32 *colcount)
{
    orc_csc *AT = orc_transpose(n, n, Ap, Ai, NULL);
    i32 *w = (i32 *) xmalloc(4 * n, sizeof(i32));
    i32 *ancestor = w, *maxfirst = w + n, *prevleaf = w + 2 * n, *first = w + 3 * n;
    i32 *delta = colcount;
    for (i64 k = 0; k < 4 * n; k++) w[k] = -1;
    for (i64 k = 0; k < n; k++) {
        i32 j = post[k];
        delta[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = (i32) k;
    }
    for (i64 i = 0; i < n; i++) ancestor[i] = (i32) i;
    for (i64 k = 0; k < n; k++) {
        i32 j = post[k];
        if (parent[j] != -1) delta[parent[j]]--;
        for (i32 p = AT->p[j]; p < AT->p[j + 1]; p++) {
            i32 i = AT->i[p];
            int jleaf;
            i32 q = row_leaf(i, j, first, maxfirst, prevleaf, ancestor, &jleaf);
            if (jleaf >= 1) delta[j]++;
            if (jleaf == 2) delta[q]--;
        }
        if (parent[j] != -1) ancestor[j] = parent[j];
    }
    for (i64 j = 0; j < n; j++)
        if (parent[j] != -1) colcount[parent[j]] += colcount[j];
    orc_csc_free(AT);
    free(w);
}

/* ======================================================================
 * approximate minimum degree ordering  (Davis chapter 7; Amestoy, Davis,
 * Duff, SIAM J. Matrix Anal. Appl. 17(4), 1996)
 *
 * Quotient graph held in one index array G with per-object pointer ptr[],
 * length len[], element-list length elen[], supervariable size nv[].
 * Object states:  live variable  elen >= 0, nv > 0
 *                 element        elen == -2
 *                 dead variable  elen == -1, nv == 0
 * Tie-breaking: degree lists are LIFO; the pivot is the head of the lowest
 * non-empty list.  The result depends on it, and the HIP library's
 * ordering must make the same choices to be bit-exact with this one.
 * ====================================================================== */

typedef struct {
    i64 n, nzmax, cnz;
    i64 *G;                       /* quotient-graph index memory */
    i64 *ptr, *len, *elen, *nv, *degree, *w;
    i64 *head, *next, *last;      /* degree lists (doubly linked) */
    i64 *hhead;                   /* hash buckets */
} qgraph;

static i64 amd_wclear(i64 mark, i64 lemax, i64 *w, i64 n)
{
    if (mark < 2 || mark + lemax < 0) {
        for (i64 k = 0; k < n; k++) if (w[k] != 0) w[k] = 1;
        mark = 2;
    }
    return mark;
}

static void deg_insert(qgraph *g, i64 i, i64 d)
{
    if (g->head[d] != -1) g->last[g->head[d]] = i;
    g->next[i] = g->head[d];
    g->last[i] = -1;
    g->head[d] = i;
}

static void deg_remove(qgraph *g, i64 i)
{
    if (g->next[i] != -1) g->last[g->next[i]] = g->last[i];
    if (g->last[i] != -1) g->next[g->last[i]] = g->next[i];
    else g->head[g->degree[i]] = g->next[i];
}

/* compact G: every live object's list is moved to the front, order kept */
static void amd_garbage_collect(qgraph *g)
{
    i64 n = g->n, *G = g->G, *ptr = g->ptr;
    for (i64 j = 0; j < n; j++) {
        i64 p = ptr[j];
        if (p >= 0) { ptr[j] = G[p]; G[p] = FLIP(j); }
    }
    i64 q = 0;
    for (i64 p = 0; p < g->cnz; ) {
        i64 j = FLIP(G[p++]);
        if (j >= 0) {
            G[q] = ptr[j];
            ptr[j] = q++;
            for (i64 t = 0; t < g->len[j] - 1; t++) G[q++] = G[p++];
        }
    }
    g->cnz = q;
}

/* postorder of the assembly tree: same traversal as tree_dfs */
static i64 amd_tdfs(i64 root, i64 k, i64 *head, const i64 *next, i32 *post, i64 *stack)
{
    return tree_dfs(root, k, head, next, post, stack);
}

int orc_amd(i64 order, i64 m, i64 n, const i32 *Ap, const i32 *Ai, i32 *perm)
{
    if (order == 0) { for (i64 k = 0; k < n; k++) perm[k] = (i32) k; return 0; }
    if (order != 1 || m != n) return -1;
    if (n == 0) return 0;

    /* C = pattern(A + A') without the diagonal: column j holds A(:,j) in
     * A's order, then the entries of A'(:,j) not already present. */
    orc_csc *AT = orc_transpose(m, n, Ap, Ai, NULL);
    orc_csc *C = AT ? orc_add(m, n, Ap, Ai, NULL, n, m, AT->p, AT->i, NULL, 0, 0) : NULL;
    orc_csc_free(AT);
    if (!C) return -2;
    {
        i64 nz = 0;
        for (i64 j = 0; j < n; j++) {
            i64 p = C->p[j];
            C->p[j] = (i32) nz;
            for (; p < C->p[j + 1]; p++)
                if (C->i[p] != j) C->i[nz++] = C->i[p];
        }
        C->p[n] = (i32) nz;
    }

    qgraph g;
    g.n = n;
    g.cnz = C->p[n];
    g.nzmax = g.cnz + g.cnz / 5 + 2 * n;
    i64 *W = (i64 *) xmalloc(10 * (n + 1), sizeof(i64));
    g.G = (i64 *) xmalloc(g.nzmax, sizeof(i64));
    if (!W || !g.G) { free(W); free(g.G); orc_csc_free(C); return -2; }
    g.len = W;                 g.nv = W + (n + 1);       g.next = W + 2 * (n + 1);
    g.head = W + 3 * (n + 1);  g.elen = W + 4 * (n + 1); g.degree = W + 5 * (n + 1);
    g.w = W + 6 * (n + 1);     g.hhead = W + 7 * (n + 1);
    g.last = W + 8 * (n + 1);  g.ptr = W + 9 * (n + 1);
    for (i64 p = 0; p < g.cnz; p++) g.G[p] = C->i[p];
    for (i64 k = 0; k <= n; k++) g.ptr[k] = C->p[k];
    orc_csc_free(C);

    i64 *G = g.G, *ptr = g.ptr, *len = g.len, *elen = g.elen, *nv = g.nv;
    i64 *degree = g.degree, *w = g.w, *head = g.head, *next = g.next;
    i64 *last = g.last, *hhead = g.hhead;

    i64 dense = (i64) MAX2(16.0, 10.0 * sqrt((double) n));
    dense = MIN2(n - 2, dense);

    for (i64 k = 0; k < n; k++) len[k] = ptr[k + 1] - ptr[k];
    len[n] = 0;
    for (i64 i = 0; i <= n; i++) {
        head[i] = last[i] = next[i] = hhead[i] = -1;
        nv[i] = 1;
        w[i] = 1;
        elen[i] = 0;
        degree[i] = len[i];
    }
    i64 mark = amd_wclear(0, 0, w, n);
    i64 lemax = 0, mindeg = 0, nel = 0;
    elen[n] = -2;             /* object n: the element that absorbs dense rows */
    ptr[n] = -1;
    w[n] = 0;

    for (i64 i = 0; i < n; i++) {
        i64 d = degree[i];
        if (d == 0) {                       /* isolated node: eliminate now */
            elen[i] = -2;
            nel++;
            ptr[i] = -1;
            w[i] = 0;
        } else if (d > dense) {             /* dense node: order last */
            nv[i] = 0;
            elen[i] = -1;
            nel++;
            ptr[i] = FLIP(n);
            nv[n]++;
        } else {
            deg_insert(&g, i, d);
        }
    }

    while (nel < n) {
        /* ---- pivot: head of the lowest non-empty degree list ---- */
        i64 k = -1;
        for (; mindeg < n && (k = head[mindeg]) == -1; mindeg++) ;
        if (next[k] != -1) last[next[k]] = -1;
        head[mindeg] = next[k];
        i64 elenk = elen[k];
        i64 nvk = nv[k];
        nel += nvk;

        if (elenk > 0 && g.cnz + mindeg >= g.nzmax) amd_garbage_collect(&g);

        /* ---- build element k: Lk = (Ak  U  union of Le, e in Ek) \ k ---- */
        i64 dk = 0;
        nv[k] = -nvk;
        i64 p = ptr[k];
        i64 pk1 = (elenk == 0) ? p : g.cnz;      /* in place if Ek is empty */
        i64 pk2 = pk1;
        for (i64 k1 = 1; k1 <= elenk + 1; k1++) {
            i64 e, pj, ln;
            if (k1 > elenk) { e = k; pj = p; ln = len[k] - elenk; }
            else            { e = G[p++]; pj = ptr[e]; ln = len[e]; }
            for (i64 k2 = 1; k2 <= ln; k2++) {
                i64 i = G[pj++];
                i64 nvi = nv[i];
                if (nvi <= 0) continue;          /* dead, or already in Lk */
                dk += nvi;
                nv[i] = -nvi;
                G[pk2++] = i;
                deg_remove(&g, i);
            }
            if (e != k) { ptr[e] = FLIP(k); w[e] = 0; }   /* absorb e */
        }
        if (elenk != 0) g.cnz = pk2;
        degree[k] = dk;
        ptr[k] = pk1;
        len[k] = pk2 - pk1;
        elen[k] = -2;

        /* ---- scan 1: w[e] - mark = |Le \ Lk| for elements touching Lk ---- */
        mark = amd_wclear(mark, lemax, w, n);
        for (i64 pk = pk1; pk < pk2; pk++) {
            i64 i = G[pk];
            i64 eln = elen[i];
            if (eln <= 0) continue;
            i64 nvi = -nv[i];
            i64 wnvi = mark - nvi;
            for (i64 t = ptr[i]; t <= ptr[i] + eln - 1; t++) {
                i64 e = G[t];
                if (w[e] >= mark) w[e] -= nvi;
                else if (w[e] != 0) w[e] = degree[e] + wnvi;
            }
        }

        /* ---- scan 2: approximate degrees, prune, hash ---- */
        for (i64 pk = pk1; pk < pk2; pk++) {
            i64 i = G[pk];
            i64 p1 = ptr[i];
            i64 p2 = p1 + elen[i] - 1;
            i64 pn = p1;
            i64 h = 0, d = 0;
            for (i64 t = p1; t <= p2; t++) {
                i64 e = G[t];
                if (w[e] == 0) continue;
                i64 dext = w[e] - mark;
                if (dext > 0) { d += dext; G[pn++] = e; h += e; }
                else { ptr[e] = FLIP(k); w[e] = 0; }     /* aggressive absorption */
            }
            elen[i] = pn - p1 + 1;
            i64 p3 = pn;
            i64 p4 = p1 + len[i];
            for (i64 t = p2 + 1; t < p4; t++) {
                i64 j = G[t];
                i64 nvj = nv[j];
                if (nvj <= 0) continue;
                d += nvj;
                G[pn++] = j;
                h += j;
            }
            if (d == 0) {                        /* mass elimination */
                ptr[i] = FLIP(k);
                i64 nvi = -nv[i];
                dk -= nvi;
                nvk += nvi;
                nel += nvi;
                nv[i] = 0;
                elen[i] = -1;
            } else {
                degree[i] = MIN2(degree[i], d);
                G[pn] = G[p3];
                G[p3] = G[p1];
                G[p1] = k;
                len[i] = pn - p1 + 1;
                h = ((h < 0) ? -h : h) % n;
                next[i] = hhead[h];
                hhead[h] = i;
                last[i] = h;
            }
        }
        degree[k] = dk;
        lemax = MAX2(lemax, dk);
        mark = amd_wclear(mark + lemax, lemax, w, n);

        /* ---- supervariable detection inside each touched hash bucket ---- */
        for (i64 pk = pk1; pk < pk2; pk++) {
            i64 i = G[pk];
            if (nv[i] >= 0) continue;
            i64 h = last[i];
            i = hhead[h];
            hhead[h] = -1;
            for (; i != -1 && next[i] != -1; i = next[i], mark++) {
                i64 ln = len[i], eln = elen[i];
                for (i64 t = ptr[i] + 1; t <= ptr[i] + ln - 1; t++) w[G[t]] = mark;
                i64 jlast = i;
                for (i64 j = next[i]; j != -1; ) {
                    int same = (len[j] == ln) && (elen[j] == eln);
                    for (i64 t = ptr[j] + 1; same && t <= ptr[j] + ln - 1; t++)
                        if (w[G[t]] != mark) same = 0;
                    if (same) {
                        ptr[j] = FLIP(i);
                        nv[i] += nv[j];
                        nv[j] = 0;
                        elen[j] = -1;
                        j = next[j];
                        next[jlast] = j;
                    } else {
                        jlast = j;
                        j = next[j];
                    }
                }
            }
        }

        /* ---- finalise Lk and put survivors back into the degree lists ---- */
        i64 pf = pk1;
        for (i64 pk = pk1; pk < pk2; pk++) {
            i64 i = G[pk];
            i64 nvi = -nv[i];
            if (nvi <= 0) continue;
            nv[i] = nvi;
            i64 d = degree[i] + dk - nvi;
            d = MIN2(d, n - nel - nvi);
            deg_insert(&g, i, d);
            mindeg = MIN2(mindeg, d);
            degree[i] = d;
            G[pf++] = i;
        }
        nv[k] = nvk;
        len[k] = pf - pk1;
        if (len[k] == 0) { ptr[k] = -1; w[k] = 0; }
        if (elenk != 0) g.cnz = pf;
    }

    /* ---- postorder the assembly tree ---- */
    for (i64 i = 0; i < n; i++) ptr[i] = FLIP(ptr[i]);
    for (i64 j = 0; j <= n; j++) head[j] = -1;
    for (i64 j = n; j >= 0; j--) {               /* absorbed variables */
        if (nv[j] > 0) continue;
        next[j] = head[ptr[j]];
        head[ptr[j]] = j;
    }
    for (i64 e = n; e >= 0; e--) {               /* elements */
        if (nv[e] <= 0) continue;
        if (ptr[e] != -1) { next[e] = head[ptr[e]]; head[ptr[e]] = e; }
    }
    i32 *P = (i32 *) xmalloc(n + 1, sizeof(i32));
    i64 k = 0;
    for (i64 i = 0; i <= n; i++)
        if (ptr[i] == -1) k = amd_tdfs(i, k, head, next, P, w);
    for (i64 i = 0; i < n; i++) perm[i] = P[i];
    free(P); free(W); free(g.G);
    return 0;
}

/* ======================================================================
 * triangular solves  (Davis section 3.1)
 * L: diagonal first in each column.  U: diagonal last in each column.
 * ====================================================================== */

void orc_lsolve(i64 n, const i32 *Lp, const i32 *Li, const double *Lx, double *x)
{
    for (i64 j = 0; j < n; j++) {
        x[j] /= Lx[Lp[j]];
        for (i32 p = Lp[j] + 1; p < Lp[j + 1]; p++) x[Li[p]] -= Lx[p] * x[j];
    }
}

void orc_ltsolve(i64 n, const i32 *Lp, const i32 *Li, const double *Lx, double *x)
{
    for (i64 j = n - 1; j >= 0; j--) {
        for (i32 p = Lp[j] + 1; p < Lp[j + 1]; p++) x[j] -= Lx[p] * x[Li[p]];
        x[j] /= Lx[Lp[j]];
    }
}

void orc_usolve(i64 n, const i32 *Up, const i32 *Ui, const double *Ux, double *x)
{
    for (i64 j = n - 1; j >= 0; j--) {
        x[j] /= Ux[Up[j + 1] - 1];
        for (i32 p = Up[j]; p < Up[j + 1] - 1; p++) x[Ui[p]] -= Ux[p] * x[j];
    }
}

void orc_utsolve(i64 n, const i32 *Up, const i32 *Ui, const double *Ux, double *x)
{
    for (i64 j = 0; j < n; j++) {
        for (i32 p = Up[j]; p < Up[j + 1] - 1; p++) x[j] -= Ux[p] * x[Ui[p]];
        x[j] /= Ux[Up[j + 1] - 1];
    }
}

/* ======================================================================
 * sparse triangular solve with sparse right-hand side  (Davis section 3.2)
 * ====================================================================== */

/* DFS from row j in the graph of L (columns reached through pinv);
 * finished nodes are pushed to xi[--top].  marks[] replaces the pointer
 * flipping of the textbook version. */
static i64 reach_dfs(i64 j, const i32 *Gp, const i32 *Gi, i64 top, i32 *xi,
                     i32 *pstack, const i32 *pinv, char *marks)
{
    i64 head = 0;
    xi[0] = (i32) j;
    while (head >= 0) {
        j = xi[head];
        i32 jnew = pinv ? pinv[j] : (i32) j;
        if (!marks[j]) {
            marks[j] = 1;
            pstack[head] = (jnew < 0) ? 0 : Gp[jnew];
        }
        int done = 1;
        i32 pend = (jnew < 0) ? 0 : Gp[jnew + 1];
        for (i32 p = pstack[head]; p < pend; p++) {
            i32 i = Gi[p];
            if (marks[i]) continue;
            pstack[head] = p;
            xi[++head] = i;
            done = 0;
            break;
        }
        if (done) {
            head--;
            xi[--top] = (i32) j;
        }
    }
    return top;
}

/* x = L \ B(:,k) on the reach; returns top, pattern in xi[top..n-1] */
static i64 sp_lsolve(i64 n, const i32 *Lp, const i32 *Li, const double *Lx,
                     const i32 *Bp, const i32 *Bi, const double *Bx, i64 k,
                     i32 *xi, double *x, const i32 *pinv, char *marks)
{
    i64 top = n;
    for (i32 p = Bp[k]; p < Bp[k + 1]; p++)
        if (!marks[Bi[p]])
            top = reach_dfs(Bi[p], Lp, Li, top, xi, xi + n, pinv, marks);
    for (i64 p = top; p < n; p++) { marks[xi[p]] = 0; x[xi[p]] = 0.0; }
    for (i32 p = Bp[k]; p < Bp[k + 1]; p++) x[Bi[p]] = Bx[p];
    for (i64 px = top; px < n; px++) {
        i32 j = xi[px];
        i32 J = pinv ? pinv[j] : j;
        if (J < 0) continue;                     /* row j not yet pivotal */
        x[j] /= Lx[Lp[J]];
        for (i32 p = Lp[J] + 1; p < Lp[J + 1]; p++) x[Li[p]] -= Lx[p] * x[j];
    }
    return top;
}

/* ======================================================================
 * left-looking LU with threshold partial pivoting  (Davis section 6.2)
 * PAQ = LU, L unit lower (diagonal first), U upper (diagonal last).
 * tol = 1: partial pivoting; tol < 1 prefers the diagonal entry.
 * ====================================================================== */

int orc_lu(i64 n, const i32 *Ap, const i32 *Ai, const double *Ax, const i32 *q,
           double tol, orc_csc **Lout, orc_csc **Uout, i32 *pinv)
{
    i64 guess = 4 * (i64) Ap[n] + n;
    orc_csc *L = csc_alloc(n, n, guess, 1);
    orc_csc *U = csc_alloc(n, n, guess, 1);
    double *x = (double *) xcalloc(n, sizeof(double));
    i32 *xi = (i32 *) xmalloc(2 * n, sizeof(i32));
    char *marks = (char *) xcalloc(n, 1);
    int status = 0;
    *Lout = *Uout = NULL;
    if (!L || !U || !x || !xi || !marks) { status = -1000000000; goto done; }

    for (i64 i = 0; i < n; i++) pinv[i] = -1;
    for (i64 k = 0; k <= n; k++) L->p[k] = 0;
    i64 lnz = 0, unz = 0;
    for (i64 k = 0; k < n; k++) {
        L->p[k] = (i32) lnz;
        U->p[k] = (i32) unz;
        if ((lnz + n > L->nzmax && !csc_resize(L, 2 * L->nzmax + n)) ||
            (unz + n > U->nzmax && !csc_resize(U, 2 * U->nzmax + n))) {
            status = -1000000000; goto done;
        }
        i32 *Li = L->i, *Ui = U->i;
        double *Lx = L->x, *Ux = U->x;
        i32 col = q ? q[k] : (i32) k;
        i64 top = sp_lsolve(n, L->p, Li, Lx, Ap, Ai, Ax, col, xi, x, pinv, marks);

        i32 ipiv = -1;
        double a = -1.0;
        for (i64 p = top; p < n; p++) {
            i32 i = xi[p];
            if (pinv[i] < 0) {
                double t = fabs(x[i]);
                if (t > a) { a = t; ipiv = i; }
            } else {
                Ui[unz] = pinv[i];
                Ux[unz++] = x[i];
            }
        }
        if (ipiv == -1 || a <= 0.0) { status = -(int) (k + 1); goto done; }
        if (pinv[col] < 0 && fabs(x[col]) >= a * tol) ipiv = col;

        double pivot = x[ipiv];
        Ui[unz] = (i32) k;
        Ux[unz++] = pivot;
        pinv[ipiv] = (i32) k;
        Li[lnz] = ipiv;
        Lx[lnz++] = 1.0;
        for (i64 p = top; p < n; p++) {
            i32 i = xi[p];
            if (pinv[i] < 0) {
                Li[lnz] = i;
                Lx[lnz++] = x[i] / pivot;
            }
            x[i] = 0.0;
        }
    }
    L->p[n] = (i32) lnz;
    U->p[n] = (i32) unz;
    for (i64 p = 0; p < lnz; p++) L->i[p] = pinv[L->i[p]];
    csc_resize(L, 0);
    csc_resize(U, 0);
done:
    free(x); free(xi); free(marks);
    if (status != 0) { orc_csc_free(L); orc_csc_free(U); return status; }
    *Lout = L; *Uout = U;
    return 0;
}

/* ======================================================================
 * up-looking Cholesky  (Davis section 4.7)
 * ====================================================================== */

/* pattern of row k of L: nodes reached in the etree from the entries of
 * the upper triangle of column k; returned in s[top..n-1], topological */
static i64 etree_reach(i64 n, const i32 *Cp, const i32 *Ci, i64 k,
                       const i32 *parent, i32 *s, char *marks)
{
    i64 top = n;
    marks[k] = 1;
    for (i32 p = Cp[k]; p < Cp[k + 1]; p++) {
        i32 i = Ci[p];
        if (i > k) continue;
        i64 len = 0;
        for (; !marks[i]; i = parent[i]) { s[len++] = i; marks[i] = 1; }
        while (len > 0) s[--top] = s[--len];
    }
    for (i64 p = top; p < n; p++) marks[s[p]] = 0;
    marks[k] = 0;
    return top;
}

int orc_chol(i64 n, const i32 *Ap, const i32 *Ai, const double *Ax, const i32 *pinv,
             const i32 *parent, const i32 *cp, orc_csc **Lout)
{
    *Lout = NULL;
    orc_csc *C = orc_symperm(n, Ap, Ai, Ax, pinv);
    orc_csc *L = csc_alloc(n, n, cp[n], 1);
    i32 *c = (i32 *) xmalloc(2 * n, sizeof(i32));
    i32 *s = c + n;
    double *x = (double *) xcalloc(n, sizeof(double));
    char *marks = (char *) xcalloc(n, 1);
    int status = 0;
    if (!C || !L || !c || !x || !marks) { status = -1000000000; goto done; }
    for (i64 k = 0; k < n; k++) L->p[k] = c[k] = cp[k];
    for (i64 k = 0; k < n; k++) {
        i64 top = etree_reach(n, C->p, C->i, k, parent, s, marks);
        x[k] = 0.0;
        for (i32 p = C->p[k]; p < C->p[k + 1]; p++)
            if (C->i[p] <= k) x[C->i[p]] = C->x[p];
        double d = x[k];
        x[k] = 0.0;
        for (; top < n; top++) {
            i32 i = s[top];
            double lki = x[i] / L->x[L->p[i]];
            x[i] = 0.0;
            for (i32 p = L->p[i] + 1; p < c[i]; p++) x[L->i[p]] -= L->x[p] * lki;
            d -= lki * lki;
            i32 p = c[i]++;
            L->i[p] = (i32) k;
            L->x[p] = lki;
        }
        if (d <= 0.0) { status = -(int) (k + 1); goto done; }
        i32 p = c[k]++;
        L->i[p] = (i32) k;
        L->x[p] = sqrt(d);
    }
    L->p[n] = cp[n];
done:
    orc_csc_free(C); free(c); free(x); free(marks);
    if (status != 0) { orc_csc_free(L); return status; }
    *Lout = L;
    return 0;
}

/* ======================================================================
 * drivers  (Davis sections 6.2 / 4.8 / 8.*)
 * ====================================================================== */

int orc_lusol(i64 order, i64 n, const i32 *Ap, const i32 *Ai, const double *Ax,
              double *b, double tol)
{
    i32 *q = (i32 *) xmalloc(n, sizeof(i32));
    i32 *pinv = (i32 *) xmalloc(n, sizeof(i32));
    double *x = (double *) xmalloc(n, sizeof(double));
    orc_csc *L = NULL, *U = NULL;
    int status = orc_amd(order, n, n, Ap, Ai, q);
    if (status == 0) status = orc_lu(n, Ap, Ai, Ax, q, tol, &L, &U, pinv);
    if (status == 0) {
        orc_ipvec(pinv, b, x, n);                /* x = P b */
        orc_lsolve(n, L->p, L->i, L->x, x);
        orc_usolve(n, U->p, U->i, U->x, x);
        orc_ipvec(q, x, b, n);                   /* b(q) = x */
    }
    orc_csc_free(L); orc_csc_free(U);
    free(q); free(pinv); free(x);
    return status;
}

int orc_cholsol(i64 order, i64 n, const i32 *Ap, const i32 *Ai, const double *Ax,
                double *b)
{
    i32 *P = (i32 *) xmalloc(n, sizeof(i32));
    i32 *pinv = (i32 *) xmalloc(n, sizeof(i32));
    i32 *parent = (i32 *) xmalloc(n, sizeof(i32));
    i32 *post = (i32 *) xmalloc(n, sizeof(i32));
    i32 *cnt = (i32 *) xmalloc(n, sizeof(i32));
    i32 *cp = (i32 *) xmalloc(n + 1, sizeof(i32));
    double *x = (double *) xmalloc(n, sizeof(double));
    orc_csc *L = NULL;
    int status = orc_amd(order, n, n, Ap, Ai, P);
    if (status == 0) {
        orc_pinv(P, pinv, n);
        orc_csc *C = orc_symperm(n, Ap, Ai, NULL, pinv);
        orc_etree(n, C->p, C->i, parent);
        orc_post(n, parent, post);
        orc_counts(n, C->p, C->i, parent, post, cnt);
        orc_cumsum(cp, cnt, n);
        orc_csc_free(C);
        status = orc_chol(n, Ap, Ai, Ax, pinv, parent, cp, &L);
    }
    if (status == 0) {
        orc_ipvec(pinv, b, x, n);                /* x = P b */
        orc_lsolve(n, L->p, L->i, L->x, x);
        orc_ltsolve(n, L->p, L->i, L->x, x);
        orc_pvec(pinv, x, b, n);                 /* b = P' x */
    }
    orc_csc_free(L);
    free(P); free(pinv); free(parent); free(post); free(cnt); free(cp); free(x);
    return status;
}
