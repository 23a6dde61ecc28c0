// HIP kernels for gfx950 (MI355X): multifrontal numeric LU / Cholesky on
// supernodal fronts and the supernodal triangular sweeps.
//
// Data layout in HBM (per matrix of a batch):
//   vals   dense panels, supernode after supernode:
//            L panel  r x w column-major (ld = r); rows 0..w-1 are the pivot
//                     block: strictly lower part = L11 (unit diagonal implied
//                     for LU), upper part incl. diagonal = U11 (LU) / diagonal
//                     = L11's diagonal (Cholesky)
//            U panel  (LU only) (r-w) x w column-major: U12 transposed, so
//                     column k holds pivot row k of U
//   cb     contribution blocks (Schur complements) (r-w) x (r-w) column-major
//   cv     contribution vectors of the forward solve, (r-w) x nrhs row-major
// A front = [F11 F12; F21 F22], order r = |row structure|, w pivots.
// One workgroup owns one front; fronts of one tree level are independent.
// No float atomics anywhere: children are added in a fixed order, so results
// are bitwise reproducible run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "cs3_device.hpp"

namespace cs3 {

// ----------------------------------------------------- assembly by gather --
// A front's assembly list holds (target, source) pairs sorted by target; runs
// of equal targets never cross a 64-entry boundary, so one wave owns whole
// runs.  Lane l of a wave takes entry l of a 64-entry chunk, loads its source
// (pool offset, or ~index into Ax) and the wave sums each run with a fixed
// shuffle tree: every front entry is written exactly once, by one lane, and
// the summation order never changes from run to run.
constexpr int ASM_DUMMY_T = 0x3fffffff;
constexpr int ASM_LONG_T = 0x40000000;
constexpr int GATHER_UNROLL = 4;

__device__ __forceinline__ double gather_value(int t, int s, int s_next, const double *__restrict__ pool,
                                               const double *__restrict__ ax, const int *__restrict__ long_src)
{
    if (t == ASM_DUMMY_T) return 0.0;
    if (t & ASM_LONG_T) {                       // rare: more than 64 sources for one entry
        double acc = 0.0;
        for (int k = 0; k < s_next; ++k) {
            const int q = long_src[s + k];
            acc += (q >= 0) ? pool[q] : ax[~q];
        }
        return acc;
    }
    return (s >= 0) ? pool[s] : ax[~s];
}

// On return v holds the run total in the last lane of each run.
__device__ __forceinline__ bool run_totals(int t, double &v)
{
    const int lane = threadIdx.x & 63;
    const int t_next = __shfl_down(t, 1);
    const int t_prev = __shfl_up(t, 1);
    const bool is_last = (lane == 63) || (t_next != t);
    if (__any((lane > 0) && (t_prev == t))) {
        for (int d = 1; d < 64; d <<= 1) {
            const double o = __shfl_up(v, d);
            const int to = __shfl_up(t, d);
            const bool take = (lane >= d) && (to == t);
            if (take) v += o;
            if (!__any(take)) break;
        }
    }
    return is_last;
}

// Store(t, v) is called once per assembled entry.
template <int THREADS, class Store>
__device__ __forceinline__ void gather_front(long long asm_begin, int nchunks, int chunk_first, int chunk_stride,
                                             const int *__restrict__ asm_src, const int *__restrict__ asm_tgt,
                                             const int *__restrict__ long_src, const double *__restrict__ pool,
                                             const double *__restrict__ ax, Store store)
{
    const int lane = threadIdx.x & 63;
    for (int c0 = chunk_first; c0 < nchunks; c0 += chunk_stride) {
        int t[GATHER_UNROLL], s[GATHER_UNROLL];
        double v[GATHER_UNROLL];
#pragma unroll
        for (int u = 0; u < GATHER_UNROLL; ++u) {
            const bool valid = c0 + u < nchunks;
            const long long idx = asm_begin + (long long) (c0 + u) * 64 + lane;
            t[u] = valid ? asm_tgt[idx] : ASM_DUMMY_T;
            s[u] = valid ? asm_src[idx] : 0;
        }
#pragma unroll
        for (int u = 0; u < GATHER_UNROLL; ++u) {
            const int s_next = __shfl_down(s[u], 1);
            v[u] = gather_value(t[u], s[u], s_next, pool, ax, long_src);
        }
#pragma unroll
        for (int u = 0; u < GATHER_UNROLL; ++u) {
            const int tt = t[u] & ~ASM_LONG_T;
            const bool is_last = run_totals(tt, v[u]);
            if (is_last && tt != ASM_DUMMY_T) store(tt, v[u]);
        }
    }
}

__device__ __forceinline__ void flag_column(int *status, int col)
{
    atomicMin(status, col);
}

// ------------------------------------------------- front resident in LDS --
// THREADS = TX * TY; tx runs down a column (consecutive LDS addresses, no
// bank conflict), ty across columns.  ld is odd so that row reads (stride ld
// doubles) spread over all banks.
template <int KIND, int THREADS, int TX>
__global__ void __launch_bounds__(THREADS)
k_front_lds(const FrontDesc *__restrict__ fdesc, int first,
            const int *__restrict__ asm_src, const int *__restrict__ asm_tgt, const int *__restrict__ long_src,
            const double *__restrict__ ax_all, double *__restrict__ pool_all,
            long long nnz_a, long long pool_stride, double inv_tol, int *status)
{
    extern __shared__ __attribute__((aligned(16))) double F[];
    constexpr int TY = THREADS / TX;
    const FrontDesc d = fdesc[first + blockIdx.x];
    const double *ax = ax_all + (long long) blockIdx.y * nnz_a;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    const int r = d.r, w = d.w, nb = r - w;
    const int ld = r | 1;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;

    // ---- assemble: F = sum of sources (A entries, children's contribution blocks)
    for (int i = tid; i < r * ld; i += THREADS) F[i] = 0.0;
    __syncthreads();
    gather_front<THREADS>(d.asm_begin, d.asm_count >> 6, (tid >> 6) * GATHER_UNROLL, (THREADS / 64) * GATHER_UNROLL,
                          asm_src, asm_tgt, long_src, pool, ax,
                          [&](int t, double v) { F[t] = v; });
    __syncthreads();

    // ---- eliminate the w pivots (right-looking inside the front)
    for (int k = 0; k < w; ++k) {
        const double piv = F[k + k * ld];
        if (KIND == CS3_LU) {
            for (int i = k + 1 + tid; i < r; i += THREADS) F[i + k * ld] /= piv;
            __syncthreads();
            for (int j = k + 1 + ty; j < r; j += TY) {
                const double u = F[k + j * ld];
                for (int i = k + 1 + tx; i < r; i += TX) F[i + j * ld] -= F[i + k * ld] * u;
            }
        } else {
            const double dg = sqrt(piv);
            for (int i = k + 1 + tid; i < r; i += THREADS) F[i + k * ld] /= dg;
            __syncthreads();
            if (tid == 0) F[k + k * ld] = (piv > 0.0) ? dg : -1.0;   // -1 marks "not SPD"
            for (int j = k + 1 + ty; j < r; j += TY) {
                const double u = F[j + k * ld];
                for (int i = j + tx; i < r; i += TX) F[i + j * ld] -= F[i + k * ld] * u;
            }
        }
        __syncthreads();
    }

    // ---- write back: factors to the panels, Schur block to the pool
    double *L = pool + d.lpan;
    for (int j = ty; j < w; j += TY) {
        for (int i = tx; i < r; i += TX) {
            const double v = F[i + j * ld];
            if (KIND == CS3_LU) {
                if (i > j) { if (!(fabs(v) <= inv_tol)) flag_column(status, d.c0 + j); }
                else if (i == j) { if (!(fabs(v) > 0.0) || !(fabs(v) < 1.0e300)) flag_column(status, d.c0 + j); }
            } else if (i == j) {
                if (!(v > 0.0)) flag_column(status, d.c0 + j);
            }
            if (KIND == CS3_LU || i >= j) L[i + (long long) j * r] = v;
        }
    }
    if (KIND == CS3_LU) {
        double *U = pool + d.upan;
        for (int k = ty; k < w; k += TY)
            for (int i = tx; i < nb; i += TX) U[i + (long long) k * nb] = F[k + (w + i) * ld];
    }
    if (d.parent >= 0) {
        double *cb = pool + d.cb;
        for (int j = ty; j < nb; j += TY)
            for (int i = tx; i < nb; i += TX)
                if (KIND == CS3_LU || i >= j) cb[i + (long long) j * nb] = F[(w + i) + (w + j) * ld];
    }
}

// ------------------------------------------- front too large for the LDS --
// The front is one dense r x r column-major buffer in the pool (zeroed at the
// start of the factorisation).  Per front: one gather launch, then per block
// of BIG_NB pivots a panel launch and a trailing-update launch -- a blocked
// right-looking LU / Cholesky without pivoting, many workgroups per launch.
constexpr int BIG_NB = 32;

__global__ void __launch_bounds__(256)
k_big_gather(const FrontDesc *__restrict__ fdesc, int first,
             const int *__restrict__ asm_src, const int *__restrict__ asm_tgt, const int *__restrict__ long_src,
             const double *__restrict__ ax_all, double *__restrict__ pool_all,
             long long nnz_a, long long pool_stride)
{
    const FrontDesc d = fdesc[first + blockIdx.z];
    const double *ax = ax_all + (long long) blockIdx.y * nnz_a;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    gather_front<256>(d.asm_begin, d.asm_count >> 6, wave * GATHER_UNROLL, gridDim.x * 4 * GATHER_UNROLL,
                      asm_src, asm_tgt, long_src, pool, ax,
                      [&](int t, double v) { pool[t] = v; });
}

// Panel step kb: every workgroup factors the diagonal block D = F[kb:ke, kb:ke]
// on its own in LDS (cheap, and it avoids a grid-wide hand-off), then each
// thread solves one row of L21 (x U_D = f) or one column of U12 (L_D u = f).
template <int KIND>
__global__ void __launch_bounds__(64)
k_big_panel(const FrontDesc *__restrict__ fdesc, int first, int kb, double *__restrict__ pool_all,
            long long pool_stride, double inv_tol, int *status)
{
    __shared__ double D[BIG_NB][BIG_NB + 1];
    const FrontDesc d = fdesc[first + blockIdx.z];
    const int r = d.r, w = d.w;
    if (kb >= w) return;
    const int ke = min(kb + BIG_NB, w), bw = ke - kb;
    double *F = pool_all + (long long) blockIdx.y * pool_stride + d.lpan;
    const long long ld = r;
    const int tid = threadIdx.x;

    for (int e = tid; e < bw * bw; e += 64) {
        const int i = e % bw, j = e / bw;
        D[i][j] = F[(kb + i) + (kb + j) * ld];
    }
    __syncthreads();
    for (int k = 0; k < bw; ++k) {
        const double piv = D[k][k];
        if (KIND == CS3_LU) {
            if (tid > k && tid < bw) D[tid][k] /= piv;
            __syncthreads();
            for (int e = tid; e < (bw - k - 1) * (bw - k - 1); e += 64) {
                const int i = k + 1 + e % (bw - k - 1), j = k + 1 + e / (bw - k - 1);
                D[i][j] -= D[i][k] * D[k][j];
            }
        } else {
            const double dg = sqrt(piv);
            if (tid > k && tid < bw) D[tid][k] /= dg;
            __syncthreads();
            if (tid == 0) D[k][k] = (piv > 0.0) ? dg : -1.0;
            for (int e = tid; e < (bw - k - 1) * (bw - k - 1); e += 64) {
                const int i = k + 1 + e % (bw - k - 1), j = k + 1 + e / (bw - k - 1);
                if (i >= j) D[i][j] -= D[i][k] * D[j][k];
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int e = tid; e < bw * bw; e += 64) {
            const int i = e % bw, j = e / bw;
            const double v = D[i][j];
            if (KIND == CS3_LU) {
                if (i > j) { if (!(fabs(v) <= inv_tol)) flag_column(status, d.c0 + kb + j); }
                else if (i == j) { if (!(fabs(v) > 0.0) || !(fabs(v) < 1.0e300)) flag_column(status, d.c0 + kb + j); }
            } else if (i == j) {
                if (!(v > 0.0)) flag_column(status, d.c0 + kb + j);
            }
            if (KIND == CS3_LU || i >= j) F[(kb + i) + (kb + j) * ld] = v;
        }
    }
    const int below = r - ke;
    const int g = blockIdx.x * 64 + tid;
    if (g < below) {                              // one row of L21:  x U_D = f  (Cholesky: x L_D' = f)
        const int i = ke + g;
        double x[BIG_NB];
#pragma unroll
        for (int c = 0; c < BIG_NB; ++c) x[c] = (c < bw) ? F[i + (kb + c) * ld] : 0.0;
#pragma unroll
        for (int c = 0; c < BIG_NB; ++c) {
            if (c < bw) {
                double acc = x[c];
#pragma unroll
                for (int k = 0; k < BIG_NB; ++k)
                    if (k < c) acc -= x[k] * ((KIND == CS3_LU) ? D[k][c] : D[c][k]);
                x[c] = acc / D[c][c];
                if (KIND == CS3_LU && !(fabs(x[c]) <= inv_tol)) flag_column(status, d.c0 + kb + c);
            }
        }
#pragma unroll
        for (int c = 0; c < BIG_NB; ++c) if (c < bw) F[i + (kb + c) * ld] = x[c];
    } else if (KIND == CS3_LU && g < 2 * below) {  // one column of U12:  L_D u = f, L_D unit lower
        const int j = ke + (g - below);
        double u[BIG_NB];
#pragma unroll
        for (int c = 0; c < BIG_NB; ++c) u[c] = (c < bw) ? F[(kb + c) + j * ld] : 0.0;
#pragma unroll
        for (int c = 0; c < BIG_NB; ++c) {
            if (c < bw) {
                double acc = u[c];
#pragma unroll
                for (int k = 0; k < BIG_NB; ++k)
                    if (k < c) acc -= D[c][k] * u[k];
                u[c] = acc;
            }
        }
#pragma unroll
        for (int c = 0; c < BIG_NB; ++c) if (c < bw) F[(kb + c) + j * ld] = u[c];
    }
}

// Trailing update after panel kb:  F[ke:, ke:] -= L[ke:, kb:ke] * U[kb:ke, ke:]
// 64 x 64 tile per workgroup of 256 threads, 4 x 4 outputs per thread.
template <int KIND>
__global__ void __launch_bounds__(256)
k_big_update(const FrontDesc *__restrict__ fdesc, int first, int kb, double *__restrict__ pool_all,
             long long pool_stride, int batch)
{
    __shared__ double As[BIG_NB][64 + 1];      // As[k][i] = L[ke + ti0 + i, kb + k]
    __shared__ double Bs[BIG_NB][64 + 1];      // Bs[k][j] = U[kb + k, ke + tj0 + j]
    const FrontDesc d = fdesc[first + blockIdx.z / batch];     // grid (tiles, tiles, fronts * batch)
    const int r = d.r, w = d.w;
    if (kb >= w) return;
    const int ke = min(kb + BIG_NB, w), bw = ke - kb;
    const int tiles = (r - ke + 63) / 64;
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bi >= tiles || bj >= tiles) return;
    if (KIND == CS3_CHOLESKY && bi < bj) return;
    double *F = pool_all + (long long) (blockIdx.z % batch) * pool_stride + d.lpan;
    const long long ld = r;
    const int ti0 = ke + bi * 64, tj0 = ke + bj * 64;
    const int tid = threadIdx.x;
    for (int e = tid; e < BIG_NB * 64; e += 256) {
        const int i = e % 64, k = e / 64;
        As[k][i] = (k < bw && ti0 + i < r) ? F[(ti0 + i) + (kb + k) * ld] : 0.0;
    }
    if (KIND == CS3_LU) {
        for (int e = tid; e < BIG_NB * 64; e += 256) {
            const int k = e % BIG_NB, j = e / BIG_NB;
            Bs[k][j] = (k < bw && tj0 + j < r) ? F[(kb + k) + (tj0 + j) * ld] : 0.0;
        }
    } else {
        for (int e = tid; e < BIG_NB * 64; e += 256) {
            const int j = e % 64, k = e / 64;
            Bs[k][j] = (k < bw && tj0 + j < r) ? F[(tj0 + j) + (kb + k) * ld] : 0.0;
        }
    }
    __syncthreads();
    const int tx = tid % 16, ty = tid / 16;
    double acc[4][4] = {};
#pragma unroll 8
    for (int k = 0; k < BIG_NB; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = As[k][tx + 16 * u]; b[u] = Bs[k][ty + 16 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int j = tj0 + ty + 16 * v;
        if (j >= r) continue;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = ti0 + tx + 16 * u;
            if (i < r && (KIND == CS3_LU || i >= j)) F[i + j * ld] -= acc[u][v];
        }
    }
}

// ------------------------------------------------------ supernodal solves --
// X is [n, nrhs] row-major in pivot order.  Each block owns one front and one
// tile of KT right-hand sides (blockIdx.z); lane t of a row handles rhs t.
template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS)
k_solve_fwd(const FrontMeta *__restrict__ meta, const int *__restrict__ sched, int first,
            const int *__restrict__ child_idx, const int *__restrict__ rel_idx,
            const double *__restrict__ pool_all, double *__restrict__ cv_all, double *__restrict__ X_all,
            int nrhs, int KT, long long pool_stride, long long cv_stride, long long x_stride)
{
    extern __shared__ __attribute__((aligned(16))) double v[];
    const int s = sched[first + blockIdx.x];
    const FrontMeta m = meta[s];
    const double *vals = pool_all + (long long) blockIdx.y * pool_stride;
    double *cvp = cv_all + (long long) blockIdx.y * cv_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = m.r, w = m.w;
    const int tid = threadIdx.x, t = tid % KT, i0 = tid / KT, IS = THREADS / KT;
    const int tt = blockIdx.z * KT + t;
    const bool live = tt < nrhs;
    const double *L = vals + m.lpan;

    for (int i = i0; i < r; i += IS)
        v[i * KT + t] = (i < w && live) ? X[(long long) (m.c0 + i) * nrhs + tt] : 0.0;
    __syncthreads();
    for (int cp = m.child_begin; cp < m.child_end; ++cp) {
        const FrontMeta mc = meta[child_idx[cp]];
        const int nbc = mc.r - mc.w;
        const int *rel = rel_idx + mc.rel;
        const double *cv = cvp + mc.cv * nrhs;
        if (live)
            for (int i = i0; i < nbc; i += IS) v[rel[i] * KT + t] += cv[(long long) i * nrhs + tt];
        __syncthreads();
    }
    for (int k = 0; k < w; ++k) {
        if (KIND == CS3_CHOLESKY) {
            if (i0 == 0) v[k * KT + t] /= L[k + (long long) k * r];
            __syncthreads();
        }
        const double xk = v[k * KT + t];
        for (int i = k + 1 + i0; i < r; i += IS) v[i * KT + t] -= L[i + (long long) k * r] * xk;
        __syncthreads();
    }
    if (live) {
        for (int i = i0; i < w; i += IS) X[(long long) (m.c0 + i) * nrhs + tt] = v[i * KT + t];
        if (m.parent >= 0) {
            double *cv = cvp + m.cv * nrhs;
            for (int i = w + i0; i < r; i += IS) cv[(long long) (i - w) * nrhs + tt] = v[i * KT + t];
        }
    }
}

template <int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS)
k_solve_bwd(const FrontMeta *__restrict__ meta, const int *__restrict__ sched, int first,
            const int *__restrict__ st_idx, const double *__restrict__ pool_all,
            double *__restrict__ X_all, int nrhs, int KT, long long pool_stride, long long x_stride)
{
    extern __shared__ __attribute__((aligned(16))) double v[];
    const int s = sched[first + blockIdx.x];
    const FrontMeta m = meta[s];
    const double *vals = pool_all + (long long) blockIdx.y * pool_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = m.r, w = m.w, nb = r - w;
    const int tid = threadIdx.x, t = tid % KT, i0 = tid / KT, IS = THREADS / KT;
    const int tt = blockIdx.z * KT + t;
    const bool live = tt < nrhs;
    const double *L = vals + m.lpan;
    const double *U = vals + m.upan;
    const int *st = st_idx + m.st;

    for (int i = i0; i < r; i += IS) {
        const long long row = (i < w) ? (m.c0 + i) : st[i];
        v[i * KT + t] = live ? X[row * nrhs + tt] : 0.0;
    }
    __syncthreads();
    // pivot rows minus the part that multiplies already-known ancestors
    for (int k = i0; k < w; k += IS) {
        double acc = 0.0;
        if (KIND == CS3_LU) {
            const double *u = U + (long long) k * m.u_sk;
            for (int j = 0; j < nb; ++j) acc += u[(long long) j * m.u_sj] * v[(w + j) * KT + t];
        } else {
            const double *l = L + (long long) k * r + w;
            for (int j = 0; j < nb; ++j) acc += l[j] * v[(w + j) * KT + t];
        }
        v[k * KT + t] -= acc;
    }
    __syncthreads();
    for (int k = w - 1; k >= 0; --k) {
        if (i0 == 0) v[k * KT + t] /= L[k + (long long) k * r];
        __syncthreads();
        const double xk = v[k * KT + t];
        for (int i = i0; i < k; i += IS) {
            const double a = (KIND == CS3_LU) ? L[i + (long long) k * r] : L[k + (long long) i * r];
            v[i * KT + t] -= a * xk;
        }
        __syncthreads();
    }
    if (live)
        for (int i = i0; i < w; i += IS) X[(long long) (m.c0 + i) * nrhs + tt] = v[i * KT + t];
}

// ----------------------------------------------------------- permutations --
// dst[k, :] = src[q[k], :]  (gather) or dst[q[k], :] = src[k, :] (scatter)
__global__ void __launch_bounds__(256)
k_permute_rows(const double *__restrict__ src, double *__restrict__ dst, const int *__restrict__ q,
               long long n, int nrhs, int scatter, long long stride)
{
    const double *s = src + (long long) blockIdx.y * stride;
    double *d = dst + (long long) blockIdx.y * stride;
    const long long total = n * nrhs;
    for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long) gridDim.x * blockDim.x) {
        const long long k = e / nrhs;
        const int t = (int) (e - k * nrhs);
        const long long o = (long long) q[k] * nrhs + t;
        if (scatter) d[o] = s[e]; else d[e] = s[o];
    }
}

// out[p] = map[p] < 0 ? 1.0 : vals[map[p]]
__global__ void __launch_bounds__(256)
k_extract(const double *__restrict__ vals, const long long *__restrict__ map,
          double *__restrict__ out, long long count)
{
    for (long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x; p < count;
         p += (long long) gridDim.x * blockDim.x) {
        const long long o = map[p];
        out[p] = o < 0 ? 1.0 : vals[o];
    }
}

// ------------------------------------- general CSC triangular solve, CSR view --
// One wave per row of the level and rhs tile; lanes split the row's entries,
// partial sums reduced in a fixed butterfly order (reproducible).
__global__ void __launch_bounds__(256)
k_tri_level(const int *__restrict__ rows, int nrows, const int *__restrict__ Rp,
            const int *__restrict__ Rj, const long long *__restrict__ Rmap,
            const long long *__restrict__ diag, const double *__restrict__ Gx,
            double *__restrict__ X, int nrhs)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= nrows) return;
    const int i = rows[wave];
    const int p0 = Rp[i], p1 = Rp[i + 1];
    for (int t = 0; t < nrhs; ++t) {
        double acc = 0.0;
        for (int p = p0 + lane; p < p1; p += 64) acc += Gx[Rmap[p]] * X[(long long) Rj[p] * nrhs + t];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) X[(long long) i * nrhs + t] = (X[(long long) i * nrhs + t] - acc) / Gx[diag[i]];
    }
}

// y = A x with A in CSR-ordered view of the CSC arrays: row i sums its entries
// in ascending column order with separate multiply and add roundings, which
// is the operation order of csc_mat_vec_ff (csc_numba.py:309-328).
__global__ void __launch_bounds__(256)
k_matvec_rows(const int *__restrict__ Rp, const int *__restrict__ Rj,
              const double *__restrict__ Rx, const double *__restrict__ X,
              double *__restrict__ Y, long long m, int nrhs)
{
    const long long total = m * nrhs;
    for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long) gridDim.x * blockDim.x) {
        const long long i = e / nrhs;
        const int t = (int) (e - i * nrhs);
        double y = 0.0;
        for (int p = Rp[i]; p < Rp[i + 1]; ++p)
            y = __dadd_rn(y, __dmul_rn(Rx[p], X[(long long) Rj[p] * nrhs + t]));
        Y[e] = y;
    }
}

// ================================================================ launchers ==
#define CS3_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

static int grid_for(long long work, int block, int cap = 4096)
{
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int) g;
}

template <int KIND>
static hipError_t launch_front_group(const DeviceFactor &D, const LaunchGroup &g, double inv_tol, hipStream_t st)
{
    const unsigned batch = (unsigned) D.batch;
    if (g.cls == FC_BIG) {
        const int chunks = (int) (g.max_asm >> 6);
        const int gx = std::max(1, std::min(64, (chunks + 4 * GATHER_UNROLL - 1) / (4 * GATHER_UNROLL)));
        hipLaunchKernelGGL(k_big_gather, dim3(gx, batch, g.count), dim3(256), 0, st, D.fdesc, g.first,
                           D.asm_src, D.asm_tgt, D.long_src, D.ax, D.pool, D.nnz_a, D.pool_size);
        CS3_LAUNCH_CHECK();
        for (int kb = 0; kb < g.max_w; kb += BIG_NB) {
            // a front narrower than kb + BIG_NB ends its panel early, so size for the largest remainder
            const int below = g.max_r - std::min(kb + 1, g.max_r);
            const int px = std::max(1, (2 * below + 63) / 64);
            hipLaunchKernelGGL((k_big_panel<KIND>), dim3(px, batch, g.count), dim3(64), 0, st, D.fdesc, g.first, kb,
                               D.pool, D.pool_size, inv_tol, D.status);
            CS3_LAUNCH_CHECK();
            const int tiles = (below + 63) / 64;
            if (tiles > 0) {
                hipLaunchKernelGGL((k_big_update<KIND>), dim3(tiles, tiles, g.count * batch), dim3(256), 0, st,
                                   D.fdesc, g.first, kb, D.pool, D.pool_size, (int) batch);
                CS3_LAUNCH_CHECK();
            }
        }
        return hipSuccess;
    }
    dim3 grid((unsigned) g.count, batch);
    const size_t ld = (size_t) (g.max_r | 1);
    const size_t lds = ld * (size_t) g.max_r * sizeof(double);
#define CS3_FRONT_ARGS D.fdesc, g.first, D.asm_src, D.asm_tgt, D.long_src, D.ax, D.pool, D.nnz_a, D.pool_size, inv_tol, D.status
    switch (g.cls) {
    case FC_R16:
        hipLaunchKernelGGL((k_front_lds<KIND, 64, 16>), grid, dim3(64), lds, st, CS3_FRONT_ARGS); break;
    case FC_R32:
        hipLaunchKernelGGL((k_front_lds<KIND, 64, 32>), grid, dim3(64), lds, st, CS3_FRONT_ARGS); break;
    case FC_R64:
        hipLaunchKernelGGL((k_front_lds<KIND, 256, 64>), grid, dim3(256), lds, st, CS3_FRONT_ARGS); break;
    default:
        hipLaunchKernelGGL((k_front_lds<KIND, 512, 64>), grid, dim3(512), lds, st, CS3_FRONT_ARGS); break;
    }
#undef CS3_FRONT_ARGS
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t prepare_kernels()
{
    // the largest LDS-resident class needs more than the default 64 KiB of dynamic LDS
    const int big = 160 * 1024;
    hipError_t e;
    e = hipFuncSetAttribute((const void *) k_front_lds<CS3_LU, 512, 64>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *) k_front_lds<CS3_CHOLESKY, 512, 64>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e != hipSuccess) return e;
    const void *solve_fns[] = {
        (const void *) k_solve_fwd<CS3_LU, 256>, (const void *) k_solve_fwd<CS3_CHOLESKY, 256>,
        (const void *) k_solve_bwd<CS3_LU, 256>, (const void *) k_solve_bwd<CS3_CHOLESKY, 256>,
        (const void *) k_solve_fwd<CS3_LU, 64>, (const void *) k_solve_fwd<CS3_CHOLESKY, 64>,
        (const void *) k_solve_bwd<CS3_LU, 64>, (const void *) k_solve_bwd<CS3_CHOLESKY, 64>};
    for (const void *f : solve_fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_factor_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &groups,
                                double inv_tol, hipStream_t st)
{
    if (D.vals_size > D.big_begin) {     // big-front buffers start from zero: the gather writes only touched entries
        hipError_t e = hipMemset2DAsync(D.pool + D.big_begin, (size_t) D.pool_size * sizeof(double), 0,
                                        (size_t) (D.vals_size - D.big_begin) * sizeof(double), (size_t) D.batch, st);
        if (e != hipSuccess) return e;
    }
    for (const LaunchGroup &g : groups) {
        hipError_t e = (D.kind == CS3_LU) ? launch_front_group<CS3_LU>(D, g, inv_tol, st)
                                          : launch_front_group<CS3_CHOLESKY>(D, g, inv_tol, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int solve_rhs_tile(int nrhs)
{
    int kt = 1;
    while (kt < nrhs && kt < 32) kt <<= 1;
    return kt;
}

template <int KIND>
static hipError_t launch_solve_group(const DeviceFactor &D, const LaunchGroup &g, double *X, int nrhs,
                                     bool forward, hipStream_t st)
{
    int KT = solve_rhs_tile(nrhs);
    while (KT > 1 && (size_t) g.max_r * KT * sizeof(double) > 144 * 1024) KT >>= 1;
    const unsigned tiles = (unsigned) ((nrhs + KT - 1) / KT);
    dim3 grid((unsigned) g.count, (unsigned) D.batch, tiles);
    const size_t lds = (size_t) g.max_r * KT * sizeof(double);
    const long long xs = D.n * (long long) nrhs;
    const long long cvs = D.cv_size * (long long) nrhs;
    const bool small = (long long) g.max_r * KT <= 64;
    if (forward) {
        if (small)
            hipLaunchKernelGGL((k_solve_fwd<KIND, 64>), grid, dim3(64), lds, st, D.meta, D.sched, g.first,
                               D.child_idx, D.rel_idx, D.pool, D.cv, X, nrhs, KT, D.pool_size, cvs, xs);
        else
            hipLaunchKernelGGL((k_solve_fwd<KIND, 256>), grid, dim3(256), lds, st, D.meta, D.sched, g.first,
                               D.child_idx, D.rel_idx, D.pool, D.cv, X, nrhs, KT, D.pool_size, cvs, xs);
    } else {
        if (small)
            hipLaunchKernelGGL((k_solve_bwd<KIND, 64>), grid, dim3(64), lds, st, D.meta, D.sched, g.first,
                               D.st_idx, D.pool, X, nrhs, KT, D.pool_size, xs);
        else
            hipLaunchKernelGGL((k_solve_bwd<KIND, 256>), grid, dim3(256), lds, st, D.meta, D.sched, g.first,
                               D.st_idx, D.pool, X, nrhs, KT, D.pool_size, xs);
    }
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_solve_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &groups,
                               double *X, int nrhs, bool forward, hipStream_t st)
{
    if (forward) {
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            hipError_t e = (D.kind == CS3_LU) ? launch_solve_group<CS3_LU>(D, groups[gi], X, nrhs, true, st)
                                              : launch_solve_group<CS3_CHOLESKY>(D, groups[gi], X, nrhs, true, st);
            if (e != hipSuccess) return e;
        }
    } else {
        for (size_t gi = groups.size(); gi-- > 0; ) {
            hipError_t e = (D.kind == CS3_LU) ? launch_solve_group<CS3_LU>(D, groups[gi], X, nrhs, false, st)
                                              : launch_solve_group<CS3_CHOLESKY>(D, groups[gi], X, nrhs, false, st);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

hipError_t launch_permute(const DeviceFactor &D, const double *src, double *dst, int nrhs, bool scatter,
                          hipStream_t st)
{
    dim3 grid(grid_for(D.n * (long long) nrhs, 256), (unsigned) D.batch);
    hipLaunchKernelGGL(k_permute_rows, grid, dim3(256), 0, st, src, dst, D.q, D.n, nrhs, scatter ? 1 : 0,
                       D.n * (long long) nrhs);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_extract(const double *vals, const long long *map, double *out, long long count, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_extract, dim3(grid_for(count, 256)), dim3(256), 0, st, vals, map, out, count);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_tri_level(const int *rows, int nrows, const int *Rp, const int *Rj, const long long *Rmap,
                            const long long *diag, const double *Gx, double *X, int nrhs, hipStream_t st)
{
    if (nrows == 0) return hipSuccess;
    const int waves_per_block = 4;
    dim3 grid((unsigned) ((nrows + waves_per_block - 1) / waves_per_block));
    hipLaunchKernelGGL(k_tri_level, grid, dim3(64 * waves_per_block), 0, st, rows, nrows, Rp, Rj, Rmap, diag,
                       Gx, X, nrhs);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_matvec_rows(const int *Rp, const int *Rj, const double *Rx, const double *X, double *Y,
                              long long m, int nrhs, hipStream_t st)
{
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_matvec_rows, dim3(grid_for(m * (long long) nrhs, 256)), dim3(256), 0, st, Rp, Rj, Rx,
                       X, Y, m, nrhs);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace cs3
