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

#include "cs3_device.hpp"

namespace cs3 {

// -------------------------------------------------------------- assembly --
// vals[o] = Ax[vsrc[o]] (or 0): one coalesced pass over the panels.
__global__ void __launch_bounds__(256)
k_assemble(const double *__restrict__ Ax, const int *__restrict__ vsrc,
           double *__restrict__ vals, long long vals_size, long long nnz_a)
{
    const double *ax = Ax + (long long) blockIdx.y * nnz_a;
    double *v = vals + (long long) blockIdx.y * vals_size;
    for (long long o = (long long) blockIdx.x * blockDim.x + threadIdx.x; o < vals_size;
         o += (long long) gridDim.x * blockDim.x) {
        const int p = vsrc[o];
        v[o] = p >= 0 ? ax[p] : 0.0;
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
k_front_lds(const FrontMeta *__restrict__ meta, const int *__restrict__ sched, int first,
            const int *__restrict__ child_idx, const int *__restrict__ rel_idx,
            double *__restrict__ vals_all, double *__restrict__ cb_all,
            long long vals_stride, long long cb_stride, double inv_tol, int *status)
{
    extern __shared__ __attribute__((aligned(16))) double F[];
    constexpr int TY = THREADS / TX;
    const int s = sched[first + blockIdx.x];
    const FrontMeta m = meta[s];
    double *vals = vals_all + (long long) blockIdx.y * vals_stride;
    double *cbp = cb_all + (long long) blockIdx.y * cb_stride;
    const int r = m.r, w = m.w, nb = r - w;
    const int ld = r | 1;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    double *L = vals + m.lpan;
    double *U = vals + m.upan;

    // ---- load the assembled panels, clear the Schur block
    for (int j = ty; j < w; j += TY)
        for (int i = tx; i < r; i += TX) F[i + j * ld] = L[i + (long long) j * r];
    if (KIND == CS3_LU) {
        for (int k = ty; k < w; k += TY)
            for (int i = tx; i < nb; i += TX) F[k + (w + i) * ld] = U[i + (long long) k * nb];
    }
    for (int j = w + ty; j < r; j += TY)
        for (int i = w + tx; i < r; i += TX) F[i + j * ld] = 0.0;
    __syncthreads();

    // ---- extend-add the children's contribution blocks, one child at a time
    for (int cp = m.child_begin; cp < m.child_end; ++cp) {
        const FrontMeta mc = meta[child_idx[cp]];
        const int nbc = mc.r - mc.w;
        const int *rel = rel_idx + mc.rel;
        const double *cb = cbp + mc.cb;
        for (int j = ty; j < nbc; j += TY) {
            const int rj = rel[j] * ld;
            for (int i = tx; i < nbc; i += TX) {
                if (KIND == CS3_LU || i >= j) F[rel[i] + rj] += cb[i + (long long) j * nbc];
            }
        }
        __syncthreads();
    }

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
            const double d = sqrt(piv);
            for (int i = k + 1 + tid; i < r; i += THREADS) F[i + k * ld] /= d;
            __syncthreads();
            if (tid == 0) F[k + k * ld] = (piv > 0.0) ? d : -1.0;   // -1 marks "not SPD"
            for (int j = k + 1 + ty; j < r; j += TY) {
                const double u = F[j + k * ld];
                for (int i = j + tx; i < r; i += TX) F[i + j * ld] -= F[i + k * ld] * u;
            }
        }
        __syncthreads();
    }

    // ---- write back: factors to the panels, Schur block to the pool
    for (int j = ty; j < w; j += TY) {
        for (int i = tx; i < r; i += TX) {
            const double v = F[i + j * ld];
            if (KIND == CS3_LU) {
                if (i > j) { if (!(fabs(v) <= inv_tol)) flag_column(status, m.c0 + j); }
                else if (i == j) { if (!(fabs(v) > 0.0) || !(fabs(v) < 1.0e300)) flag_column(status, m.c0 + j); }
            } else if (i == j) {
                if (!(v > 0.0)) flag_column(status, m.c0 + j);
            }
            if (KIND == CS3_LU || i >= j) L[i + (long long) j * r] = v;
        }
    }
    if (KIND == CS3_LU) {
        for (int k = ty; k < w; k += TY)
            for (int i = tx; i < nb; i += TX) U[i + (long long) k * nb] = F[k + (w + i) * ld];
    }
    if (m.parent >= 0) {
        double *cb = cbp + m.cb;
        for (int j = ty; j < nb; j += TY)
            for (int i = tx; i < nb; i += TX)
                if (KIND == CS3_LU || i >= j) cb[i + (long long) j * nb] = F[(w + i) + (w + j) * ld];
    }
}

// ------------------------------------------- front too large for the LDS --
// Same algorithm with the front left in HBM/L2: pivot panels in vals, Schur
// block in its own contribution-block slot.  One workgroup, so ordering
// between steps is the workgroup barrier (which carries a workgroup-scope
// fence; all waves share one L1).
template <int KIND>
struct GlobalFront {
    double *L, *U, *C;
    int r, w, nb;
    __device__ __forceinline__ double &at(int i, int j) const
    {
        if (j < w) return L[i + (long long) j * r];
        if (i < w) return U[(j - w) + (long long) i * nb];
        return C[(i - w) + (long long) (j - w) * nb];
    }
};

template <int KIND, int THREADS, int TX>
__global__ void __launch_bounds__(THREADS)
k_front_big(const FrontMeta *__restrict__ meta, const int *__restrict__ sched, int first,
            const int *__restrict__ child_idx, const int *__restrict__ rel_idx,
            double *vals_all, double *cb_all,
            long long vals_stride, long long cb_stride, double inv_tol, int *status)
{
    constexpr int TY = THREADS / TX;
    const int s = sched[first + blockIdx.x];
    const FrontMeta m = meta[s];
    double *vals = vals_all + (long long) blockIdx.y * vals_stride;
    double *cbp = cb_all + (long long) blockIdx.y * cb_stride;
    const int r = m.r, w = m.w, nb = r - w;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    GlobalFront<KIND> F{vals + m.lpan, vals + m.upan, cbp + m.cb, r, w, nb};

    for (int j = ty; j < nb; j += TY)
        for (int i = tx; i < nb; i += TX) F.C[i + (long long) j * nb] = 0.0;
    __syncthreads();
    for (int cp = m.child_begin; cp < m.child_end; ++cp) {
        const FrontMeta mc = meta[child_idx[cp]];
        const int nbc = mc.r - mc.w;
        const int *rel = rel_idx + mc.rel;
        const double *cb = cbp + mc.cb;
        for (int j = ty; j < nbc; j += TY) {
            const int rj = rel[j];
            for (int i = tx; i < nbc; i += TX) {
                if (KIND == CS3_LU || i >= j) F.at(rel[i], rj) += cb[i + (long long) j * nbc];
            }
        }
        __syncthreads();
    }
    for (int k = 0; k < w; ++k) {
        const double piv = F.at(k, k);
        if (KIND == CS3_LU) {
            __syncthreads();
            for (int i = k + 1 + tid; i < r; i += THREADS) F.at(i, k) /= piv;
            __syncthreads();
            for (int j = k + 1 + ty; j < r; j += TY) {
                const double u = F.at(k, j);
                for (int i = k + 1 + tx; i < r; i += TX) F.at(i, j) -= F.at(i, k) * u;
            }
        } else {
            const double d = sqrt(piv);
            __syncthreads();
            for (int i = k + 1 + tid; i < r; i += THREADS) F.at(i, k) /= d;
            if (tid == 0) F.at(k, k) = (piv > 0.0) ? d : -1.0;
            __syncthreads();
            for (int j = k + 1 + ty; j < r; j += TY) {
                const double u = F.at(j, k);
                for (int i = j + tx; i < r; i += TX) F.at(i, j) -= F.at(i, k) * u;
            }
        }
        __syncthreads();
    }
    for (int j = ty; j < w; j += TY) {
        for (int i = tx; i < r; i += TX) {
            const double v = F.at(i, j);
            if (KIND == CS3_LU) {
                if (i > j) { if (!(fabs(v) <= inv_tol)) flag_column(status, m.c0 + j); }
                else if (i == j) { if (!(fabs(v) > 0.0) || !(fabs(v) < 1.0e300)) flag_column(status, m.c0 + j); }
            } else if (i == j) {
                if (!(v > 0.0)) flag_column(status, m.c0 + j);
            }
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
            const double *__restrict__ vals_all, double *__restrict__ cv_all, double *__restrict__ X_all,
            int nrhs, int KT, long long vals_stride, long long cv_stride, long long x_stride)
{
    extern __shared__ __attribute__((aligned(16))) double v[];
    const int s = sched[first + blockIdx.x];
    const FrontMeta m = meta[s];
    const double *vals = vals_all + (long long) blockIdx.y * vals_stride;
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
            const int *__restrict__ st_idx, const double *__restrict__ vals_all,
            double *__restrict__ X_all, int nrhs, int KT, long long vals_stride, long long x_stride)
{
    extern __shared__ __attribute__((aligned(16))) double v[];
    const int s = sched[first + blockIdx.x];
    const FrontMeta m = meta[s];
    const double *vals = vals_all + (long long) blockIdx.y * vals_stride;
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
            const double *u = U + (long long) k * nb;
            for (int j = 0; j < nb; ++j) acc += u[j] * v[(w + j) * KT + t];
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

hipError_t launch_assemble(const DeviceFactor &D, const double *Ax_dev, hipStream_t st)
{
    dim3 grid(grid_for(D.vals_size, 256), (unsigned) D.batch);
    hipLaunchKernelGGL(k_assemble, grid, dim3(256), 0, st, Ax_dev, D.vsrc, D.vals, D.vals_size, D.nnz_a);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

template <int KIND>
static hipError_t launch_front_group(const DeviceFactor &D, const LaunchGroup &g, double inv_tol, hipStream_t st)
{
    dim3 grid((unsigned) g.count, (unsigned) D.batch);
    const size_t ld = (size_t) (g.max_r | 1);
    const size_t lds = ld * (size_t) g.max_r * sizeof(double);
#define CS3_FRONT_ARGS D.meta, D.sched, g.first, D.child_idx, D.rel_idx, D.vals, D.cb, D.vals_size, D.cb_size, inv_tol, D.status
    switch (g.cls) {
    case FC_R16:
        hipLaunchKernelGGL((k_front_lds<KIND, 64, 16>), grid, dim3(64), lds, st, CS3_FRONT_ARGS); break;
    case FC_R32:
        hipLaunchKernelGGL((k_front_lds<KIND, 64, 32>), grid, dim3(64), lds, st, CS3_FRONT_ARGS); break;
    case FC_R64:
        hipLaunchKernelGGL((k_front_lds<KIND, 256, 64>), grid, dim3(256), lds, st, CS3_FRONT_ARGS); break;
    case FC_LDS:
        hipLaunchKernelGGL((k_front_lds<KIND, 512, 64>), grid, dim3(512), lds, st, CS3_FRONT_ARGS); break;
    default:
        hipLaunchKernelGGL((k_front_big<KIND, 1024, 64>), grid, dim3(1024), 0, st, CS3_FRONT_ARGS); break;
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
                               D.child_idx, D.rel_idx, D.vals, D.cv, X, nrhs, KT, D.vals_size, cvs, xs);
        else
            hipLaunchKernelGGL((k_solve_fwd<KIND, 256>), grid, dim3(256), lds, st, D.meta, D.sched, g.first,
                               D.child_idx, D.rel_idx, D.vals, D.cv, X, nrhs, KT, D.vals_size, cvs, xs);
    } else {
        if (small)
            hipLaunchKernelGGL((k_solve_bwd<KIND, 64>), grid, dim3(64), lds, st, D.meta, D.sched, g.first,
                               D.st_idx, D.vals, X, nrhs, KT, D.vals_size, xs);
        else
            hipLaunchKernelGGL((k_solve_bwd<KIND, 256>), grid, dim3(256), lds, st, D.meta, D.sched, g.first,
                               D.st_idx, D.vals, X, nrhs, KT, D.vals_size, xs);
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
