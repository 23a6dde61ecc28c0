// Device versions of the reference's substrate kernels: the callers and data
// formats either side of the factor/solve path.  Every function reproduces the reference's OUTPUT ORDER
// and arithmetic order exactly (csc_numba.py, cited per function), so integer outputs and values are
// bit-identical with the golden vectors under tests/golden/.
//
// The reference fills buckets sequentially; here a bucket (a row of the transpose, a column of the
// assembled CSC) is filled with atomic cursors in arbitrary order and then SORTED BY SOURCE POSITION:
// the reference visits sources in ascending position, so ascending position IS its output order
// (duplicates included).  No floating-point value is ever combined in a data-dependent order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/csparse3_amd.h"
#include "cs3_internal.hpp"

#pragma clang fp contract(off)          // a * b + c stays two roundings, as in the reference's interpreter / JIT

namespace cs3 {

// ------------------------------------------------------------------ building blocks --
__global__ void __launch_bounds__(256)
k_histogram(const int *__restrict__ key, long long count, int nbucket, int *__restrict__ hist, int *bad)
{
    for (long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x; p < count; p += (long long) gridDim.x * blockDim.x) {
        const int k = key[p];
        if (k < 0 || k >= nbucket) { *bad = 1; continue; }
        atomicAdd(&hist[k + 1], 1);                          // integer counts: order-independent
    }
}

// In-place inclusive scan of ptr[1 .. n] (ptr[0] = 0 stays): one workgroup of 1024 threads walks the array in tiles of
// 4096 entries -- coalesced loads, four entries per thread, wave-shuffle scans, one LDS exchange per tile -- and carries
// the running total from tile to tile.  (Counts are integers: any association gives the same sums.)
__global__ void __launch_bounds__(1024)
k_scan(int *ptr, long long n)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 1; base <= n; base += 4096) {
        const long long i0 = base + 4LL * tid;
        long long v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (i0 + q <= n) ? ptr[i0 + q] : 0;
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        long long incl = v[3];
        for (int off = 1; off < 64; off <<= 1) { const long long o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        long long before = carry_s;
        for (int w = 0; w < wv; ++w) before += wsum[w];
        const long long excl = before + incl - v[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) if (i0 + q <= n) ptr[i0 + q] = (int) (excl + v[q]);
        __syncthreads();
        if (tid == 1023) carry_s = before + incl;
        __syncthreads();
    }
}

// slot[cursor of bucket]++ = source position
__global__ void __launch_bounds__(256)
k_bucket_fill(const int *__restrict__ key, long long count, const int *__restrict__ ptr, int *__restrict__ cursor,
              int *__restrict__ slot)
{
    for (long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x; p < count; p += (long long) gridDim.x * blockDim.x) {
        const int k = key[p];
        const int q = ptr[k] + atomicAdd(&cursor[k], 1);
        slot[q] = (int) p;
    }
}

// ascending source positions inside every bucket: insertion sort for short buckets, heap sort beyond
__device__ void sort_ints(int *a, int len)
{
    if (len <= 32) {
        for (int i = 1; i < len; ++i) {
            const int v = a[i];
            int j = i - 1;
            while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
            a[j + 1] = v;
        }
        return;
    }
    auto sift = [&](int root, int end) {
        for (;;) {
            int child = 2 * root + 1;
            if (child > end) return;
            if (child + 1 <= end && a[child] < a[child + 1]) ++child;
            if (a[root] >= a[child]) return;
            const int t = a[root]; a[root] = a[child]; a[child] = t;
            root = child;
        }
    };
    for (int s = (len - 2) / 2; s >= 0; --s) sift(s, len - 1);
    for (int e = len - 1; e > 0; --e) { const int t = a[0]; a[0] = a[e]; a[e] = t; sift(0, e - 1); }
}

__global__ void __launch_bounds__(256)
k_bucket_sort(const int *__restrict__ ptr, int nbucket, int *__restrict__ slot)
{
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nbucket; b += gridDim.x * blockDim.x)
        sort_ints(slot + ptr[b], ptr[b + 1] - ptr[b]);
}

// column of every CSC position
__global__ void __launch_bounds__(256)
k_expand_columns(const int *__restrict__ Ap, int n, int *__restrict__ col)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) col[p] = j;
}

__global__ void __launch_bounds__(256)
k_gather_pairs(const int *__restrict__ slot, long long count, const int *__restrict__ idx_src, const double *__restrict__ val_src,
               int *__restrict__ idx_out, double *__restrict__ val_out)
{
    for (long long q = (long long) blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (long long) gridDim.x * blockDim.x) {
        const int p = slot[q];
        idx_out[q] = idx_src[p];
        val_out[q] = val_src[p];
    }
}

// ------------------------------------------------------------------------- csc_norm --
// csc_norm (csc_numba.py:723-739): max over columns of the sum of |x| taken in storage order.
__global__ void __launch_bounds__(256)
k_col_abs_sums(const int *__restrict__ Ap, const double *__restrict__ Ax, int n, double *__restrict__ sums)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) s += fabs(Ax[p]);
        sums[j] = s;
    }
}

// max is exact in any order; the scan order of the reference (norm = max(norm, s)) matters only for NaN,
// which compares false there as here
__global__ void __launch_bounds__(1024)
k_max_reduce(const double *__restrict__ v, int n, double *out)
{
    __shared__ double part[1024];
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) m = (v[i] > m) ? v[i] : m;
    part[threadIdx.x] = m;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int) threadIdx.x < off) part[threadIdx.x] = (part[threadIdx.x + off] > part[threadIdx.x]) ? part[threadIdx.x + off] : part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = part[0];
}

// -------------------------------------------------------------------------- csc_add --
// csc_add_ff (csc_numba.py:183-219): column j of C = scatter(alpha A(:, j)) then scatter(beta B(:, j))
// (csc_scatter_f :125-151: first touch of a row appends it and stores beta * x, a later touch adds).
// One thread per column; membership by linear search over the column built so far.
__global__ void __launch_bounds__(128)
k_add_columns(int n, const int *__restrict__ Ap, const int *__restrict__ Ai, const double *__restrict__ Ax,
              const int *__restrict__ Bp, const int *__restrict__ Bi, const double *__restrict__ Bx,
              double alpha, double beta, const int *__restrict__ Cp, int *__restrict__ Ci, double *__restrict__ Cx,
              int *__restrict__ count)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const int base = Cp ? Cp[j] : 0;
        int nz = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const int *Pp = pass ? Bp : Ap, *Pi = pass ? Bi : Ai;
            const double *Px = pass ? Bx : Ax;
            const double f = pass ? beta : alpha;
            for (int p = Pp[j]; p < Pp[j + 1]; ++p) {
                const int i = Pi[p];
                int hit = -1;
                if (Cp) {
                    for (int q = 0; q < nz; ++q) if (Ci[base + q] == i) { hit = q; break; }
                    if (hit < 0) { Ci[base + nz] = i; Cx[base + nz] = f * Px[p]; ++nz; }
                    else Cx[base + hit] = Cx[base + hit] + f * Px[p];
                } else {
                    // counting pass: rows seen earlier in this column = earlier entries of A(:, j) / B(:, j)
                    for (int q = Ap[j]; q < (pass ? Ap[j + 1] : p) && hit < 0; ++q) if (Ai[q] == i) hit = 0;
                    if (pass) for (int q = Bp[j]; q < p && hit < 0; ++q) if (Bi[q] == i) hit = 0;
                    if (hit < 0) ++nz;
                }
            }
        }
        if (!Cp) count[j + 1] = nz;
    }
}

// ------------------------------------------------------------------- csc_sub_matrix --
// csc_sub_matrix (csc_numba.py:464-502) INCLUDING its row numbering: for a selected column the reference walks the
// selected rows in order and, inside a row, the column's entries in storage order; the new row index of a match is a
// running counter that advances on every match and is bumped from 0 to 1 after a selected row without a match.  Once
// the counter is positive it is never bumped again, so: the t-th match of the column (t from 0) gets index
// t + (1 if the FIRST selected row has no match in this column else 0).  That closed form makes the loop parallel: one
// wave per selected column, lane = position in the row selection, matches counted per lane, wave prefix sums in
// selection order.  (The one-thread-per-column transcription of the loop took 2.5 ms for a 2000 x 2000 selection.)
__global__ void __launch_bounds__(256)
k_sub_matrix(const int *__restrict__ Ap, const int *__restrict__ Ai, const double *__restrict__ Ax,
             const int *__restrict__ rows, int nrows, const int *__restrict__ cols, int ncols,
             const int *__restrict__ Bp, int *__restrict__ Bi, double *__restrict__ Bx, int *__restrict__ count)
{
    const int lane = threadIdx.x & 63;
    for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < ncols; c += (gridDim.x * blockDim.x) >> 6) {
        const int j = cols[c], p0 = Ap[j], p1 = Ap[j + 1];
        const int base = Bp ? Bp[c] : 0;
        // does the first selected row match anything in this column?  (decides the numbering offset)
        int bump = 0;
        if (nrows > 0) {
            const int r0 = rows[0];
            bool hit = false;
            for (int k = p0 + lane; k < p1; k += 64) hit |= Ai[k] == r0;
            bump = __any(hit) ? 0 : 1;
        }
        int carried = 0;                                            // matches in earlier chunks of the selection
        for (int rr0 = 0; rr0 < nrows; rr0 += 64) {
            const int rr = rr0 + lane;
            const int r = rr < nrows ? rows[rr] : -1;
            int mine = 0;
            for (int k = p0; k < p1; ++k) mine += (rr < nrows && Ai[k] == r) ? 1 : 0;
            int incl = mine;
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
            if (Bp && mine) {
                int t = carried + incl - mine;                      // matches before mine, in (row position, entry) order
                for (int k = p0; k < p1; ++k)
                    if (Ai[k] == r) { Bx[base + t] = Ax[k]; Bi[base + t] = t + bump; ++t; }
            }
            carried += __shfl(incl, 63);
        }
        if (!Bp && lane == 0) count[c + 1] = carried;
    }
}

// --------------------------------------------------------------------- find_islands --
// find_islands (csc_numba.py:744-808) walks the graph of the pattern from every unvisited node in
// ascending order, so islands come out ordered by their smallest node; CscMat.islands then sorts each
// (csc.py:520-521).  Here: label[i] = smallest node of i's component, by min-label propagation over the
// symmetrised pattern: every entry hooks the larger of its two ROOTS under the smaller, then all paths are
// compressed; a component's smallest node is never hooked, so it ends up as everybody's label.
__global__ void __launch_bounds__(256)
k_label_init(int *label, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) label[i] = i; }

__global__ void __launch_bounds__(256)
k_label_hook(const int *__restrict__ Ap, const int *__restrict__ Ai, int n, int *label, int *changed)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
            const int i = Ai[p];
            if (i < 0 || i >= n) continue;
            const int li = label[i], lj = label[j];        // roots: every label is compressed between rounds
            if (li < lj) { atomicMin(&label[lj], li); *changed = 1; }      // hook the larger root under the smaller
            else if (lj < li) { atomicMin(&label[li], lj); *changed = 1; }
        }
    }
}

__global__ void __launch_bounds__(256)
k_label_jump(int *label, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int l = label[i];
        while (label[l] != l) l = label[l];
        label[i] = l;
    }
}

// ---- exact find_islands on patterns that are NOT structurally symmetric ------------------------------------------
// The reference opens an island at the smallest unvisited node s and puts in it every unvisited node reachable from s
// along column -> row edges (v -> indices[indptr[v] .. indptr[v + 1]), csc_numba.py:768-800).  The visited set is closed
// under successors, so node i ends up in the island of  m(i) = the smallest node that reaches i  (itself included):
// nothing smaller reaches m(i), so the loop finds it unvisited and starts there, and no earlier start reaches i.
// m is the least fixed point of  label[k] <- min(label[k], label[v])  over the edges v -> k; label[i] <- label[label[i]]
// is valid too (reachability is transitive) and makes chains converge in O(log n) rounds instead of O(n).
// On a structurally symmetric pattern m(i) is the smallest node of i's connected component: the root-hooking kernels
// above give the same labels in fewer rounds and are used there.
__global__ void __launch_bounds__(256)
k_pattern_symmetric(const int *__restrict__ Ap, const int *__restrict__ Ai, int n, int *unsym)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
            const int i = Ai[p];
            if (i < 0 || i >= n || i == j) continue;
            bool found = false;
            for (int q = Ap[i]; q < Ap[i + 1] && !found; ++q) found = Ai[q] == j;
            if (!found) { *unsym = 1; return; }
        }
}

__global__ void __launch_bounds__(256)
k_reach_hook(const int *__restrict__ Ap, const int *__restrict__ Ai, int n, int *label, int *changed)
{
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        const int lv = label[v];
        for (int p = Ap[v]; p < Ap[v + 1]; ++p) {
            const int k = Ai[p];
            if (k < 0 || k >= n) continue;
            if (lv < label[k]) { atomicMin(&label[k], lv); *changed = 1; }
        }
    }
}

__global__ void __launch_bounds__(256)
k_reach_jump(int *label, int n, int *changed)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int l = label[i], ll = label[l];
        if (ll < l) { atomicMin(&label[i], ll); *changed = 1; }
    }
}

static unsigned blocks_for(long long work, int block)
{
    return (unsigned) std::max<long long>(1, std::min<long long>((work + block - 1) / block, 4096));
}

}  // namespace cs3

using namespace cs3;

namespace {

// device scratch that frees itself
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 8)); }
    template <class T> T *as() { return static_cast<T *>(p); }
};

int no_device(const char *who)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error(std::string("no HIP device visible: ") + who + " runs on the GPU only");
        return 1;
    }
    return 0;
}

#define SUB_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_));               \
            return CS3_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

// buckets[key[p]] gets p, every bucket in ascending p: ptr[nbucket + 1] and slot[count] on the device
int bucket_by_key(const int *d_key, long long count, int nbucket, int *d_ptr, int *d_slot, const char *who)
{
    DevBuf cursor, bad;
    SUB_HIP(cursor.alloc((size_t) nbucket * sizeof(int)));
    SUB_HIP(bad.alloc(sizeof(int)));
    SUB_HIP(hipMemset(cursor.p, 0, std::max<size_t>((size_t) nbucket * sizeof(int), 8)));
    SUB_HIP(hipMemset(bad.p, 0, sizeof(int)));
    SUB_HIP(hipMemset(d_ptr, 0, (size_t) (nbucket + 1) * sizeof(int)));
    if (count) hipLaunchKernelGGL(k_histogram, dim3(blocks_for(count, 256)), dim3(256), 0, 0, d_key, count, nbucket, d_ptr, bad.as<int>());
    int h_bad = 0;
    SUB_HIP(hipMemcpy(&h_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost));
    if (h_bad) { set_error(std::string(who) + ": index out of range"); return CS3_ERR_ARG; }
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, 0, d_ptr, (long long) nbucket);
    if (count) {
        hipLaunchKernelGGL(k_bucket_fill, dim3(blocks_for(count, 256)), dim3(256), 0, 0, d_key, count, d_ptr, cursor.as<int>(), d_slot);
        hipLaunchKernelGGL(k_bucket_sort, dim3(blocks_for(nbucket, 256)), dim3(256), 0, 0, d_ptr, nbucket, d_slot);
    }
    SUB_HIP(hipGetLastError());
    return CS3_OK;
}

}  // namespace

extern "C" {

// C = A' (csc_transpose, csc_numba.py:400-436) = the CSR arrays of A (csc_to_csr, :360-397): rows of A
// become columns of C, entries of a row in ascending column order, duplicates in storage order.
int cs3_csc_transpose(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                      int32_t *Cp, int32_t *Ci, double *Cx)
{
    if (m < 0 || n < 0 || m > INT_MAX || n > INT_MAX || !Ap || !Cp) { set_error("cs3_csc_transpose: bad argument"); return CS3_ERR_ARG; }
    if (no_device("cs3_csc_transpose")) return CS3_ERR_HIP;
    const long long nnz = Ap[n];
    if (nnz < 0 || (nnz > 0 && (!Ai || !Ax || !Ci || !Cx))) { set_error("cs3_csc_transpose: bad argument"); return CS3_ERR_ARG; }
    DevBuf ap, ai, ax, col, ptr, slot, ci, cx;
    SUB_HIP(ap.alloc((size_t) (n + 1) * 4)); SUB_HIP(ai.alloc((size_t) nnz * 4)); SUB_HIP(ax.alloc((size_t) nnz * 8));
    SUB_HIP(col.alloc((size_t) nnz * 4)); SUB_HIP(ptr.alloc((size_t) (m + 1) * 4)); SUB_HIP(slot.alloc((size_t) nnz * 4));
    SUB_HIP(ci.alloc((size_t) nnz * 4)); SUB_HIP(cx.alloc((size_t) nnz * 8));
    SUB_HIP(hipMemcpy(ap.p, Ap, (size_t) (n + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) {
        SUB_HIP(hipMemcpy(ai.p, Ai, (size_t) nnz * 4, hipMemcpyHostToDevice));
        SUB_HIP(hipMemcpy(ax.p, Ax, (size_t) nnz * 8, hipMemcpyHostToDevice));
    }
    int rc = bucket_by_key(ai.as<int>(), nnz, (int) m, ptr.as<int>(), slot.as<int>(), "cs3_csc_transpose");
    if (rc) return rc;
    if (nnz) {
        hipLaunchKernelGGL(k_expand_columns, dim3(blocks_for(n, 256)), dim3(256), 0, 0, ap.as<int>(), (int) n, col.as<int>());
        hipLaunchKernelGGL(k_gather_pairs, dim3(blocks_for(nnz, 256)), dim3(256), 0, 0, slot.as<int>(), nnz, col.as<int>(),
                           ax.as<double>(), ci.as<int>(), cx.as<double>());
        SUB_HIP(hipGetLastError());
        SUB_HIP(hipMemcpy(Ci, ci.p, (size_t) nnz * 4, hipMemcpyDeviceToHost));
        SUB_HIP(hipMemcpy(Cx, cx.p, (size_t) nnz * 8, hipMemcpyDeviceToHost));
    }
    SUB_HIP(hipMemcpy(Cp, ptr.p, (size_t) (m + 1) * 4, hipMemcpyDeviceToHost));
    return CS3_OK;
}

// coo_to_csc (csc_numba.py:331-357): triplets to CSC, entries of a column in triplet order, duplicates kept.
int cs3_coo_to_csc(int64_t m, int64_t n, int64_t nz, const int32_t *Ti, const int32_t *Tj, const double *Tx,
                   int32_t *Cp, int32_t *Ci, double *Cx)
{
    if (m < 0 || n < 0 || nz < 0 || n > INT_MAX || nz > INT_MAX || !Cp || (nz > 0 && (!Ti || !Tj || !Tx || !Ci || !Cx))) {
        set_error("cs3_coo_to_csc: bad argument"); return CS3_ERR_ARG;
    }
    if (no_device("cs3_coo_to_csc")) return CS3_ERR_HIP;
    DevBuf ti, tj, tx, ptr, slot, ci, cx;
    SUB_HIP(ti.alloc((size_t) nz * 4)); SUB_HIP(tj.alloc((size_t) nz * 4)); SUB_HIP(tx.alloc((size_t) nz * 8));
    SUB_HIP(ptr.alloc((size_t) (n + 1) * 4)); SUB_HIP(slot.alloc((size_t) nz * 4));
    SUB_HIP(ci.alloc((size_t) nz * 4)); SUB_HIP(cx.alloc((size_t) nz * 8));
    if (nz) {
        SUB_HIP(hipMemcpy(ti.p, Ti, (size_t) nz * 4, hipMemcpyHostToDevice));
        SUB_HIP(hipMemcpy(tj.p, Tj, (size_t) nz * 4, hipMemcpyHostToDevice));
        SUB_HIP(hipMemcpy(tx.p, Tx, (size_t) nz * 8, hipMemcpyHostToDevice));
    }
    int rc = bucket_by_key(tj.as<int>(), nz, (int) n, ptr.as<int>(), slot.as<int>(), "cs3_coo_to_csc");
    if (rc) return rc;
    if (nz) {
        hipLaunchKernelGGL(k_gather_pairs, dim3(blocks_for(nz, 256)), dim3(256), 0, 0, slot.as<int>(), (long long) nz, ti.as<int>(),
                           tx.as<double>(), ci.as<int>(), cx.as<double>());
        SUB_HIP(hipGetLastError());
        SUB_HIP(hipMemcpy(Ci, ci.p, (size_t) nz * 4, hipMemcpyDeviceToHost));
        SUB_HIP(hipMemcpy(Cx, cx.p, (size_t) nz * 8, hipMemcpyDeviceToHost));
    }
    SUB_HIP(hipMemcpy(Cp, ptr.p, (size_t) (n + 1) * 4, hipMemcpyDeviceToHost));
    return CS3_OK;
}

// csc_norm (csc_numba.py:723-739): 1-norm.
int cs3_csc_norm(int64_t n, const int32_t *Ap, const double *Ax, double *norm)
{
    if (n < 0 || n > INT_MAX || !Ap || !norm) { set_error("cs3_csc_norm: bad argument"); return CS3_ERR_ARG; }
    if (no_device("cs3_csc_norm")) return CS3_ERR_HIP;
    const long long nnz = Ap[n];
    if (nnz > 0 && !Ax) { set_error("cs3_csc_norm: bad argument"); return CS3_ERR_ARG; }
    DevBuf ap, ax, sums, out;
    SUB_HIP(ap.alloc((size_t) (n + 1) * 4)); SUB_HIP(ax.alloc((size_t) nnz * 8)); SUB_HIP(sums.alloc((size_t) n * 8)); SUB_HIP(out.alloc(8));
    SUB_HIP(hipMemcpy(ap.p, Ap, (size_t) (n + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) SUB_HIP(hipMemcpy(ax.p, Ax, (size_t) nnz * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_col_abs_sums, dim3(blocks_for(n, 256)), dim3(256), 0, 0, ap.as<int>(), ax.as<double>(), (int) n, sums.as<double>());
    hipLaunchKernelGGL(k_max_reduce, dim3(1), dim3(1024), 0, 0, sums.as<double>(), (int) n, out.as<double>());
    SUB_HIP(hipGetLastError());
    SUB_HIP(hipMemcpy(norm, out.p, 8, hipMemcpyDeviceToHost));
    return CS3_OK;
}

// C = alpha A + beta B (csc_add_ff, csc_numba.py:183-219).  Ci / Cx hold at least nnz(A) + nnz(B) entries;
// the number used is Cp[n].
int cs3_csc_add(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                const int32_t *Bp, const int32_t *Bi, const double *Bx, double alpha, double beta,
                int32_t *Cp, int32_t *Ci, double *Cx)
{
    if (m < 0 || n < 0 || n > INT_MAX || !Ap || !Bp || !Cp) { set_error("cs3_csc_add: bad argument"); return CS3_ERR_ARG; }
    if (no_device("cs3_csc_add")) return CS3_ERR_HIP;
    const long long na = Ap[n], nb = Bp[n];
    if ((na > 0 && (!Ai || !Ax)) || (nb > 0 && (!Bi || !Bx)) || (na + nb > 0 && (!Ci || !Cx))) { set_error("cs3_csc_add: bad argument"); return CS3_ERR_ARG; }
    DevBuf ap, ai, ax, bp, bi, bx, cp, ci, cx;
    SUB_HIP(ap.alloc((size_t) (n + 1) * 4)); SUB_HIP(ai.alloc((size_t) na * 4)); SUB_HIP(ax.alloc((size_t) na * 8));
    SUB_HIP(bp.alloc((size_t) (n + 1) * 4)); SUB_HIP(bi.alloc((size_t) nb * 4)); SUB_HIP(bx.alloc((size_t) nb * 8));
    SUB_HIP(cp.alloc((size_t) (n + 1) * 4));
    SUB_HIP(hipMemcpy(ap.p, Ap, (size_t) (n + 1) * 4, hipMemcpyHostToDevice));
    SUB_HIP(hipMemcpy(bp.p, Bp, (size_t) (n + 1) * 4, hipMemcpyHostToDevice));
    if (na) { SUB_HIP(hipMemcpy(ai.p, Ai, (size_t) na * 4, hipMemcpyHostToDevice)); SUB_HIP(hipMemcpy(ax.p, Ax, (size_t) na * 8, hipMemcpyHostToDevice)); }
    if (nb) { SUB_HIP(hipMemcpy(bi.p, Bi, (size_t) nb * 4, hipMemcpyHostToDevice)); SUB_HIP(hipMemcpy(bx.p, Bx, (size_t) nb * 8, hipMemcpyHostToDevice)); }
    SUB_HIP(hipMemset(cp.p, 0, (size_t) (n + 1) * 4));
    hipLaunchKernelGGL(k_add_columns, dim3(blocks_for(n, 128)), dim3(128), 0, 0, (int) n, ap.as<int>(), ai.as<int>(), ax.as<double>(),
                       bp.as<int>(), bi.as<int>(), bx.as<double>(), alpha, beta, (const int *) nullptr, (int *) nullptr,
                       (double *) nullptr, cp.as<int>());
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, 0, cp.as<int>(), (long long) n);
    SUB_HIP(hipGetLastError());
    SUB_HIP(hipMemcpy(Cp, cp.p, (size_t) (n + 1) * 4, hipMemcpyDeviceToHost));
    const long long nc = Cp[n];
    SUB_HIP(ci.alloc((size_t) nc * 4)); SUB_HIP(cx.alloc((size_t) nc * 8));
    hipLaunchKernelGGL(k_add_columns, dim3(blocks_for(n, 128)), dim3(128), 0, 0, (int) n, ap.as<int>(), ai.as<int>(), ax.as<double>(),
                       bp.as<int>(), bi.as<int>(), bx.as<double>(), alpha, beta, cp.as<int>(), ci.as<int>(), cx.as<double>(),
                       (int *) nullptr);
    SUB_HIP(hipGetLastError());
    if (nc) {
        SUB_HIP(hipMemcpy(Ci, ci.p, (size_t) nc * 4, hipMemcpyDeviceToHost));
        SUB_HIP(hipMemcpy(Cx, cx.p, (size_t) nc * 8, hipMemcpyDeviceToHost));
    }
    return CS3_OK;
}

// B = A[rows, cols] with the reference's semantics (csc_sub_matrix, csc_numba.py:464-502); Bi / Bx hold b_cap
// entries (the reference allocates nnz(A)), the number used is Bp[ncols].  With repeated rows or columns the
// result can outgrow nnz(A): the reference then runs off its arrays, this function fills Bp, writes nothing
// else and returns CS3_ERR_ARG, so the caller can retry with Bp[ncols] entries.
int cs3_csc_sub_matrix(int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                       const int32_t *rows, int64_t nrows, const int32_t *cols, int64_t ncols,
                       int32_t *Bp, int32_t *Bi, double *Bx, int64_t b_cap)
{
    if (n < 0 || n > INT_MAX || nrows < 0 || ncols < 0 || nrows > INT_MAX || ncols > INT_MAX || !Ap || !Bp ||
        (nrows > 0 && !rows) || (ncols > 0 && !cols)) { set_error("cs3_csc_sub_matrix: bad argument"); return CS3_ERR_ARG; }
    for (int64_t c = 0; c < ncols; ++c)
        if (cols[c] < 0 || cols[c] >= n) { set_error("cs3_csc_sub_matrix: column index out of range"); return CS3_ERR_ARG; }
    if (no_device("cs3_csc_sub_matrix")) return CS3_ERR_HIP;
    const long long nnz = Ap[n];
    DevBuf ap, ai, ax, dr, dc, bp, bi, bx;
    SUB_HIP(ap.alloc((size_t) (n + 1) * 4)); SUB_HIP(ai.alloc((size_t) nnz * 4)); SUB_HIP(ax.alloc((size_t) nnz * 8));
    SUB_HIP(dr.alloc((size_t) nrows * 4)); SUB_HIP(dc.alloc((size_t) ncols * 4)); SUB_HIP(bp.alloc((size_t) (ncols + 1) * 4));
    SUB_HIP(hipMemcpy(ap.p, Ap, (size_t) (n + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) { SUB_HIP(hipMemcpy(ai.p, Ai, (size_t) nnz * 4, hipMemcpyHostToDevice)); SUB_HIP(hipMemcpy(ax.p, Ax, (size_t) nnz * 8, hipMemcpyHostToDevice)); }
    if (nrows) SUB_HIP(hipMemcpy(dr.p, rows, (size_t) nrows * 4, hipMemcpyHostToDevice));
    if (ncols) SUB_HIP(hipMemcpy(dc.p, cols, (size_t) ncols * 4, hipMemcpyHostToDevice));
    SUB_HIP(hipMemset(bp.p, 0, (size_t) (ncols + 1) * 4));
    hipLaunchKernelGGL(k_sub_matrix, dim3(blocks_for((long long) ncols * 64, 256)), dim3(256), 0, 0, ap.as<int>(), ai.as<int>(), ax.as<double>(),
                       dr.as<int>(), (int) nrows, dc.as<int>(), (int) ncols, (const int *) nullptr, (int *) nullptr,
                       (double *) nullptr, bp.as<int>());
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, 0, bp.as<int>(), (long long) ncols);
    SUB_HIP(hipGetLastError());
    SUB_HIP(hipMemcpy(Bp, bp.p, (size_t) (ncols + 1) * 4, hipMemcpyDeviceToHost));
    const long long nb = Bp[ncols];
    if (nb > b_cap || (nb > 0 && (!Bi || !Bx))) {
        set_error("cs3_csc_sub_matrix: result has " + std::to_string(nb) + " entries, room for " + std::to_string(b_cap) +
                  " (repeated rows / columns?)");
        return CS3_ERR_ARG;
    }
    SUB_HIP(bi.alloc((size_t) nb * 4)); SUB_HIP(bx.alloc((size_t) nb * 8));
    hipLaunchKernelGGL(k_sub_matrix, dim3(blocks_for((long long) ncols * 64, 256)), dim3(256), 0, 0, ap.as<int>(), ai.as<int>(), ax.as<double>(),
                       dr.as<int>(), (int) nrows, dc.as<int>(), (int) ncols, bp.as<int>(), bi.as<int>(), bx.as<double>(),
                       (int *) nullptr);
    SUB_HIP(hipGetLastError());
    if (nb) {
        SUB_HIP(hipMemcpy(Bi, bi.p, (size_t) nb * 4, hipMemcpyDeviceToHost));
        SUB_HIP(hipMemcpy(Bx, bx.p, (size_t) nb * 8, hipMemcpyDeviceToHost));
    }
    return CS3_OK;
}

// label[i] = start (= smallest) node of the island that find_islands puts node i in (csc_numba.py:744-808), exact on
// unsymmetric patterns too (see above).
int cs3_find_islands(int64_t n, const int32_t *Ap, const int32_t *Ai, int32_t *label)
{
    if (n < 0 || n > INT_MAX || !Ap || (n > 0 && !label)) { set_error("cs3_find_islands: bad argument"); return CS3_ERR_ARG; }
    if (no_device("cs3_find_islands")) return CS3_ERR_HIP;
    if (n == 0) return CS3_OK;
    const long long nnz = Ap[n];
    DevBuf ap, ai, lab, chg;
    SUB_HIP(ap.alloc((size_t) (n + 1) * 4)); SUB_HIP(ai.alloc((size_t) nnz * 4)); SUB_HIP(lab.alloc((size_t) n * 4)); SUB_HIP(chg.alloc(4));
    SUB_HIP(hipMemcpy(ap.p, Ap, (size_t) (n + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) SUB_HIP(hipMemcpy(ai.p, Ai, (size_t) nnz * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_label_init, dim3(blocks_for(n, 256)), dim3(256), 0, 0, lab.as<int>(), (int) n);
    int unsym = 0;
    SUB_HIP(hipMemset(chg.p, 0, 4));
    hipLaunchKernelGGL(k_pattern_symmetric, dim3(blocks_for(n, 256)), dim3(256), 0, 0, ap.as<int>(), ai.as<int>(), (int) n, chg.as<int>());
    SUB_HIP(hipMemcpy(&unsym, chg.p, 4, hipMemcpyDeviceToHost));
    for (int64_t round = 0; round <= n; ++round) {
        int changed = 0;
        SUB_HIP(hipMemset(chg.p, 0, 4));
        if (!unsym) {               // structurally symmetric: root hooking + full compression, O(log n) rounds in practice
            hipLaunchKernelGGL(k_label_hook, dim3(blocks_for(n, 256)), dim3(256), 0, 0, ap.as<int>(), ai.as<int>(), (int) n, lab.as<int>(), chg.as<int>());
            hipLaunchKernelGGL(k_label_jump, dim3(blocks_for(n, 256)), dim3(256), 0, 0, lab.as<int>(), (int) n);
        } else {                    // directed reachability, the reference's exact semantics
            hipLaunchKernelGGL(k_reach_hook, dim3(blocks_for(n, 256)), dim3(256), 0, 0, ap.as<int>(), ai.as<int>(), (int) n, lab.as<int>(), chg.as<int>());
            hipLaunchKernelGGL(k_reach_jump, dim3(blocks_for(n, 256)), dim3(256), 0, 0, lab.as<int>(), (int) n, chg.as<int>());
        }
        SUB_HIP(hipMemcpy(&changed, chg.p, 4, hipMemcpyDeviceToHost));
        if (!changed) break;
    }
    SUB_HIP(hipGetLastError());
    SUB_HIP(hipMemcpy(label, lab.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    return CS3_OK;
}

}  // extern "C"
