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
#include <cstdlib>

#include "cs3_device.hpp"
#include "cs3_devfn.hpp"

namespace cs3 {

hipError_t set_withhold_handover(int on)         // diagnostics: see handover_publish (cs3_devfn.hpp)
{
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_withhold_handover), &on, sizeof(int));
    return e != hipSuccess ? e : set_withhold_handover_forest(on);
}

typedef double double4_t __attribute__((ext_vector_type(4)));

// ----------------------------------------------------- assembly by gather --
// A front's assembly list holds (target, source) pairs sorted by target; runs
// of equal targets never cross a 64-entry boundary, so one wave owns whole
// runs.  Lane l of a wave takes entry l of a 64-entry chunk, loads its source
// (pool offset, or ~index into Ax) and the wave sums each run with a fixed
// shuffle tree: every front entry is written exactly once, by one lane, and
// the summation order never changes from run to run.
// Interleaved block: entry `off` (< il.len) of matrix m is  il_lane_base(il, m)[off * 64].
__device__ __forceinline__ double *il_lane_base(const IlView &il, long long m)
{
    return il.base + (m >> 6) * 64 * il.len + (m & 63);
}

constexpr int ASM_DUMMY_T = 0x3fffffff;
constexpr int ASM_LONG_T = 0x40000000;
constexpr int GATHER_UNROLL = 4;

// Fetch(s) returns the address of source s (an unconditional load keeps loads in flight).
template <class Fetch>
__device__ __forceinline__ double gather_value(int t, int s, int s_next, const int *__restrict__ long_src, Fetch fetch)
{
    const bool is_long = (t & ASM_LONG_T) != 0 && t != ASM_DUMMY_T;
    const bool plain = !is_long && t != ASM_DUMMY_T;
    double v = *fetch(plain ? s : 0);
    if (!plain) v = 0.0;
    if (is_long) {                              // rare: more than 64 sources for one entry
        for (int k = 0; k < s_next; ++k) v += *fetch(long_src[s + k]);
    }
    return v;
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

// Store(t, v) is called once per assembled entry.  Wave `wave` of `nwaves` takes the chunks wave, wave + nwaves, ...,
// U of them per pass: the index loads of a pass go out together, then its value loads (two round trips per pass).
template <int U = GATHER_UNROLL, class Fetch, class Store>
__device__ __forceinline__ void gather_front(long long asm_begin, int nchunks, int wave, int nwaves,
                                             const int *__restrict__ asm_src, const int *__restrict__ asm_tgt,
                                             const int *__restrict__ long_src, Fetch fetch, Store store)
{
    const int lane = threadIdx.x & 63;
    for (int c0 = wave; c0 < nchunks; c0 += nwaves * U) {
        int t[U], s[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = c0 + u * nwaves < nchunks;
            const long long idx = asm_begin + (long long) (valid ? c0 + u * nwaves : c0) * 64 + lane;
            const int tl = asm_tgt[idx], sl = asm_src[idx];
            t[u] = valid ? tl : ASM_DUMMY_T;
            s[u] = valid ? sl : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s_next = __shfl_down(s[u], 1);
            v[u] = gather_value(t[u], s[u], s_next, long_src, fetch);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tt = t[u] & ~ASM_LONG_T;
            const bool is_last = run_totals(tt, v[u]);
            if (is_last && tt != ASM_DUMMY_T) store(tt, v[u]);
        }
    }
}

// ------------------------------------------------ assembly by extend-add --
// The factor kernels assemble a front from (i) its entries of A, scattered into the zeroed front, and (ii) its children's
// contribution blocks, added THROUGH THEIR ROW MAPS: F(rel[i], rel[j]) += C(i, j), children in order, a barrier between
// two children (they may meet in an entry; inside one child every entry has a target of its own).  The blocks are read
// column by column, coalesced, and the only index data is one row map of nbc entries per child -- round 1 and 2 read a
// sorted (target, source) list, 8 bytes of index per 8 bytes of value, and summed runs of equal targets with shuffles.
// The order of the additions is fixed, so results stay bitwise reproducible.
//   put(t, v)        stores an entry of A at the target the analysis computed (image index or pool offset)
//   acc(i, j, v)     adds v to entry (i, j) of the front
//   sync()           barrier of the threads that assemble this front together (nth of them, thread `tid`)
// F += v without reading F back: the hardware's f64 add (ds_add_f64 / global_atomic_add_f64, no return value).  Inside one
// child of an extend-add every entry has one source, and two children are separated by a barrier: the ORDER of the
// additions to an entry is fixed, atomic or not -- what the instruction saves is the load, the wait and the store.
__device__ __forceinline__ void front_add(double *p, double v) { unsafeAtomicAdd(p, v); }

template <int KIND, int EUB = 4, class Put, class Acc, class Sync>
__device__ __forceinline__ void assemble_extend_add(const FrontDesc &d, const AsmLists &al, const double *__restrict__ ax,
                                                    const double *__restrict__ pool, const double *__restrict__ pil, long long il_len,
                                                    int tid, int nth, Put put, Acc acc, Sync sync)
{
    for (int e = tid; e < d.a_count; e += nth) {
        const long long idx = d.a_begin + e;
        put(al.fa_tgt[idx], ax[al.fa_src[idx]]);
    }
    sync();
    const int lane = tid & 63;
    const int4 *tab = (const int4 *) al.ch_tab + d.ch_begin;
    int4 ce_next = (d.ch_count > 0) ? tab[0] : int4{0, 0, 0, 0};
    for (int c = 0; c < d.ch_count; ++c) {
        const int4 ce = ce_next;
        if (c + 1 < d.ch_count) ce_next = tab[c + 1];            // the next child's entry travels while this one is added
        const int nbc = ce.x, cbo = ce.z, cld = ce.w;
        const int *__restrict__ rel = al.rel_idx + ce.y;
        const bool il = cbo < il_len;                            // a block in the matrix-interleaved region (large batches)
        // (four entries per thread and pass: their loads go out together -- one entry at a time every pass waited out a
        //  round trip to the block, and a batch of 512 matrices ran 10 % slower than through the gather lists)
        constexpr int EU = 4;
        if (nbc <= 32) {                                         // two columns of the block per wave and step
            const int ii = lane & 31, slots = nth >> 5;
            const bool mine = ii < nbc;
            const int myrel = rel[mine ? ii : 0];
            for (int j0 = tid >> 5; j0 < nbc; j0 += EU * slots) {
                double v[EU];
                int cj[EU];
                bool on[EU];
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int jj = j0 + u * slots;
                    on[u] = mine && jj < nbc && (KIND == CS3_LU || ii >= jj);
                    const long long off = on[u] ? cbo + ii + (long long) jj * cld : cbo;
                    v[u] = il ? pil[off * 64] : pool[off];
                    cj[u] = rel[jj < nbc ? jj : 0];
                }
#pragma unroll
                for (int u = 0; u < EU; ++u)
                    if (on[u]) acc(myrel, cj[u], v[u]);
            }
        } else {
            // the block's entries dealt to the threads one after the other (rows fastest: coalesced), so that every
            // lane has work whatever the block's order; e -> (row, column) by a float reciprocal and two corrections.
            // Cholesky walks the LOWER TRIANGLE only (packed by columns: column j starts at j (2 n - j + 1) / 2; the column
            // of an entry from a float square root and two corrections -- exact: the discriminant stays below 2^24): a
            // walk over the whole square issued a dummy load for every entry above the diagonal, half of all passes.
            const int total = (KIND == CS3_LU) ? nbc * nbc : (nbc * (nbc + 1)) >> 1;
            const float inv = 1.0f / (float) nbc;
            const float twon1 = (float) (2 * nbc + 1);
            constexpr int EU = EUB;                              // (a workgroup that assembles one big front by itself: 12)
            for (int e0 = tid; e0 < total; e0 += EU * nth) {
                double v[EU];
                int ri[EU], cj[EU];
                bool on[EU];
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int e = e0 + u * nth;
                    int ii, jj;
                    if (KIND == CS3_LU) {
                        jj = (int) ((float) e * inv);
                        jj += ((jj + 1) * nbc <= e) ? 1 : 0;
                        jj -= (jj * nbc > e) ? 1 : 0;
                        ii = e - jj * nbc;
                    } else {
                        const int ec = min(e, total - 1);
                        jj = (int) ((twon1 - sqrtf(twon1 * twon1 - 8.0f * (float) ec)) * 0.5f);
                        jj = min(max(jj, 0), nbc - 1);
                        while ((((jj + 1) * (2 * nbc - jj)) >> 1) <= ec) ++jj;             // (start of column jj + 1)
                        while (((jj * (2 * nbc - jj + 1)) >> 1) > ec) --jj;                 // (start of column jj)
                        ii = jj + (ec - ((jj * (2 * nbc - jj + 1)) >> 1));
                    }
                    on[u] = e < total;
                    const long long off = on[u] ? cbo + ii + (long long) jj * cld : cbo;
                    v[u] = il ? pil[off * 64] : pool[off];
                    ri[u] = rel[on[u] ? ii : 0];
                    cj[u] = rel[on[u] ? jj : 0];
                }
#pragma unroll
                for (int u = 0; u < EU; ++u)
                    if (on[u]) acc(ri[u], cj[u], v[u]);
            }
        }
        sync();
    }
}

// The same gather for NV values per source (NV right-hand sides, contiguous in memory):
// fetch(s) returns the address of value 0 of source s, nlive of the NV are real.
template <int NV, class Fetch, class Store>
__device__ __forceinline__ void gather_front_vec(long long asm_begin, int nchunks, const int *__restrict__ asm_src,
                                                 const int *__restrict__ asm_tgt, const int *__restrict__ long_src,
                                                 int nlive, Fetch fetch, Store store)
{
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < nchunks; ++c) {
        const long long idx = asm_begin + (long long) c * 64 + lane;
        const int t = asm_tgt[idx], s = asm_src[idx];
        const int s_next = __shfl_down(s, 1);
        const bool is_long = (t & ASM_LONG_T) != 0 && t != ASM_DUMMY_T;
        const bool plain = !is_long && t != ASM_DUMMY_T;
        const double *src = fetch(plain ? s : 0);
        double v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const double x = src[q < nlive ? q : 0];
            v[q] = (plain && q < nlive) ? x : 0.0;
        }
        if (is_long) {
            for (int k = 0; k < s_next; ++k) {
                const double *p = fetch(long_src[s + k]);
#pragma unroll
                for (int q = 0; q < NV; ++q) if (q < nlive) v[q] += p[q];
            }
        }
        const int tt = t & ~ASM_LONG_T;
        bool is_last = false;
#pragma unroll
        for (int q = 0; q < NV; ++q) is_last = run_totals(tt, v[q]);
        if (is_last && tt != ASM_DUMMY_T) {
#pragma unroll
            for (int q = 0; q < NV; ++q) store(tt, q, v[q]);
        }
    }
}

// ------------------------------------------------- front resident in LDS --
// The front is assembled in LDS (leading dimension r | 1), then every thread
// takes a fixed block-cyclic set of its entries into registers:
//   thread (tx, ty) of a TX x TY grid owns F(tx + TX a, ty + TY b), a < RI, b < RJ.
// Right-looking elimination, pivot k:
//   1. the owners of row k publish it (urow), the owner of (k,k) is among them
//   2. the owners of column k divide by the pivot and publish the column (lcol)
//   3. everyone updates the entries it owns: F(i,j) -= lcol(i) urow(j), i, j > k
// so per pivot only r + r doubles cross the LDS instead of the whole trailing
// block being read and written there.  Results go back through the LDS image
// so that the stores to the panels are row-contiguous.
template <int KIND, int THREADS, int TX, int RI, int RJ>
__device__ __forceinline__ void
front_lds_body(const FrontDesc &d, int first, double *F,
               AsmLists al,
               const double *__restrict__ ax_all, double *__restrict__ pool_all,
               long long nnz_a, long long pool_stride, IlView il, double inv_tol, int *status, long long *tbuf, long long t_start)
{
    constexpr int TY = THREADS / TX;
    const double *pil = il_lane_base(il, blockIdx.y);
#define CS3_STAMP(p) do { if (tbuf && threadIdx.x == 0) tbuf[(long long) (first + blockIdx.x) * 8 + (p)] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    const double *ax = ax_all + (long long) blockIdx.y * nnz_a;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    const int r = d.r, w = d.w, nb = r - w;
    const int ld = r | 1;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    double *urow = F + r * ld + 1;             // [r] pivot row, then [r] scaled pivot column; entry r * ld is the gather's dummy slot

    // ---- assemble: F = sum of sources (A entries, children's contribution blocks)
    CS3_STAMP(0);
    for (int i = tid; i < r * ld; i += THREADS) F[i] = 0.0;
    __syncthreads();
    CS3_STAMP(1);
    assemble_extend_add<KIND>(d, al, ax, pool, pil, il.len, tid, THREADS,
                              [&](int t, double v) { F[t] = v; },
                              [&](int i, int j, double v) { front_add(&F[i + j * ld], v); },
                              [&]() { __syncthreads(); });
    CS3_STAMP(2);

    // ---- my entries into registers
    double R[RI][RJ];
#pragma unroll
    for (int b = 0; b < RJ; ++b)
#pragma unroll
        for (int a = 0; a < RI; ++a) {
            const int i = tx + TX * a, j = ty + TY * b;
            R[a][b] = (i < r && j < r) ? F[i + j * ld] : 0.0;
        }

    // ---- eliminate the w pivots (two barriers per pivot: row, then scaled column)
    double *lcol = urow + r;
    for (int k = 0; k < w; ++k) {
        const int ak = k / TX, txk = k % TX, bk = k / TY, tyk = k % TY;
        if (tx == txk) {                           // 1. row k (incl. the pivot) -> urow
#pragma unroll
            for (int a = 0; a < RI; ++a)
                if (a == ak) {
#pragma unroll
                    for (int b = 0; b < RJ; ++b) {
                        const int j = ty + TY * b;
                        if (j >= k && j < r) urow[j] = R[a][b];
                    }
                }
        }
        __syncthreads();
        const double piv = urow[k];
        double dg, rdg;                            // one reciprocal per pivot: multipliers are x * (1 / pivot)
        pivot_scale<KIND>(piv, dg, rdg);
        if (ty == tyk) {                           // 2. column k / pivot -> lcol
#pragma unroll
            for (int b = 0; b < RJ; ++b)
                if (b == bk) {
#pragma unroll
                    for (int a = 0; a < RI; ++a) {
                        const int i = tx + TX * a;
                        if (i > k && i < r) {
                            R[a][b] *= rdg;
                            lcol[i] = R[a][b];
                            if (KIND == CS3_LU && !(fabs(R[a][b]) <= inv_tol)) flag_column(status, d.c0 + k);
                        }
                        if (i == k) {
                            if (KIND == CS3_LU) {
                                if (!(fabs(piv) > 0.0) || !(fabs(piv) < 1.0e300)) flag_column(status, d.c0 + k);
                            } else {
                                R[a][b] = dg;
                                if (!(piv > 0.0)) flag_column(status, d.c0 + k);
                            }
                        }
                    }
                }
        }
        __syncthreads();
        double lv[RI], uv[RJ];                     // 3. rank-1 update of what I own
#pragma unroll
        for (int a = 0; a < RI; ++a) { const int i = tx + TX * a; lv[a] = (i > k && i < r) ? lcol[i] : 0.0; }
#pragma unroll
        for (int b = 0; b < RJ; ++b) {
            const int j = ty + TY * b;
            uv[b] = (j > k && j < r) ? ((KIND == CS3_LU) ? urow[j] : lcol[j]) : 0.0;
        }
#pragma unroll
        for (int a = 0; a < RI; ++a)
#pragma unroll
            for (int b = 0; b < RJ; ++b) R[a][b] -= lv[a] * uv[b];
    }

    CS3_STAMP(3);
    CS3_STAMP(4);
    // ---- store straight from registers: L panel, U panel, contribution block
    double *L = pool + d.lpan;
    double *U = pool + d.upan;
    double *cb = pool + d.cb;
    const bool has_parent = d.parent >= 0;
#pragma unroll
    for (int b = 0; b < RJ; ++b)
#pragma unroll
        for (int a = 0; a < RI; ++a) {
            const int i = tx + TX * a, j = ty + TY * b;
            if (i >= r || j >= r) continue;
            const double v = R[a][b];
            if (j < w) {
                if (KIND == CS3_LU || i >= j) L[i + (long long) j * r] = v;
            } else if (i < w) {
                if (KIND == CS3_LU) U[(long long) (j - w) * d.u_sj + (long long) i * d.u_sk] = v;
            } else if (has_parent) {
                if (KIND == CS3_LU || i >= j) cb[(i - w) + (long long) (j - w) * nb] = v;
            }
        }
    CS3_STAMP(5);
#undef CS3_STAMP
}


constexpr int PAIR_NC = 16;                   // columns per wave of a two-wave elimination (half a 32-pivot block)

// The elimination of eliminate_block (below) on a SLICE of NC columns of the stacked block: d[c] = column c0 + c, whose pivot sits in lane
// c0 + c (lanes 0..31 are the block's rows, lanes 32..63 the stacked rows).  publish(c, l) receives the multipliers of
// every pivot as soon as they exist, for the wave that holds the columns to the right (eliminate_pair below).
// npiv (a front instead of a padded block): only pivots < npiv are eliminated -- the other steps run with zero multipliers
// rather than behind a branch (a wave-uniform branch per pivot and column group cost the lone wave 200 cycles per pivot).
// SKIP: steps past npiv are skipped behind one wave-uniform branch each (for a slice nobody waits for: the trailing
// steps of a front's second half are mostly such steps).
// (Cholesky: a column update runs in EVERY lane below the pivot row, also where the entry lies above the diagonal of the
//  block -- those entries are never read, stored or checked, and leaving them alone cost a select or a lane mask per
//  column update, a quarter of the instructions of a step.)
template <int KIND, int NC, bool SKIP = false, class Publish>
__device__ __forceinline__ void eliminate_slice(double (&d)[NC], int c0, bool keep_unscaled, Publish publish, int npiv = 1 << 30)
{
    constexpr int EB = 8;
    const int lane = threadIdx.x & 63;
    const bool stacked = lane >= 32;
    double piv = bcast_lane(d[0], c0);
    double dg, rp;
    pivot_scale<KIND>(piv, dg, rp);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int pl = c0 + k;                                  // the pivot's lane
        if (SKIP && pl >= npiv) continue;
        const bool below = lane > pl && pl < npiv;
        const double l = below ? d[k] * rp : 0.0;
        if (below && !(keep_unscaled && stacked)) d[k] = l;
        if (KIND == CS3_CHOLESKY && lane == pl && pl < npiv) d[k] = (piv > 0.0) ? dg : -1.0;
        if (k + 1 < NC) {
            if (KIND == CS3_LU) d[k + 1] -= l * bcast_lane(d[k + 1], pl);
            else { const double lj = bcast_lane(d[k], pl + 1); d[k + 1] -= l * lj; }
            piv = bcast_lane(d[k + 1], pl + 1);
            pivot_scale<KIND>(piv, dg, rp);
        }
        publish(k, l);
#pragma unroll
        for (int j0 = k + 2; j0 < NC; j0 += EB) {
            double bc[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int j = j0 + u;
                if (j < NC) bc[u] = (KIND == CS3_LU) ? bcast_lane(d[j], pl) : bcast_lane(d[k], c0 + j);
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int j = j0 + u;
                if (j < NC) {
                    if (KIND == CS3_LU) d[j] -= l * bc[u];
                    else d[j] -= l * bc[u];
                }
            }
        }
    }
}

// A stacked 32-pivot elimination by TWO waves: a lone wave spends 425 cycles per pivot on it, nearly all of them issuing
// two v_readlane and one FMA per column update.  Wave `part` 0 holds columns 0..15 of the 64 rows, wave 1 columns
// 16..31.  Wave 0 eliminates its 16 pivots and hands the multipliers of each to wave 1 through LDS (lm[k][lane], then
// *ready = k + 1: LDS operations of a wave complete in order); wave 1 applies them to its columns one pivot behind,
// then eliminates pivots 16..31 alone.  Wave 0 never waits for wave 1, so the wait below cannot deadlock; it is bounded
// anyway, and a wave that gives up raises status[3]: cs3_factor_status reports the step as failed (handover_wait).
template <int KIND>
__device__ __forceinline__ void eliminate_pair(double (&d)[PAIR_NC], int part, bool keep_unscaled, double *lm_generic, int *ready_generic,
                                               int *status, int npiv = 1 << 30)
{
    constexpr int NC = PAIR_NC;
    const int lane = threadIdx.x & 63;
    // (explicitly LDS: through generic pointers these become flat accesses, and every hand-over waits out a flat store)
    // volatile on both sides instead of fences: the compiler keeps volatile accesses in program order, the LDS performs a
    // wave's operations in order -- a release fence after every hand-over (s_waitcnt lgkmcnt(0)) sat on the pivot chain
    auto lm = (lds_vdouble_ptr) lm_generic;
    auto ready = (lds_int_ptr) ready_generic;
    const bool withhold = g_withhold_handover != 0;
    if (part == 0) {
        eliminate_slice<KIND, NC>(d, 0, keep_unscaled, [&](int k, double l) {
            lm[k * 64 + lane] = l;
            if (lane == 0) handover_publish(ready, k + 1, withhold);
        }, npiv);
        return;
    }
    bool alive = true;                                          // (a consumer that gave up once does not wait again)
#pragma unroll
    for (int k = 0; k < NC; ++k) {                              // (all NC hand-overs happen, with zeros past npiv)
        if (alive) alive = handover_wait(ready, k, status, withhold);
        const double l = lm[k * 64 + lane];                     // zero on and above the pivot row
#pragma unroll
        for (int j0 = 0; j0 < NC; j0 += 8) {
            double bc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bc[u] = (KIND == CS3_LU) ? bcast_lane(d[j0 + u], k) : bcast_lane(l, NC + j0 + u);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == CS3_LU) d[j0 + u] -= l * bc[u];
                else d[j0 + u] -= l * bc[u];
            }
        }
    }
    eliminate_slice<KIND, NC, true>(d, NC, keep_unscaled, [](int, double) {}, npiv);
}

// ---------------------------------------------- front owned by ONE wave ----
// Fronts of order r <= NC <= 64: lane i keeps ROW i of the front in NC registers, so a pivot
// needs no barrier and no LDS: the pivot row is read lane-to-scalar (v_readlane) and each lane
// updates its own row.  Register indices must be static, so pivots are taken 8 at a time from
// registers 0..7 (unrolled); the 8 finished columns are stored and the row is shifted down by 8,
// which keeps "register j = column kb + j".  LDS is only the target of the assembly gather.
template <int KIND, int NC>
__device__ __forceinline__ void
front_wave_body(const FrontDesc &d, int first, double *F,
                AsmLists al,
                const double *__restrict__ ax_all, double *__restrict__ pool_all,
                long long nnz_a, long long pool_stride, IlView il, double inv_tol, int *status, long long *tbuf, long long t_start)
{
#define CS3_STAMP(p) do { if (tbuf && threadIdx.x == 0) tbuf[(long long) (first + blockIdx.x) * 8 + (p)] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    double *pil = il_lane_base(il, blockIdx.y);
    const double *ax = ax_all + (long long) blockIdx.y * nnz_a;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    const int r = d.r, w = d.w, nb = r - w;
    const int ld = r | 1;
    // The image.  LU: column-major with the odd leading dimension ld.  Cholesky: only the lower triangle exists, PACKED by
    // columns (entry (i, j), i >= j, at i + j (2 r - 1 - j) / 2): half the LDS, so twice the workgroups per CU in a batch
    // (what is read "above the diagonal" below is some other entry of the image -- finite, and it only reaches results
    // nobody keeps).
    const int img = (KIND == CS3_LU) ? r * ld : (r * (r + 1)) >> 1;
    auto at = [&](int i, int j) -> int { return (KIND == CS3_LU) ? i + j * ld : i + ((j * (2 * r - 1 - j)) >> 1); };
    // a pool offset below il.len lives in the matrix-interleaved region (the panels of a front whose sweeps run lane =
    // matrix): entry `off` of this matrix is pil[off * 64]
    auto home = [&](int off) -> double * { return (off < il.len) ? pil + (long long) off * 64 : pool + off; };
    // four waves assemble the front (the gather is latency-bound: more loads in flight), one eliminates it
    CS3_STAMP(0);
    for (int i = threadIdx.x; i < img; i += blockDim.x) F[i] = 0.0;
    __syncthreads();
    CS3_STAMP(1);
    const float inv_ld = 1.0f / (float) ld;
    assemble_extend_add<KIND>(d, al, ax, pool, pil, il.len, (int) threadIdx.x, (int) blockDim.x,
                              [&](int t, double v) {
                                  if (KIND == CS3_LU) { F[t] = v; return; }
                                  int j = (int) ((float) t * inv_ld);       // the targets of A's entries are i + j ld: unpack (t < 2^15)
                                  j += ((j + 1) * ld <= t) ? 1 : 0;
                                  j -= (j * ld > t) ? 1 : 0;
                                  F[at(t - j * ld, j)] = v;
                              },
                              [&](int i, int j, double v) { front_add(&F[at(i, j)], v); },
                              [&]() { __syncthreads(); });
    CS3_STAMP(2);
    const int lane = threadIdx.x & 63;
    bool bad = false;
    int bad_col = 0;
    const bool has_parent = d.parent >= 0;
    const bool live = lane < r;
    {
        // PANEL + SCHUR COMPLEMENT (at most 32 pivots).  Wave 0, lane = row, takes the w pivot columns into registers
        // and eliminates inside them (LU: also the pivot rows, transposed: lane = column); the factored panels go back
        // into the image, and the trailing matrix is formed from them by MFMA (schur_tiles) by all the waves there
        // are, straight into the parent's contribution block.  Until round 3 the one wave carried all r columns
        // through every pivot step: two lane reads and an FMA per column and pivot, 490 cycles per 1024 FMAs
        // against the matrix pipe's 64.
        static_assert(NC == 32, "sub_eliminate holds 32 register columns");
        const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6), nwv = blockDim.x >> 6;
        double row[NC], ut[NC], unused = 0.0;
        bool suspect = false;
        // LU with waves to spare: the pivot rows are wave 1's, on its own (rows_eliminate_lu), while wave 0 factors
        // the pivot columns -- in one wave the two halves queue behind each other, 2 (w - k) column updates per pivot
        const bool split = (KIND == CS3_LU) && nwv > 1;
        const int uwave = split ? 1 : 0;
        const int ucol = split ? lane : w + lane;                       // (wave `uwave`) my column of the pivot rows
        const bool is_ucol = (KIND == CS3_LU) && ucol >= w && ucol < r;
        // no masks on the reads: lanes / registers past the front copy its last row / column -- what they compute
        // goes nowhere (a select on a scalar turns every one of these reads into a branch with a wait of its own)
        if (KIND == CS3_LU && wave == uwave) {
            const int cj = min(ucol, r - 1);
#pragma unroll
            for (int j0 = 0; j0 < NC; j0 += 8)
                if (j0 < w) {
#pragma unroll
                    for (int j = j0; j < j0 + 8; ++j) ut[j] = F[at(min(j, w - 1), cj)];
                }
        }
        if (wave == 0) {
            const int li = min(lane, r - 1);
#pragma unroll
            for (int j0 = 0; j0 < NC; j0 += 8)
                if (j0 < w) {
#pragma unroll
                    for (int j = j0; j < j0 + 8; ++j) row[j] = F[at(li, min(j, w - 1))];
                }
            CS3_STAMP(3);
            if (split) sub_eliminate<KIND, false, false>(row, unused, w, w, inv_tol, suspect, ut);
            else sub_eliminate<KIND, false, KIND == CS3_LU>(row, unused, w, w, inv_tol, suspect, ut);
            // what the tiles read goes back into the image: L21 ...
            if (has_parent && live && lane >= w) {
#pragma unroll
                for (int j0 = 0; j0 < NC; j0 += 8)
                    if (j0 < w) {
#pragma unroll
                        for (int j = j0; j < j0 + 8; ++j)
                            if (j < w) F[at(lane, j)] = row[j];
                    }
            }
        } else if (split && wave == 1) {
            rows_eliminate_lu(ut, w);
        }
        if (KIND == CS3_LU && wave == uwave && has_parent && is_ucol) {         // ... and U12
#pragma unroll
            for (int j0 = 0; j0 < NC; j0 += 8)
                if (j0 < w) {
#pragma unroll
                    for (int j = j0; j < j0 + 8; ++j)
                        if (j < w) F[at(j, ucol)] = ut[j];
                }
        }
        if (blockDim.x > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        CS3_STAMP(4);
        // Pool offsets fit 32 bits (analysis refuses larger pools); a panel below il.len lives in the matrix-interleaved
        // region (entry `off` of this matrix is pil[off * 64]).
        if (has_parent && nb > 0) {
            const int mul = d.cb < il.len ? 64 : 1;
            double *cbg = home((int) d.cb);
            const int sj = nb * mul;
            schur_tiles<KIND>(F, at, r, w, nwv - 1 - wave, nwv, [&](int i, int c, double v) {
                if (KIND == CS3_LU || i >= c) cbg[(i - w) * mul + (c - w) * sj] = v;
            });
        }
        if (KIND == CS3_LU && wave == uwave && is_ucol) {
            const int mul = d.upan < il.len ? 64 : 1;
            double *Up = home((int) d.upan) + (ucol - w) * d.u_sj * mul;
            const int sk = d.u_sk * mul;
#pragma unroll
            for (int j0 = 0; j0 < NC; j0 += 8)
                if (j0 < w) {
#pragma unroll
                    for (int j = j0; j < j0 + 8; ++j)
                        if (j < w) Up[j * sk] = ut[j];
                }
        }
        if (wave == 0) {
            // the panels: one lane-dependent region per destination, groups of eight register columns behind one
            // wave-uniform branch each
            if (live) {
                const int mul = d.lpan < il.len ? 64 : 1;
                double *Lp = home((int) d.lpan) + lane * mul;
                const int sj = r * mul;
#pragma unroll
                for (int j0 = 0; j0 < NC; j0 += 8)
                    if (j0 < w) {
#pragma unroll
                        for (int j = j0; j < j0 + 8; ++j)
                            if (j < w) {
                                if (KIND == CS3_LU) Lp[j * sj] = row[j];
                                else if (lane >= j) Lp[j * sj] = row[j];
                            }
                    }
            }
            if (__any(suspect & live)) {                        // rare: find the first rejected column
                asm volatile("; rejected pivot: look for its column" ::: "memory");     // (keeps the search behind the branch)
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    if (j < w) {
                        const double v = row[j];
                        const double av = fabs(v);
                        bool rej;
                        if (KIND == CS3_LU) {
                            const double lim = (lane == j) ? 1.0e300 : inv_tol;
                            rej = (live & (lane >= j) & !(av <= lim)) | ((lane == j) & !(av > 0.0));
                        } else {
                            rej = (lane == j) & !(v > 0.0);
                        }
                        bad_col = (rej & !bad) ? j : bad_col;
                        bad = bad | rej;
                    }
                }
            }
        }
    }
    if (bad) flag_column(status, d.c0 + bad_col);
    CS3_STAMP(5);
#undef CS3_STAMP
}

// The one-wave path as its own launch, one wave per front: used when a batch of matrices keeps
// the chip busy anyway, so the three helper waves of k_front_mix would only cost occupancy.
template <int KIND>
__global__ void __launch_bounds__(64)
k_front_wave(const FrontDesc *__restrict__ fdesc, int first,
             AsmLists al,
             const double *__restrict__ ax_all, double *__restrict__ pool_all,
             long long nnz_a, long long pool_stride, IlView il, double inv_tol, int *status, long long *tbuf)
{
    extern __shared__ __attribute__((aligned(16))) double F[];
    const long long t_start = tbuf ? (long long) __builtin_amdgcn_s_memtime() : 0;
    const FrontDesc d = fdesc[first + blockIdx.x];
    front_wave_body<KIND, 32>(d, first, F, al, ax_all, pool_all, nnz_a, pool_stride, il,
                              inv_tol, status, tbuf, t_start);
}

// Every front of order <= 64 in one launch: one wave eliminates the small ones (r <= 32), the
// 16 x 16 thread grid the others.  One launch instead of two per tree level.
template <int KIND>
__global__ void __launch_bounds__(256)
k_front_mix(const FrontDesc *__restrict__ fdesc, int first,
            AsmLists al,
            const double *__restrict__ ax_all, double *__restrict__ pool_all,
            long long nnz_a, long long pool_stride, IlView il, double inv_tol, int *status, long long *tbuf)
{
    extern __shared__ __attribute__((aligned(16))) double F[];
    const long long t_start = tbuf ? (long long) __builtin_amdgcn_s_memtime() : 0;
    const FrontDesc d = fdesc[first + blockIdx.x];
    if (d.w <= 32)                      // (four waves assemble, one factors the panel, four form the Schur complement)
        front_wave_body<KIND, 32>(d, first, F, al, ax_all, pool_all, nnz_a, pool_stride, il,
                                  inv_tol, status, tbuf, t_start);
    else
        front_lds_body<KIND, 256, 16, 4, 4>(d, first, F, al, ax_all, pool_all, nnz_a,
                                            pool_stride, il, inv_tol, status, tbuf, t_start);
}

// ------------------------------------------- front too large for the LDS --
// The front is one dense r x r column-major buffer in the pool (zeroed at the
// start of the factorisation).  Per front: one gather launch, then per block
// of BIG_NB pivots a panel launch and a trailing-update launch -- a blocked
// right-looking LU / Cholesky without pivoting, many workgroups per launch.
constexpr int BIG_NB = 32;
static_assert(BIG_NB == 2 * PAIR_NC, "eliminate_pair splits a block of BIG_NB pivots in two");

__global__ void __launch_bounds__(256)
k_big_gather(const FrontDesc *__restrict__ fdesc, int first, int kind,
             AsmLists al,
             const double *__restrict__ ax_all, double *__restrict__ pool_all,
             long long nnz_a, long long pool_stride, IlView il)
{
    // Several workgroups assemble one front: workgroup b OWNS a block of the front's columns, so that no two of them ever
    // touch the same entry and the children can still be added in order with block barriers only.  A child's columns that
    // land in my block are a contiguous range of its row map (the map ascends).
    const FrontDesc d = fdesc[first + blockIdx.z];
    const double *pil = il_lane_base(il, blockIdx.y);
    const double *ax = ax_all + (long long) blockIdx.y * nnz_a;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    const int r = d.r;
    const int cw = (r + (int) gridDim.x - 1) / (int) gridDim.x, c_lo = (int) blockIdx.x * cw, c_hi = min(r, c_lo + cw);
    if (c_lo >= r) return;
    double *F = pool + d.lpan;                       // dense r x r, leading dimension r (zeroed by the prologue)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float inv_r = 1.0f / (float) r;
    for (int e = tid; e < d.a_count; e += 256) {
        const long long idx = d.a_begin + e;
        const int t = al.fa_tgt[idx], rel_t = t - (int) d.lpan;
        int col = (int) ((float) rel_t * inv_r);     // t = lpan + row + col r: unpack (exact after the two corrections)
        col += ((long long) (col + 1) * r <= rel_t) ? 1 : 0;
        col -= ((long long) col * r > rel_t) ? 1 : 0;
        if (col >= c_lo && col < c_hi) pool[t] = ax[al.fa_src[idx]];
    }
    __syncthreads();
    const int4 *tab = (const int4 *) al.ch_tab + d.ch_begin;
    int4 ce_next = (d.ch_count > 0) ? tab[0] : int4{0, 0, 0, 0};
    for (int c = 0; c < d.ch_count; ++c) {
        const int4 ce = ce_next;
        if (c + 1 < d.ch_count) ce_next = tab[c + 1];           // the next child's entry travels while this one is added
        const int nbc = ce.x, cbo = ce.z, cld = ce.w;
        const int *__restrict__ rel = al.rel_idx + ce.y;
        const bool inter = cbo < il.len;
        // first child column at or beyond c_lo / c_hi: every wave COUNTS the map entries below the bound (one round of
        // loads, a ballot per 64 entries) -- a binary search is log2(nbc) dependent round trips, twice per child
        int j0 = 0, j1 = 0;
        for (int base = 0; base < nbc; base += 256) {
            int rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = base + 64 * u + lane;
                rv[u] = rel[e < nbc ? e : 0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = base + 64 * u + lane < nbc;
                j0 += __popcll(__builtin_amdgcn_ballot_w64(in && rv[u] < c_lo));
                j1 += __popcll(__builtin_amdgcn_ballot_w64(in && rv[u] < c_hi));
            }
        }
        for (int jj = j0 + wv; jj < j1; jj += 4) {
            const long long cj = (long long) rel[jj] * r;
            for (int ii = lane; ii < nbc; ii += 64) {
                if (kind == CS3_LU || ii >= jj) {
                    const long long off = cbo + ii + (long long) jj * cld;
                    front_add(&F[rel[ii] + cj], inter ? pil[off * 64] : pool[off]);
                }
            }
        }
        __syncthreads();
    }
}

// Unblocked LU / Cholesky of a 32 x 32 block by ONE wave without barriers, with a panel solve for
// free: lanes 0..31 keep the rows of the block in 32 registers each, lanes 32..63 keep 32 rows of a
// panel tile STACKED under it.  For pivot k the pivot row is read lane-to-scalar (v_readlane) and
// every lane below it (all stacked lanes included) takes its multiplier and updates its row: when the
// loop ends the stacked lanes hold  T U^-1  (LU; Cholesky: T L^-T), i.e. the solved panel rows --
// the triangular solve that used to follow the factorisation costs nothing.
//   * the block is identity-padded past its order, so the 32 steps run without a branch and the
//     next pivot's reciprocal (v_rcp + two Newton steps) is issued right after the first column
//     update of the current step, behind which its latency hides;
//   * keep_unscaled (block-ROW tiles of LU): the block lanes hold D' and the stacked lanes columns of
//     the tile; the multiplier t_k / u_kk drives the updates but the entry that is kept is t_k
//     itself, which is U(k, column) for the unit-lower solve  L_D u = t.
template <int KIND, int NBK>
__device__ __forceinline__ void eliminate_block(double (&d)[NBK], bool keep_unscaled, int nsteps = NBK)
{
    constexpr int EB = 8;                       // columns whose broadcasts are issued together
    const int lane = threadIdx.x & 63;
    const bool stacked = lane >= 32;
    double piv = bcast_lane(d[0], 0);
    double dg, rp;
    pivot_scale<KIND>(piv, dg, rp);
#pragma unroll
    for (int k = 0; k < NBK; ++k) {
        if (k < nsteps) {                       // (wave-uniform: the identity-padded steps of a short block do nothing)
        const bool below = lane > k;
        const double l = below ? d[k] * rp : 0.0;
        if (below && !(keep_unscaled && stacked)) d[k] = l;
        if (KIND == CS3_CHOLESKY && lane == k) d[k] = (piv > 0.0) ? dg : -1.0;
        if (k + 1 < NBK) {
            if (KIND == CS3_LU) d[k + 1] -= l * bcast_lane(d[k + 1], k);
            else { const double lj = bcast_lane(d[k], k + 1); d[k + 1] -= l * lj; }
            piv = bcast_lane(d[k + 1], k + 1);
            pivot_scale<KIND>(piv, dg, rp);
        }
        // the broadcasts of EB columns go out before their FMAs: a lane-to-scalar read needs wait states before the
        // vector instruction that consumes it, which the next broadcasts fill
#pragma unroll
        for (int j0 = k + 2; j0 < NBK; j0 += EB) {
            double bc[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int j = j0 + u;
                if (j < NBK) bc[u] = (KIND == CS3_LU) ? bcast_lane(d[j], k) : bcast_lane(d[k], j);
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int j = j0 + u;
                if (j < NBK) {
                    if (KIND == CS3_LU) d[j] -= l * bc[u];
                    else d[j] -= l * bc[u];                                         // L(j, k): lane j, register k
                }
            }
        }
        }
    }
}

template <int KIND>
__device__ __forceinline__ void eliminate32(double (&d)[BIG_NB], bool keep_unscaled) { eliminate_block<KIND, BIG_NB>(d, keep_unscaled); }


// Fronts of order 65 .. 136: the front image lives in LDS (one workgroup of 8 waves per front) and is
// factorised 32 pivots at a time with the same two tools as the big fronts, without leaving the workgroup:
//   1. eliminate32 -- waves 0..3 stack 32 rows of the block column under the 32 x 32 diagonal block D, waves
//      4..7 stack 32 columns of the block row under D'; the stacked lanes come out as the solved panels;
//   2. the trailing part of the image gets  F22 -= L21 U12  by v_mfma_f64_16x16x4 with both operands read straight
//      from the LDS image (column-major with an odd leading dimension: conflict-free either way).
// NBK = pivots per block: 32, or 16 for a launch whose fronts have at most 16 pivots (the elimination runs all NBK
// identity-padded steps).  The previous kernel for this class kept 5 x 9 register tiles and crossed the LDS and two block barriers for
// every pivot (about 3 k cycles per pivot; this one: about 0.6 k).
template <int KIND, int NBK, int NW = 8>
__global__ void __launch_bounds__(NW * 64)
k_front_block(const FrontDesc *__restrict__ fdesc, int first,
              AsmLists al,
              const double *__restrict__ ax_all, double *__restrict__ pool_all,
              long long nnz_a, long long pool_stride, IlView il, double inv_tol, int *status, long long *tbuf)
{
    extern __shared__ __attribute__((aligned(16))) double F[];
    const double *pil = il_lane_base(il, blockIdx.y);
    const long long t_start = tbuf ? (long long) __builtin_amdgcn_s_memtime() : 0;
#define CS3_STAMP(p) do { if (tbuf && threadIdx.x == 0) tbuf[(long long) (first + blockIdx.x) * 8 + (p)] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    const FrontDesc d = fdesc[first + blockIdx.x];
    const double *ax = ax_all + (long long) blockIdx.y * nnz_a;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    const int r = d.r, w = d.w, nb = r - w;
    const int ld = r | 1;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 31;
    const bool stacked = lane >= 32;
    // The image.  LU: column-major with the odd leading dimension ld.  Cholesky: only the lower triangle exists,
    // PACKED by columns (entry (i, j), i >= j, at i + j (2 r - 1 - j) / 2): half the LDS, so twice the workgroups per CU
    // -- the phases of one front are serial (gather, one-wave eliminations, barriers) and a batch hides them only with
    // other fronts on the same CU.  F[img] (one past the image) stays zero for the masked loads.
    const int img = (KIND == CS3_LU) ? r * ld : (r * (r + 1)) >> 1;
    auto at = [&](int i, int j) -> int { return (KIND == CS3_LU) ? i + j * ld : i + ((j * (2 * r - 1 - j)) >> 1); };

    // ---- assemble: F = sum of sources (A entries, children's contribution blocks)
    CS3_STAMP(0);
    for (int i = tid; i < img + 1; i += NW * 64) F[i] = 0.0;
    __syncthreads();
    CS3_STAMP(1);
    const float inv_ld = 1.0f / (float) ld;
    assemble_extend_add<KIND>(d, al, ax, pool, pil, il.len, tid, NW * 64,
                              [&](int t, double v) {
                                  if (KIND == CS3_LU) { F[t] = v; return; }
                                  int j = (int) ((float) t * inv_ld);       // the targets of A's entries are i + j ld: unpack (t < 2^15)
                                  j += ((j + 1) * ld <= t) ? 1 : 0;
                                  j -= (j * ld > t) ? 1 : 0;
                                  F[at(t - j * ld, j)] = v;
                              },
                              [&](int i, int j, double v) { front_add(&F[at(i, j)], v); },
                              [&]() { __syncthreads(); });
    CS3_STAMP(2);

    bool bad = false;
    int bad_col = 0;
    long long t_ph[3] = {0, 0, 0}, t_last = tbuf ? (long long) __builtin_amdgcn_s_memtime() : 0;
#define CS3_PHASE(p) do { if (tbuf) { const long long t_now = (long long) __builtin_amdgcn_s_memtime(); t_ph[p] += t_now - t_last; t_last = t_now; } } while (0)
    // Blocks of equal width (17 pivots: 9 + 8, not 16 + 1 -- a block step costs its eliminations whatever it holds).  Inside
    // the loop only the REMAINING PIVOT columns and rows are updated; the contribution block is formed once, after the
    // last block, from all w pivots and goes straight to the parent's buffer (step 3): it used to be read and written in
    // the LDS image once per block step, and copied out at the end.
    const bool has_parent = d.parent >= 0;
    const int nblk = (w + NBK - 1) / NBK, bsz = (w + nblk - 1) / nblk;
    for (int kb = 0; kb < w; kb += bsz) {
        const int bw = min(bsz, w - kb), ke = kb + bw, nrem = r - ke;
        // ---- 1. the block and its panels: one stacked elimination per wave (at most 128 rows / columns lie
        // beyond a block: the analysis sends the rare front that would need a fifth group to the big-front class)
        const bool row_wave = wv >= 4;                      // D' with columns of the block row stacked (LU only)
        const int grp = wv & 3, s0 = ke + 32 * grp, si = s0 + li;      // my stacked row (column for a row wave)
        const bool owner = wv == 0;                         // keeps the factored block
        const bool active = !(KIND == CS3_CHOLESKY && row_wave) && (owner || s0 < r);
        double e[NBK];
        if (active) {
            // entry j of my row: one LDS read.  Block lanes: D (D' for a row wave), identity past bw; stacked lanes: my
            // row of the block column (my column of the block row for a row wave).  Cholesky reads the lower triangle.
            const int line = stacked ? si : kb + li;
            const bool mine = stacked ? si < r : li < bw;
#pragma unroll
            for (int j = 0; j < NBK; ++j) {
                const bool in = mine && j < bw && (KIND == CS3_LU || stacked || li >= j);
                const int off = (KIND == CS3_LU) ? (row_wave ? kb + j + line * ld : line + (kb + j) * ld) : at(line, kb + j);
                const double v = F[in ? off : img];
                e[j] = in ? v : ((!stacked && li == j) ? 1.0 : 0.0);
            }
        }
        __syncthreads();                                    // everybody has read D before wave 0 writes its factors back
        CS3_PHASE(0);
        if (active) {
            eliminate_block<KIND, NBK>(e, row_wave, bw);
            // pivots and multipliers, then home into the image
#pragma unroll
            for (int j = 0; j < NBK; ++j) {
                if (j < bw) {
                    const double v = e[j], av = fabs(v);
                    if (!stacked) {
                        if (owner && li < bw) {
                            bool rej;
                            if (KIND == CS3_LU) {
                                const double lim = (li == j) ? 1.0e300 : inv_tol;
                                rej = ((li >= j) & !(av <= lim)) | ((li == j) & !(av > 0.0));
                            } else {
                                rej = (li == j) & !(v > 0.0);
                            }
                            if (rej && !bad) { bad = true; bad_col = kb + j; }
                            if (KIND == CS3_LU || li >= j) F[at(kb + li, kb + j)] = v;
                        }
                    } else if (si < r) {
                        if (!row_wave) {
                            if (KIND == CS3_LU && !(av <= inv_tol) && !bad) { bad = true; bad_col = kb + j; }
                            F[at(si, kb + j)] = v;
                        } else {
                            F[at(kb + j, si)] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
        CS3_PHASE(1);
        // ---- 2. trailing update by MFMA, operands from the image: 16 x 16 tiles of F22 dealt to the 8 waves
        if (nrem > 0 && ke < w) {
            const int nt = (nrem + 15) / 16, mi = lane & 15, mq = lane >> 4;
            int turn = 0;                                       // (tiles dealt round-robin; no division in the walk)
            for (int tj = 0; tj < nt; ++tj)
              for (int ti = (KIND == CS3_CHOLESKY) ? tj : 0; ti < nt; ++ti) {     // tile rows ke + 16 ti.., columns ke + 16 tj..
                if (ke + 16 * ti >= w && ke + 16 * tj >= w) continue;      // inside the contribution block: step 3
                const bool my_turn = turn == wv;
                turn = (turn + 1 == NW) ? 0 : turn + 1;
                if (!my_turn) continue;
                const int i = ke + 16 * ti + mi;                // my row (B operand / output lanes % 16)
                double4_t acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = ke + 16 * tj + mq + 4 * v;
                    const bool in = i < r && c < r && (KIND == CS3_LU || i >= c);
                    acc[v] = F[in ? at(i, c) : img];
                }
                const int ca = ke + 16 * tj + mi;               // A operand: U(kb + k, column ca) (Cholesky: L(ca, kb + k))
#pragma unroll
                for (int k0 = 0; k0 < NBK; k0 += 4) {
                    if (k0 < bw) {
                        const int k = kb + k0 + mq;
                        const bool kin = k0 + mq < bw;
                        const double au = F[(kin && ca < r) ? ((KIND == CS3_LU) ? k + ca * ld : at(ca, k)) : img];
                        const double bl = -F[(kin && i < r) ? at(i, k) : img];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(au, bl, acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = ke + 16 * tj + mq + 4 * v;
                    if (i < r && c < r && (i < w || c < w) && (KIND == CS3_LU || i >= c)) F[at(i, c)] = acc[v];
                }
            }
        }
        __syncthreads();
        CS3_PHASE(2);
    }
    // ---- 3. the contribution block:  C(i, j) = F(i, j) - sum_{k < w} L(i, k) U(k, j)  for i, j >= w, by MFMA from the
    // image (whose pivot columns and rows are final), in pivot order like the block-wise updates it replaces
    if (has_parent && nb > 0) {
        double *cb = pool + d.cb;
        const int nt = (nb + 15) / 16, mi = lane & 15, mq = lane >> 4;
        int turn = 0;
        for (int tj = 0; tj < nt; ++tj)
          for (int ti = (KIND == CS3_CHOLESKY) ? tj : 0; ti < nt; ++ti) {
            const bool my_turn = turn == wv;
            turn = (turn + 1 == NW) ? 0 : turn + 1;
            if (!my_turn) continue;
            const int i = w + 16 * ti + mi, ca = w + 16 * tj + mi;
            double4_t acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = w + 16 * tj + mq + 4 * v;
                const bool in = i < r && c < r && (KIND == CS3_LU || i >= c);
                acc[v] = F[in ? at(i, c) : img];
            }
            for (int k0 = 0; k0 < w; k0 += 16) {                // sixteen pivots per pass: their operands first, then four MFMAs
                double au[4], bl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 4 * u + mq;
                    const bool kin = k < w;
                    au[u] = F[(kin && ca < r) ? ((KIND == CS3_LU) ? k + ca * ld : at(ca, k)) : img];
                    bl[u] = -F[(kin && i < r) ? at(i, k) : img];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(au[u], bl[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = w + 16 * tj + mq + 4 * v;
                if (i < r && c < r && (KIND == CS3_LU || i >= c)) cb[(i - w) + (c - w) * nb] = acc[v];
            }
        }
    }
#undef CS3_PHASE
    if (tbuf && threadIdx.x == 0) {                        // (diagnostics: block loads / eliminations + write-back / updates)
        tbuf[(long long) (first + blockIdx.x) * 8 + 6] = t_ph[1];
        tbuf[(long long) (first + blockIdx.x) * 8 + 7] = t_ph[2];
    }
    CS3_STAMP(3);
    CS3_STAMP(4);
    if (bad) flag_column(status, d.c0 + bad_col);
    // ---- store: L panel, U panel, contribution block
    double *L = pool + d.lpan;
    double *U = pool + d.upan;
    for (int e = tid; e < r * w; e += NW * 64) {                    // L panel, column-major r x w
        const int i = e % r, j = e / r;
        if (KIND == CS3_LU || i >= j) L[e] = F[at(i, j)];
    }
    if (KIND == CS3_LU)
        for (int e = tid; e < w * nb; e += NW * 64) {               // U panel: pivot rows contiguous (u_sk = 1, u_sj = w)
            const int i = e % w, j = e / w;
            U[(long long) j * d.u_sj + (long long) i * d.u_sk] = F[i + (w + j) * ld];
        }
    CS3_STAMP(5);
#undef CS3_STAMP
}

// ------------------------------------ big front, ONE workgroup, ONE launch (batches) --
// A batch of matrices keeps every CU busy on its own: the front of one matrix then needs no help from other
// workgroups, and the launch per block step that k_big_step pays for a single matrix (each step waits for the
// whole grid of the one before) only costs time.  Here one workgroup of 8 waves factorises one front of one
// matrix from start to end: zero + gather into the dense r x r buffer in HBM (it stays in this CU's L1 / the
// XCD's L2), then per block of 32 pivots
//   1. the 32 x 32 diagonal block goes to LDS; every wave eliminates it with 32 rows of the block column (LU:
//      or 32 columns of the block row) stacked underneath (eliminate_block: the stacked lanes come out as the
//      solved panel rows), writes them home and into the LDS panels Lp / Up;
//   2. the trailing matrix gets  F22 -= L21 U12  in 16 x 16 tiles by v_mfma_f64_16x16x4, accumulators loaded
//      from and stored to the buffer, both operands read from the LDS panels.
// Two block barriers per 32 pivots, no grid-wide dependency.  Used when the batch alone fills the chip and the
// panels fit the LDS (launch_front_group decides); results equal k_big_step's to rounding, not bit for bit
// (the MFMA sums 32 products per step either way, but the diagonal block is updated tile-wise here).
template <int KIND, int NB>
__global__ void __launch_bounds__(512, (NB <= 16) ? 4 : 2)
k_front_wg(const FrontDesc *__restrict__ fdesc, int first,
           AsmLists al,
           const double *__restrict__ ax_all, double *__restrict__ pool_all,
           long long nnz_a, long long pool_stride, IlView il, double inv_tol, int *status, int pld, long long *tbuf)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const double *pil = il_lane_base(il, blockIdx.y);
    // diagnostics (CS3_PROFILE=1), matrix 0 of the batch: slot 0 zeroed, 1 gathered, 2 sum of the panel phases,
    // 3 sum of the update phases, 4 block steps, 5 end -- shader-clock cycles
    const bool prof = tbuf && blockIdx.y == 0 && threadIdx.x == 0;
    const long long t_start = prof ? (long long) __builtin_amdgcn_s_memtime() : 0;
    long long t_panel = 0, t_update = 0, t_mark = 0;
#define CS3_WSTAMP(p) do { if (prof) tbuf[(long long) (first + blockIdx.x) * 8 + (p)] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    const FrontDesc d = fdesc[first + blockIdx.x];
    const double *ax = ax_all + (long long) blockIdx.y * nnz_a;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    const int r = d.r, w = d.w;
    double *F = pool + d.lpan;
    const long long ld = r;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 31;
    const bool stacked = lane >= 32;
    double *Dl = sm;                           // Dl[j * 33 + i] = D(i, j)
    double *Lp = sm + NB * 33;             // Lp[k * pld + i] = L(ke + i, kb + k)
    double *Up = Lp + NB * pld;            // Up[k * pld + j] = U(kb + k, ke + j)   (LU only)

    // ---- assemble: zero the buffer, then F = sum of sources (A entries, children's contribution blocks)
    if (KIND == CS3_LU) {
        for (long long e = tid; e < (long long) r * r; e += 512) F[e] = 0.0;
    } else {
        // Cholesky never reads an entry above the diagonal for a result it keeps (what the eliminations and updates
        // compute there stays there): half the buffer need not be written
        for (int j = wv; j < r; j += 8)
            for (int i = j + lane; i < r; i += 64) F[i + (long long) j * ld] = 0.0;
    }
    __syncthreads();
    CS3_WSTAMP(0);
    assemble_extend_add<KIND, 12>(d, al, ax, pool, pil, il.len, tid, 512,
                              [&](int t, double v) { pool[t] = v; },
                              [&](int i, int j, double v) { front_add(&F[i + (long long) j * ld], v); },
                              [&]() { __syncthreads(); });
    CS3_WSTAMP(1);

    bool bad = false;
    int bad_col = 0;
    if (KIND == CS3_CHOLESKY) {
        // LEFT-LOOKING (round 3).  The right-looking form below reads and writes the whole trailing matrix of the front
        // -- which lives in global memory here -- once per 16 pivots (1.9 MB per matrix of a 226 x 226 root, every block
        // step waiting for those round trips).  Here a block column is touched ONCE: its 16-row tiles take
        //      C(rows, block) -= L(rows, 0:kb) L(block, 0:kb)'
        // on the matrix cores (the block's own rows of L staged in LDS, the tile's rows streamed from the factor: 0.5 MB
        // per matrix in all), land in LDS as the block column, are eliminated there with the panels stacked under the
        // diagonal block, and go home as final columns of L.  Nothing is left to update afterwards.
        double *P = Lp;                             // P[j * pld + ii] = entry (kb + ii, kb + j) of the block column
        double *Ak = Lp + NB * pld;                 // Ak[k * 16 + j] = L(kb + j, k), k < K
        const int mi = lane & 15, mq = lane >> 4;
        const bool has_cb = d.parent >= 0 && r > w;
        // pivot blocks [kb, kb + bw) with K = kb pivots to their left, then -- the contribution block of a front that has
        // one -- the remaining columns 16 at a time with all K = w pivots and no elimination
        for (int kb = 0, ke = 0; kb < (has_cb ? r : w); kb = ke) {
            const bool pivots = kb < w;
            const int bw = pivots ? min(NB, w - kb) : min(NB, r - kb), nrow = r - kb;
            ke = kb + bw;
            const int nrem = r - ke;
            const int K = pivots ? kb : w;
            if (prof) t_mark = (long long) __builtin_amdgcn_s_memtime();
            for (int e = tid; e < K * 16; e += 512) {
                const int k = e >> 4, j = e & 15;
                Ak[e] = load_if(F, (kb + j) + (long long) k * ld, j < bw);
            }
            __syncthreads();
            const int nt = (nrow + 15) >> 4;
            for (int t = wv; t < nt; t += 8) {
                const int i = kb + 16 * t + mi;                 // my row: B operand and the outputs' lane & 15
                const bool irow = i < r;
                double4_t acc;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = load_if(F, i + (long long) (kb + mq + 4 * v) * ld, irow && mq + 4 * v < bw);
                const double *Frow = F + (irow ? i : 0);
                // 64 pivots per pass: the 16 operand loads of a lane go out together (the tile's rows of the factor come
                // from L2 / HBM: with four loads per pass a late block column waited out 13 round trips per tile)
                for (int k0 = 0; k0 < K; k0 += 64) {
                    double bl[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int k = k0 + 4 * u + mq;
                        bl[u] = Frow[(long long) (k < K ? k : 0) * ld];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int k = k0 + 4 * u + mq;
                        const bool kin = k < K;
                        if (k0 + 4 * u < K)                     // (wave-uniform)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kin ? Ak[k * 16 + mi] : 0.0, (kin && irow) ? -bl[u] : 0.0, acc, 0, 0, 0);
                    }
                }
                if (pivots) {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (irow) P[(mq + 4 * v) * pld + 16 * t + mi] = acc[v];
                } else {                                        // a block of the contribution block: final, in place
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (irow && mq + 4 * v < bw) F[i + (long long) (kb + mq + 4 * v) * ld] = acc[v];
                }
            }
            __syncthreads();
            if (prof) { const long long now = (long long) __builtin_amdgcn_s_memtime(); t_update += now - t_mark; t_mark = now; }
            if (!pivots) continue;
            // the block column: every wave eliminates the diagonal block with 32 of the rows below it stacked underneath
            const int ng = max(1, (nrem + 31) / 32);
            for (int t = wv; t < ng; t += 8) {
                const int s0 = ke + 32 * t, si = s0 + li;
                const bool owner = t == 0;
                double e[NB];
                if (!stacked) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const double v = P[j * pld + li];
                        e[j] = (li < bw && j < bw) ? v : ((li == j) ? 1.0 : 0.0);
                    }
                } else {
                    const bool mine = si < r;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const double v = P[j * pld + (mine ? si - kb : 0)];
                        e[j] = (mine && j < bw) ? v : 0.0;
                    }
                }
                eliminate_block<KIND, NB>(e, false, bw);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (j < bw) {
                        const double v = e[j];
                        if (!stacked) {
                            if (owner && li < bw) {
                                if ((li == j) & !(v > 0.0) && !bad) { bad = true; bad_col = kb + j; }
                                if (li >= j) F[(kb + li) + (long long) (kb + j) * ld] = v;
                            }
                        } else if (si < r) {
                            F[si + (long long) (kb + j) * ld] = v;
                        }
                    }
                }
            }
            __syncthreads();                        // the columns are in the factor before the next block reads them
            if (prof) t_panel += (long long) __builtin_amdgcn_s_memtime() - t_mark;
        }
    } else
    for (int kb = 0; kb < w; kb += NB) {
        const int bw = min(NB, w - kb), ke = kb + bw, nrem = r - ke;
        if (prof) t_mark = (long long) __builtin_amdgcn_s_memtime();
        for (int e = tid; e < NB * NB; e += 512) {
            const int i = e % NB, j = e / NB;
            Dl[j * 33 + i] = load_if(F, (kb + i) + (long long) (kb + j) * ld, i < bw && j < bw);
        }
        __syncthreads();
        // ---- 1. panels: task 0 = the owner (keeps D) with the first 32 rows of the block column stacked; tasks
        // 1 .. ng - 1 the other row groups; LU: tasks ng .. 2 ng - 1 the column groups of the block row under D'
        const int ng = max(1, (nrem + 31) / 32);
        const int ntask = (KIND == CS3_LU) ? 2 * ng : ng;
        for (int t = wv; t < ntask; t += 8) {
            const bool row_wave = t >= ng;
            const int grp = row_wave ? t - ng : t, s0 = ke + 32 * grp, si = s0 + li;
            const bool owner = t == 0;
            double e[NB];
            if (!stacked) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const double v = row_wave ? Dl[li * 33 + j] : Dl[j * 33 + li];
                    e[j] = (li < bw && j < bw) ? v : ((li == j) ? 1.0 : 0.0);
                }
            } else {
                const bool mine = si < r;
                const long long base = row_wave ? (long long) kb + (long long) si * ld : (long long) si + (long long) kb * ld;
                const long long stride = row_wave ? 1 : ld;
#pragma unroll
                for (int j = 0; j < NB; ++j) e[j] = load_if(F, base + j * stride, mine && j < bw);
            }
            eliminate_block<KIND, NB>(e, row_wave);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < bw) {
                    const double v = e[j], av = fabs(v);
                    if (!stacked) {
                        if (owner && li < bw) {
                            bool rej;
                            if (KIND == CS3_LU) {
                                const double lim = (li == j) ? 1.0e300 : inv_tol;
                                rej = ((li >= j) & !(av <= lim)) | ((li == j) & !(av > 0.0));
                            } else {
                                rej = (li == j) & !(v > 0.0);
                            }
                            if (rej && !bad) { bad = true; bad_col = kb + j; }
                            F[(kb + li) + (long long) (kb + j) * ld] = v;
                        }
                    } else if (si < r) {
                        if (!row_wave) {
                            if (KIND == CS3_LU && !(av <= inv_tol) && !bad) { bad = true; bad_col = kb + j; }
                            F[si + (long long) (kb + j) * ld] = v;
                            Lp[j * pld + (si - ke)] = v;
                        } else {
                            F[(kb + j) + (long long) si * ld] = v;
                            Up[j * pld + (si - ke)] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (prof) { const long long now = (long long) __builtin_amdgcn_s_memtime(); t_panel += now - t_mark; t_mark = now; }
        // ---- 2. trailing update by MFMA: 16 x 16 tiles of F22 dealt to the 8 waves, operands from the LDS panels.  The
        // tiles that exist (Cholesky: the lower ones) are numbered 0 .. ntile - 1 and a wave takes TU of them per pass: their
        // accumulator loads go out together, so a pass costs one round trip to the buffer instead of TU
        if (nrem > 0) {
            constexpr int TU = 3;
            const int nt = (nrem + 15) / 16, mi = lane & 15, mq = lane >> 4;
            const int ntile = (KIND == CS3_LU) ? nt * nt : nt * (nt + 1) / 2;
            const double *Ua = (KIND == CS3_LU) ? Up : Lp;
            for (int u0 = wv; u0 < ntile; u0 += 8 * TU) {
                int ti[TU], tj[TU];
                bool on[TU];
                double4_t acc[TU];
#pragma unroll
                for (int q = 0; q < TU; ++q) {
                    int u = u0 + 8 * q;
                    on[q] = u < ntile;
                    if (!on[q]) u = 0;
                    if (KIND == CS3_LU) { ti[q] = u % nt; tj[q] = u / nt; }
                    else {                                      // column tj holds the tiles tj .. nt - 1
                        int c = 0;
                        while (u >= nt - c) { u -= nt - c; ++c; }
                        tj[q] = c; ti[q] = c + u;
                    }
                }
#pragma unroll
                for (int q = 0; q < TU; ++q) {
                    const int i = ke + 16 * ti[q] + mi;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int c = ke + 16 * tj[q] + mq + 4 * v;
                        acc[q][v] = load_if(F, i + (long long) c * ld, on[q] && i < r && c < r);
                    }
                }
#pragma unroll
                for (int q = 0; q < TU; ++q) {
                    if (!on[q]) continue;                       // wave-uniform
                    const int ia = 16 * ti[q] + mi, ca = 16 * tj[q] + mi;   // panel-local row (B operand / output lanes), column (A operand)
#pragma unroll
                    for (int k0 = 0; k0 < NB; k0 += 4) {
                        if (k0 < bw) {
                            const int k = k0 + mq;
                            const bool kin = k < bw;
                            const double au = (kin && ca < nrem) ? Ua[k * pld + ca] : 0.0;
                            const double bl = (kin && ia < nrem) ? -Lp[k * pld + ia] : 0.0;
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(au, bl, acc[q], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < TU; ++q) {
                    if (!on[q]) continue;
                    const int i = ke + 16 * ti[q] + mi;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int c = ke + 16 * tj[q] + mq + 4 * v;
                        if (i < r && c < r) F[i + (long long) c * ld] = acc[q][v];
                    }
                }
            }
        }
        __syncthreads();
        if (prof) t_update += (long long) __builtin_amdgcn_s_memtime() - t_mark;
    }
    if (bad) flag_column(status, d.c0 + bad_col);
    if (prof) {
        tbuf[(long long) (first + blockIdx.x) * 8 + 2] = t_panel;
        tbuf[(long long) (first + blockIdx.x) * 8 + 3] = t_update;
        tbuf[(long long) (first + blockIdx.x) * 8 + 4] = (w + NB - 1) / NB;
    }
    CS3_WSTAMP(5);
#undef CS3_WSTAMP
}

// Blocked right-looking LU / Cholesky with ONE launch per block step.
// Launch kb does, tile by tile over the trailing matrix F[kb:, kb:]:
//   * every tile applies the update of the PREVIOUS panel  C -= L[:, kp:kb] U[kp:kb, :]
//     (K = BIG_NB, operands written by the previous launch);
//   * tiles on the new block column / block row also need the new diagonal block
//     D = F[kb:ke, kb:ke] (updated the same way); each of them recomputes it in LDS on its own
//     (cheap, and it avoids any hand-off inside the launch) and eliminates it with its tile stacked
//     underneath (eliminate32: waves 0 and 1 take 32 tile rows each), which yields
//     L[I, kb:ke] = C U_D^-1   or   U[kb:ke, J] = L_D^-1 C   directly.
// The factored D cannot be written into F during the launch (the other tiles read the
// unfactored block), so tile (0,0) parks it in dbuf and copies it into place at the NEXT launch
// (the closing launch, kb >= w: last update only, for the last block): after launch b + 1 every
// column of the factor up to block b is final in F, which is what lets the forward sweep of a
// chunk start while later blocks are still being factorised.
// Tile index 0 = the panel block [kb, ke); index t >= 1 = 64 rows/columns from ke + 64 (t-1).
template <int KIND>
__global__ void __launch_bounds__(256)
k_big_step(const FrontDesc *__restrict__ fdesc, int first, int kb, double *__restrict__ pool_all,
           long long pool_stride, double *__restrict__ dbuf_all, long long dbuf_stride,
           double inv_tol, int *status, int batch, long long *tbuf)
{
    // diagnostics (CS3_PROFILE=1): block-column tile (1, 0) of the step kb == 64 stamps its phases into the front's slot
    const long long t_start = tbuf ? (long long) __builtin_amdgcn_s_memtime() : 0;
#define CS3_BSTAMP(p) do { if (tbuf && kb == 64 && blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0) \
        tbuf[(long long) (first + blockIdx.z / batch) * 8 + (p)] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    // ... and the block-row tile (0, 1) of the same step: slots 6 (eliminated) and 7 (stored)
#define CS3_BSTAMP_ROW(p) do { if (tbuf && kb == 64 && blockIdx.x == 0 && blockIdx.y == 1 && threadIdx.x == 0) \
        tbuf[(long long) (first + blockIdx.z / batch) * 8 + (p)] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    __shared__ double As[BIG_NB][64 + 1];       // As[k][i] = L(row0 + i, kp + k)
    __shared__ double Bs[BIG_NB][64 + 1];       // Bs[k][j] = U(kp + k, col0 + j)
    __shared__ double Ad[BIG_NB][BIG_NB + 1];   // Ad[k][i] = L(kb + i, kp + k)
    __shared__ double Bd[BIG_NB][BIG_NB + 1];   // Bd[k][j] = U(kp + k, kb + j)
    __shared__ double D[BIG_NB][BIG_NB + 1];
    __shared__ double T[64][BIG_NB + 1];
    __shared__ int pair_ready[2];               // eliminate_pair: pivots whose multipliers wave 0 / 1 has handed over
    if (threadIdx.x < 2) pair_ready[threadIdx.x] = 0;           // (two block barriers lie between this and the first use)
    const FrontDesc d = fdesc[first + blockIdx.z / batch];     // grid (tiles, tiles, fronts * batch)
    const int bz = blockIdx.z % batch;
    const int r = d.r, w = d.w;
    const int nblk = (w + BIG_NB - 1) / BIG_NB;
    if (kb > nblk * BIG_NB) return;
    const bool has_panel = kb < w;
    const int ke = has_panel ? min(kb + BIG_NB, w) : w, bw = ke - kb;
    const int kp = kb - BIG_NB, pw = min(kb, w) - kp;          // previous panel [kp, kp + pw), if kb > 0
    double *F = pool_all + (long long) bz * pool_stride + d.lpan;
    double *dbuf = dbuf_all + (long long) bz * dbuf_stride + d.dbuf;
    const long long ld = r;
    const int tid = threadIdx.x;
    const int bi = blockIdx.x, bj = blockIdx.y;
    const int nsl = (r - ke + 63) / 64;
    if (bi > nsl || bj > nsl) return;
    if (KIND == CS3_CHOLESKY && bi < bj) return;
    if (kb > 0 && bi == 0 && bj == 0) {             // the diagonal block parked by the previous launch goes home:
        const int blk = kb / BIG_NB - 1;            // nobody reads that part of F any more (for the last block
        const int k0 = blk * BIG_NB, cw = min(BIG_NB, w - k0);      // this is the closing launch)
        for (int e = tid; e < cw * cw; e += 256) {
            const int i = e % cw, j = e / cw;
            if (KIND == CS3_LU || i >= j) F[(k0 + i) + (k0 + j) * ld] = dbuf[blk * (BIG_NB * BIG_NB) + i + j * BIG_NB];
        }
    }
    if (!has_panel && (bi == 0 || bj == 0)) return;
    const int row0 = (bi == 0) ? kb : ke + (bi - 1) * 64, nrow = (bi == 0) ? bw : min(64, r - row0);
    const int col0 = (bj == 0) ? kb : ke + (bj - 1) * 64, ncol = (bj == 0) ? bw : min(64, r - col0);
    const bool needs_d = has_panel && (bi == 0 || bj == 0);

    // ---- every global load of this tile goes out before anything waits for one: ONE round trip.  The loads are RAW
    // (from a safe address where the entry does not exist); their masks are applied where the values are used -- with
    // the select next to the load the compiler waited for the panel operands and stored them to LDS before it issued the
    // loads of the accumulators (two round trips; stamps: 4.9 k cycles until "loads issued").
    auto raw = [](const double *__restrict__ p, long long off, bool ok) -> double { return p[ok ? off : 0]; };
    double ra[8], rb[8], rad[4], rbd[4];
    if (kb > 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q;
            { const int i = e % 64, k = e / 64; ra[q] = raw(F, (row0 + i) + (long long) (kp + k) * ld, k < pw && i < nrow); }
            if (KIND == CS3_LU) { const int k = e % BIG_NB, j = e / BIG_NB; rb[q] = raw(F, (kp + k) + (long long) (col0 + j) * ld, k < pw && j < ncol); }
            else { const int j = e % 64, k = e / 64; rb[q] = raw(F, (col0 + j) + (long long) (kp + k) * ld, k < pw && j < ncol); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            { const int i = e % BIG_NB, k = e / BIG_NB; rad[q] = raw(F, (kb + i) + (long long) (kp + k) * ld, needs_d && k < pw && i < bw); }
            if (KIND == CS3_LU) { const int k = e % BIG_NB, j = e / BIG_NB; rbd[q] = raw(F, (kp + k) + (long long) (kb + j) * ld, needs_d && k < pw && j < bw); }
            else { const int j = e % BIG_NB, k = e / BIG_NB; rbd[q] = raw(F, (kb + j) + (long long) (kp + k) * ld, needs_d && k < pw && j < bw); }
        }
    }
    // my outputs, in the register layout of v_mfma_f64_16x16x4 with the tile transposed (A operand = U
    // panel, B operand = L panel, so that lanes % 16 run along tile ROWS and global accesses stay
    // coalesced): wave wv owns tile rows [16 wv, 16 wv + 16), acc[cb][v] = F(row0 + 16 wv + mi, col0 + 16 cb + mq + 4 v)
    const int wv = tid >> 6, mi = tid & 15, mq = (tid >> 4) & 3;
    const int ti = 16 * wv + mi;
    double4_t acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = 16 * cb + mq + 4 * v;
            acc[cb][v] = raw(F, (row0 + ti) + (long long) (col0 + j) * ld, ti < nrow && j < ncol);
        }
    // D, same scheme: wave wv owns the 16 x 16 sub-block (wv & 1, wv >> 1): dacc[v] = D(dr, dc0 + 4 v)
    const int dr = 16 * (wv & 1) + mi, dc0 = 16 * (wv >> 1) + mq;
    double4_t dacc;
#pragma unroll
    for (int v = 0; v < 4; ++v)
        dacc[v] = raw(F, (kb + dr) + (long long) (kb + dc0 + 4 * v) * ld, needs_d && dr < bw && dc0 + 4 * v < bw);
    __builtin_amdgcn_sched_barrier(0);
    CS3_BSTAMP(0);
    // (the masks now)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = 16 * cb + mq + 4 * v;
            if (!(ti < nrow && j < ncol)) acc[cb][v] = 0.0;
        }
#pragma unroll
    for (int v = 0; v < 4; ++v)
        if (!(needs_d && dr < bw && dc0 + 4 * v < bw)) dacc[v] = 0.0;
    if (kb > 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 256 * q;
            { const int i = e % 64, k = e / 64; As[e / 64][e % 64] = (k < pw && i < nrow) ? -ra[q] : 0.0; }     // negated: the MFMA accumulates acc += U' (-L)'
            if (KIND == CS3_LU) { const int k = e % BIG_NB, j = e / BIG_NB; Bs[k][j] = (k < pw && j < ncol) ? rb[q] : 0.0; }
            else { const int j = e % 64, k = e / 64; Bs[k][j] = (k < pw && j < ncol) ? rb[q] : 0.0; }
        }
        if (needs_d) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q;
                { const int i = e % BIG_NB, k = e / BIG_NB; Ad[k][i] = (k < pw && i < bw) ? -rad[q] : 0.0; }
                if (KIND == CS3_LU) { const int k = e % BIG_NB, j = e / BIG_NB; Bd[k][j] = (k < pw && j < bw) ? rbd[q] : 0.0; }
                else { const int j = e % BIG_NB, k = e / BIG_NB; Bd[k][j] = (k < pw && j < bw) ? rbd[q] : 0.0; }
            }
        }
    }
    __syncthreads();
    CS3_BSTAMP(1);
    if (kb > 0) {
#pragma unroll
        for (int k0 = 0; k0 < BIG_NB; k0 += 4) {
            const double bl = As[k0 + mq][ti];                  // B operand: -L(row ti, k0 + mq), lane mi + 16 mq
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)                      // A operand: U(k0 + mq, column 16 cb + mi)
                acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bs[k0 + mq][16 * cb + mi], bl, acc[cb], 0, 0, 0);
        }
        if (needs_d) {
#pragma unroll
            for (int k0 = 0; k0 < BIG_NB; k0 += 4)
                dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Bd[k0 + mq][16 * (wv >> 1) + mi], Ad[k0 + mq][dr], dacc, 0, 0, 0);
        }
    }
    if (!needs_d) {                                 // plain trailing tile: store and leave
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int j = 16 * cb + mq + 4 * v;
                if (ti < nrow && j < ncol && (KIND == CS3_LU || row0 + ti >= col0 + j))
                    F[(row0 + ti) + (long long) (col0 + j) * ld] = acc[cb][v];
            }
        return;
    }
    CS3_BSTAMP(2);

    // ---- the updated D (identity-padded past bw) and my tile meet in LDS
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int j = dc0 + 4 * v;
        D[dr][j] = (dr < bw && j < bw) ? dacc[v] : (dr == j ? 1.0 : 0.0);
    }
    if (bi > 0) {                                   // block-column tile: T[row][col], 32 columns
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) T[ti][16 * cb + mq + 4 * v] = acc[cb][v];
    } else if (bj > 0 && wv < 2) {                  // block-row tile: T[col][row], 32 rows
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) T[16 * cb + mq + 4 * v][ti] = acc[cb][v];
    }
    __syncthreads();
    CS3_BSTAMP(3);
    // all four waves carry on: waves 0 and 1 take 32 stacked tile rows each with columns 0..15 of the block, waves 2 and 3
    // the same rows with columns 16..31 (eliminate_pair).  The multipliers travel through the dead As / Bs buffers.
    const int lane = tid & 63, half = (tid >> 6) & 1, part = tid >> 7, li = lane & 31;
    const bool stacked = lane >= 32;
    const bool row_tile = (bi == 0 && bj > 0);      // block-row tile (LU only): eliminate D' with tile columns stacked
    const bool diag_tile = (bi == 0 && bj == 0);
    if (diag_tile && half == 1) return;
    constexpr int HC = PAIR_NC;
    const int cbase = HC * part;                    // my first column of the block
    double e[HC];
#pragma unroll
    for (int j = 0; j < HC; ++j) {
        const double dv = row_tile ? D[cbase + j][li] : D[li][cbase + j];
        const double tv = T[32 * half + li][cbase + j];
        e[j] = stacked ? (diag_tile ? 0.0 : tv) : dv;
    }
    eliminate_pair<KIND>(e, part, row_tile, (half == 0) ? &As[0][0] : &Bs[0][0], &pair_ready[half], status);
    CS3_BSTAMP(4);
    CS3_BSTAMP_ROW(6);
    if (diag_tile) {                                // park the factored block, check its pivots
        double *db = dbuf + (long long) (kb / BIG_NB) * (BIG_NB * BIG_NB);
        if (lane < bw) {
#pragma unroll
            for (int jj = 0; jj < HC; ++jj) {
                const int j = cbase + jj;
                if (j < bw) {
                    const double v = e[jj];
                    if (KIND == CS3_LU) {
                        if (lane > j) { if (!(fabs(v) <= inv_tol)) flag_column(status, d.c0 + kb + j); }
                        else if (lane == j) { if (!(fabs(v) > 0.0) || !(fabs(v) < 1.0e300)) flag_column(status, d.c0 + kb + j); }
                    } else if (lane == j) {
                        if (!(v > 0.0)) flag_column(status, d.c0 + kb + j);
                    }
                    db[lane + j * BIG_NB] = v;
                }
            }
        }
        return;
    }
    const int tr = 32 * half + li;                  // my row (block-column tile) or column (block-row tile) of the tile
    if (stacked) {
        if (!row_tile) {
            if (tr < nrow) {
#pragma unroll
                for (int cc = 0; cc < HC; ++cc) {
                    const int c = cbase + cc;
                    if (c < bw) {
                        if (KIND == CS3_LU && !(fabs(e[cc]) <= inv_tol)) flag_column(status, d.c0 + kb + c);
                        F[(row0 + tr) + (long long) (kb + c) * ld] = e[cc];
                    }
                }
            }
        } else if (tr < ncol) {
#pragma unroll
            for (int cc = 0; cc < HC; ++cc) {
                const int c = cbase + cc;
                if (c < bw) F[(kb + c) + (long long) (col0 + tr) * ld] = e[cc];
            }
        }
    }
    CS3_BSTAMP(5);
    CS3_BSTAMP_ROW(7);
#undef CS3_BSTAMP
#undef CS3_BSTAMP_ROW
}


// ------------------------------------------------------ supernodal solves --
// X is [n, nrhs] row-major in pivot order; blockIdx.z = right-hand side,
// blockIdx.y = matrix of the batch.
//   forward  (L y = b):  front vector v = [X rows of my pivots ; 0] + children's
//       contribution vectors (gather list), then column by column
//       y_k = v_k (/ L_kk for Cholesky), v_i -= L_ik y_k for every row i > k of the
//       front; pivot rows go back to X, the rest is my contribution vector.
//   backward (U x = y):  v = [X rows of my pivots ; X rows of my ancestors],
//       pivot rows minus U12 times the ancestors, then back substitution with U11.
// Panel columns are prefetched SOLVE_PF at a time so that a column step never
// waits on memory by itself.
constexpr int SOLVE_PF = 8;


// One wave per front (r <= 128, w <= 64): lane l owns rows l and l + 64, for KT
// right-hand sides at once (blockIdx.z = tile of KT columns of X): the panel is
// read once per tile.
template <int KIND, int KT>
__global__ void __launch_bounds__(256)
k_fwd_wave(const SolveDesc *__restrict__ sd, int first, int count,
           const int *__restrict__ fsrc, const int *__restrict__ ftgt, const int *__restrict__ flong,
           const double *__restrict__ pool_all, double *__restrict__ cv_all, double *__restrict__ X_all,
           int nrhs, long long pool_stride, long long cv_stride, long long x_stride)
{
    __shared__ double vs[4][KT][132];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + wv;
    if (task >= count) return;
    const SolveDesc d = sd[first + task];
    const int t0 = blockIdx.z * KT, nlive = min(KT, nrhs - t0);
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *cv = cv_all + (long long) blockIdx.y * cv_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = d.r, w = d.w;
#pragma unroll
    for (int q = 0; q < KT; ++q) { vs[wv][q][lane] = 0.0; vs[wv][q][lane + 64] = 0.0; }
    __builtin_amdgcn_wave_barrier();
    gather_front_vec<KT>(d.fasm_begin, d.fasm_count >> 6, fsrc, ftgt, flong, nlive,
                         [&](int q) -> const double * {
                             return (q >= 0) ? cv + (long long) q * nrhs + t0 : X + (long long) (~q) * nrhs + t0;
                         },
                         [&](int t, int q, double val) { vs[wv][q][t] = val; });
    __builtin_amdgcn_wave_barrier();
    double v0[KT], v1[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) { v0[q] = vs[wv][q][lane]; v1[q] = vs[wv][q][lane + 64]; }
    const double *L = pool + d.lpan;
    // Cholesky: row i < w of the panel and of the vector is divided by L_ii up front (rdg = 1 on the other rows), so that a
    // column step is one lane-to-scalar broadcast and the FMAs
    const double rdg = (KIND == CS3_CHOLESKY) ? recip_diag(L, lane, r, lane < w) : 1.0;
    if (KIND == CS3_CHOLESKY) {
#pragma unroll
        for (int q = 0; q < KT; ++q) v0[q] *= rdg;
    }
    for (int k0 = 0; k0 < w; k0 += SOLVE_PF) {
        double l0[SOLVE_PF], l1[SOLVE_PF];
#pragma unroll
        for (int j = 0; j < SOLVE_PF; ++j) {
            const int k = k0 + j;
            l0[j] = load_if(L, lane + (long long) k * r, k < w && lane < r && lane > k);
            l1[j] = load_if(L, lane + 64 + (long long) k * r, k < w && lane + 64 < r);
        }
        if (KIND == CS3_CHOLESKY) {
#pragma unroll
            for (int j = 0; j < SOLVE_PF; ++j) l0[j] *= rdg;
        }
        // columns past w hold zeros (masked loads): the steps run without a branch on the front's width
#pragma unroll
        for (int j = 0; j < SOLVE_PF; ++j) {
            const int k = (k0 + j) & 63;
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const double xk = bcast_lane(v0[q], k);
                v0[q] -= l0[j] * xk;
                v1[q] -= l1[j] * xk;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        if (q < nlive) {
            if (lane < w) X[(long long) (d.c0 + lane) * nrhs + t0 + q] = v0[q];
            if (d.parent >= 0) {
                if (lane >= w && lane < r) cv[(d.cv + lane - w) * nrhs + t0 + q] = v0[q];
                if (lane + 64 < r) cv[(d.cv + lane + 64 - w) * nrhs + t0 + q] = v1[q];
            }
        }
    }
}

template <int KIND, int KT>
__global__ void __launch_bounds__(256)
k_bwd_wave(const SolveDesc *__restrict__ sd, int first, int count, const int *__restrict__ st_idx,
           const double *__restrict__ pool_all, double *__restrict__ X_all,
           int nrhs, long long pool_stride, long long x_stride)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + wv;
    if (task >= count) return;
    const SolveDesc d = sd[first + task];
    const int t0 = blockIdx.z * KT, nlive = min(KT, nrhs - t0);
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = d.r, w = d.w;
    const int *st = st_idx + d.st;
    const int i1 = lane + 64;
    const int row0 = (lane < w) ? d.c0 + lane : st[lane < r ? lane : 0];
    const int row1 = st[i1 < r ? i1 : 0];
    double v0[KT], v1[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        v0[q] = load_if(X, (long long) row0 * nrhs + t0 + q, lane < r && q < nlive);
        v1[q] = load_if(X, (long long) row1 * nrhs + t0 + q, i1 < r && q < nlive);
    }
    const double *L = pool + d.lpan;
    const double *U = pool + d.upan;
    const double rdg = recip_diag(L, lane, r, lane < w);
    // pivot row `lane` minus U(lane, k) x_k over the ancestors k = w .. r-1
    double a0[KT], a1[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) a0[q] = a1[q] = 0.0;
    for (int k0 = w; k0 < r; k0 += SOLVE_PF) {
        double u[SOLVE_PF];
#pragma unroll
        for (int j = 0; j < SOLVE_PF; ++j) {
            const int k = k0 + j;
            const long long off = (KIND == CS3_LU) ? (long long) lane * d.u_sk + (long long) (k - w) * d.u_sj
                                                   : (long long) k + (long long) lane * r;
            u[j] = load_if((KIND == CS3_LU) ? U : L, off, k < r && lane < w);
        }
        // (entries past r are zeros: no branch on the front's order; the products of a row go to two sums in turn, so
        //  that consecutive FMAs do not wait for each other)
#pragma unroll
        for (int j = 0; j < SOLVE_PF; ++j) {
            const int k = k0 + j;
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const double xk = (k < 64) ? bcast_lane(v0[q], k) : bcast_lane(v1[q], (k - 64) & 63);
                if (j & 1) a1[q] += u[j] * xk; else a0[q] += u[j] * xk;
            }
        }
    }
    // back substitution with U11, columns w-1 .. 0: the rows of U11 and the vector are divided by the diagonal up front, a
    // column step is one lane-to-scalar broadcast and one FMA
#pragma unroll
    for (int q = 0; q < KT; ++q) v0[q] = (v0[q] - (a0[q] + a1[q])) * rdg;
    for (int k0 = w - 1; k0 >= 0; k0 -= SOLVE_PF) {
        double u[SOLVE_PF];
#pragma unroll
        for (int j = 0; j < SOLVE_PF; ++j) {
            const int k = k0 - j;
            const long long off = (KIND == CS3_LU) ? (long long) lane + (long long) k * r
                                                   : (long long) k + (long long) lane * r;
            u[j] = load_if(L, off, k >= 0 && lane < k);
        }
#pragma unroll
        for (int j = 0; j < SOLVE_PF; ++j) u[j] *= rdg;
#pragma unroll
        for (int j = 0; j < SOLVE_PF; ++j) {
            const int k = (k0 - j) & 63;
#pragma unroll
            for (int q = 0; q < KT; ++q) v0[q] -= u[j] * bcast_lane(v0[q], k);
        }
    }
#pragma unroll
    for (int q = 0; q < KT; ++q)
        if (lane < w && q < nlive) X[(long long) (d.c0 + lane) * nrhs + t0 + q] = v0[q];
}

// ---------------------------------------------------------------- many right-hand sides --
// Fronts of order r <= 64 with 16 or more right-hand sides: one wave per (front, tile of 64
// right-hand sides), LANE = RIGHT-HAND SIDE.  X and the contribution vectors are [row][rhs], so a
// row of the tile is one coalesced 512-byte access and every lane runs the same scalar recurrence
// on its own column: no cross-lane traffic in the data.  (The lane = row kernels above serve 8
// right-hand sides per pass.)
//   * the front vector sits in statically indexed registers, the assembly included: the children's
//     additions come as SLOT ROUNDS (round j = the j-th source of every row, -1 where a row has fewer),
//     so the target of an addition is a compile-time register and only the source address is data
//     (round 2; the first version kept the vector in LDS for the assembly, which capped the occupancy
//     at 10 waves per CU and paid an LDS read-modify-write per row);
//   * the panel is the same for every lane.  It is fetched ONCE, zero-padded, by one round of
//     coalesced vector loads into registers (PanelRegs) and its entries reach the FMAs by v_readlane.  Scalar loads would feed the FMAs
//     for free, but a lone wave then waits out ~60 scalar-cache misses per front (measured: 30 us
//     per launch whatever the level's size);
//   * lane t loads the slot indices of row t (one coalesced load per round), the sources are read 16
//     rows at a time (16 loads in flight); rounds beyond a row's last source read nothing.
// A matrix M(i, t), i, t < RMAX, the same for every lane, held in registers: RMAX <= 32 packs two
// columns per register (lane = i + 32 (t & 1), register = t / 2), RMAX = 64 one (lane = i).
template <int RMAX> struct PanelRegs {
    static constexpr int NREG = (RMAX <= 32) ? RMAX / 2 : RMAX;
    double reg[NREG];
    // lane's (row, column) in register q
    static __device__ __forceinline__ int row_of(int lane) { return (RMAX <= 32) ? (lane & 31) : lane; }
    static __device__ __forceinline__ int col_of(int lane, int q) { return (RMAX <= 32) ? 2 * q + (lane >> 5) : q; }
    __device__ __forceinline__ double at(int i, int t) const     // i, t compile-time after unrolling
    {
        return (RMAX <= 32) ? bcast_lane(reg[t >> 1], i + 32 * (t & 1)) : bcast_lane(reg[t], i);
    }
};

template <int KIND, int RMAX>
__global__ void __launch_bounds__(64, (RMAX <= 16) ? 4 : (RMAX <= 32) ? 2 : 1)
k_fwd_rhs(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ slots,
          const double *__restrict__ pool_all, double *cv_all, double *X_all,
          int nrhs, long long pool_stride, long long cv_stride, long long x_stride, XMap xm)
{
    const SolveDesc d = sd[first + blockIdx.x];
    const int lane = threadIdx.x;
    // fused permutation: row k of the right-hand sides in pivot order is row q[k] of the caller's array
    const double *Xin = xm.src ? xm.src + (long long) blockIdx.y * x_stride : X_all + (long long) blockIdx.y * x_stride;
    const int qrow = xm.src ? xm.q[d.c0 + (lane < d.w ? lane : 0)] : d.c0 + lane;
    const int col = blockIdx.z * 64 + lane;
    const bool live = col < nrhs;
    const long long lo = live ? col : 0;
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *cv = cv_all + (long long) blockIdx.y * cv_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = d.r, w = d.w;
    const double *L = pool + d.lpan;
    // the first rounds of slot indices: lane t holds the sources of row t (one round trip for up to SR rounds)
    constexpr int SR = 4;
    const int rounds = d.rl_count, stride = (r + 15) & ~15;
    const int *sl = slots + d.rl_begin;
    int sidx[SR];
#pragma unroll
    for (int j = 0; j < SR; ++j) sidx[j] = (j < rounds && lane < r) ? sl[j * stride + lane] : -1;
    PanelRegs<RMAX> P;                                         // P(i, k) = L(i, k), zero outside r x w
    {
        const int i = P.row_of(lane);
#pragma unroll
        for (int q = 0; q < P.NREG; ++q) {
            const int k = P.col_of(lane, q);
            P.reg[q] = load_if(L, i + (long long) k * r, i < r && k < w);
        }
    }
    double rd = 1.0;
    if (KIND == CS3_CHOLESKY) rd = L[(lane < w ? lane : 0) * (long long) (r + 1)];
    // own rows of X, zeros below them
    double v[RMAX + (RMAX == 16 ? 1 : 0)];                     // (17, not 16: hipcc turns a 16-double array into a vector
                                                               //  value and spills 3 k registers around its updates)
#pragma unroll
    for (int t = 0; t < RMAX; ++t) {
        v[t] = 0.0;
        if (t < w) v[t] = Xin[(long long) bcast_lane_i(qrow, t) * nrhs + lo];
    }
    // what the children add: round by round, 16 rows at a time -- 16 independent loads, then 16 additions into
    // statically indexed registers; a (round, 16 rows) block without any source is skipped
    for (int j0 = 0; j0 < rounds; j0 += SR) {
        if (j0 > 0) {
#pragma unroll
            for (int j = 0; j < SR; ++j) sidx[j] = (j0 + j < rounds && lane < r) ? sl[(j0 + j) * stride + lane] : -1;
        }
#pragma unroll
        for (int j = 0; j < SR; ++j) {
            if (j0 + j >= rounds) break;
            const unsigned long long has = __builtin_amdgcn_ballot_w64(sidx[j] >= 0);
#pragma unroll
            for (int t0 = 0; t0 < RMAX; t0 += 16) {
                if (t0 < r && ((has >> t0) & 0xffffull)) {
                    double val[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int sx = bcast_lane_i(sidx[j], t0 + u);
                        val[u] = load_if(cv, (long long) sx * nrhs + lo, sx >= 0);
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[t0 + u] += val[u];
                }
            }
        }
    }
    if (KIND == CS3_CHOLESKY) rd = 1.0 / rd;                   // 1 / L(k, k), pivot k in lane k
#pragma unroll
    for (int k = 0; k < RMAX; ++k) {
        if (k < w) {
            if (KIND == CS3_CHOLESKY) v[k] *= bcast_lane(rd, k);
#pragma unroll
            for (int i0 = (k + 1) & ~7; i0 < RMAX; i0 += 8) {
                if (i0 < r) {                                   // the group's panel entries first, then its FMAs (see eliminate_block)
                    double pe[8];
#pragma unroll
                    for (int i = (i0 > k + 1 ? i0 : k + 1); i < i0 + 8; ++i) pe[i - i0] = P.at(i, k);
#pragma unroll
                    for (int i = (i0 > k + 1 ? i0 : k + 1); i < i0 + 8; ++i) v[i] -= pe[i - i0] * v[k];
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int t = 0; t < RMAX; ++t)
        if (t < w) X[(long long) (d.c0 + t) * nrhs + col] = v[t];
    if (d.parent >= 0) {
#pragma unroll
        for (int t = 0; t < RMAX; ++t)
            if (t >= w && t < r) cv[(d.cv + t - w) * nrhs + col] = v[t];
    }
}

// Backward: the whole front vector x[0 .. r) in registers (pivot rows, then the ancestors' rows, which
// are final), one descending recurrence  x[t] final -> x[i] -= M(i, t) x[t]  for the pivot rows i < t,
// with M = [U11 U12] (Cholesky: [L11' L21']) and zero rows below w.
template <int KIND, int RMAX>
__global__ void __launch_bounds__(64, (RMAX <= 16) ? 4 : (RMAX <= 32) ? 2 : 1)
k_bwd_rhs(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ st_idx,
          const double *__restrict__ pool_all, double *X_all,
          int nrhs, long long pool_stride, long long x_stride, XMap xm)
{
    const SolveDesc d = sd[first + blockIdx.x];
    const int lane = threadIdx.x;
    const int col = blockIdx.z * 64 + lane;
    const bool live = col < nrhs;
    const long long lo = live ? col : 0;
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = d.r, w = d.w;
    const double *L = pool + d.lpan;
    const double *U = pool + d.upan;
    // row of X behind front position `lane`
    const int myrow = (lane < w) ? d.c0 + lane : st_idx[d.st + (lane < r ? lane : 0)];
    PanelRegs<RMAX> M;
    {
        const int i = M.row_of(lane);
#pragma unroll
        for (int q = 0; q < M.NREG; ++q) {
            const int t = M.col_of(lane, q);
            long long off;
            if (KIND == CS3_LU) off = (t < w) ? (long long) i + (long long) t * r
                                              : (long long) i * d.u_sk + (long long) (t - w) * d.u_sj;
            else off = (long long) t + (long long) i * r;
            M.reg[q] = load_if((KIND == CS3_LU && t >= w) ? U : L, off, i < w && i <= t && t < r);
        }
    }
    double rd = L[(lane < w ? lane : 0) * (long long) (r + 1)];
    double x[RMAX + (RMAX == 16 ? 1 : 0)];                     // (17: see k_fwd_rhs)
#pragma unroll
    for (int t = 0; t < RMAX; ++t) {
        x[t] = 0.0;
        if (t < r) x[t] = X[(long long) bcast_lane_i(myrow, t) * nrhs + lo];
    }
    rd = 1.0 / rd;                                             // reciprocal pivots from one vector division
#pragma unroll
    for (int t = RMAX - 1; t >= 0; --t) {
        if (t < r) {
            if (t < w) x[t] *= bcast_lane(rd, t);
#pragma unroll
            for (int i0 = 0; i0 < t; i0 += 8) {
                if (i0 < w) {                                   // the group's panel entries first, then its FMAs
                    double pe[8];
#pragma unroll
                    for (int i = i0; i < i0 + 8; ++i)
                        if (i < t) pe[i - i0] = M.at(i, t);
#pragma unroll
                    for (int i = i0; i < i0 + 8; ++i)
                        if (i < t) x[i] -= pe[i - i0] * x[t];
                }
            }
        }
    }
    // fused permutation: the solution row also goes home, to row q[k] of the caller's array.  The row map is read by ALL
    // lanes before the dead ones leave: it is fetched lane-to-scalar below, also from lanes beyond the last right-hand side
    const int qrow = xm.dst ? xm.q[d.c0 + (lane < w ? lane : 0)] : 0;
    if (!live) return;
#pragma unroll
    for (int t = 0; t < RMAX; ++t)
        if (t < w) X[(long long) (d.c0 + t) * nrhs + col] = x[t];
    if (xm.dst) {
        double *Xo = xm.dst + (long long) blockIdx.y * x_stride;
#pragma unroll
        for (int t = 0; t < RMAX; ++t)
            if (t < w) Xo[(long long) bcast_lane_i(qrow, t) * nrhs + col] = x[t];
    }
}

// One workgroup per front (any size): the front vector lives in LDS, the pivot
// block is walked in chunks of 64 columns -- wave 0 solves the 64 x 64 triangle
// with every lane owning one row (whole row prefetched, shuffles for y_k), then
// all threads apply the chunk to the rows below (one row per thread, 64 loads
// in flight).
constexpr int SOLVE_BW = 64;

// Triangle of a block [kb, kb + bw), bw <= 64, by one wave: lane = row kb + lane keeps its row of the
// triangle in registers.  FORWARD: unit lower L (Cholesky: L with its diagonal), else upper U with its
// diagonal (Cholesky: L').  Loading and solving are separate so that the loads of a second block
// are in flight while the first is being solved.
template <int KIND, bool FORWARD>
struct BlockTriangle {
    // Row `lane` of the 64 x 64 triangle, STRICTLY off the diagonal and already divided by the row's own diagonal entry
    // (backward sweep, Cholesky forward sweep), zero outside bw x bw: a substitution step is then one lane-to-scalar
    // broadcast and one FMA -- no per-step scaling, no predicate, no branch on the block's width (round 2: the chain of
    // 64 steps took 136-192 cycles per step in the root's sweeps).
    double t[SOLVE_BW];
    double rdg;                                 // 1 / diagonal entry of my row (1 where nothing is divided)
    __device__ __forceinline__ void load(const double *__restrict__ L, long long r, int kb, int bw)
    {
        const int lane = threadIdx.x & 63;
        const int i = kb + lane;
        rdg = (!FORWARD || KIND == CS3_CHOLESKY) ? recip_diag(L, i, r, lane < bw) : 1.0;
#pragma unroll
        for (int j = 0; j < SOLVE_BW; ++j) {
            long long off;
            bool need;
            if (FORWARD) { off = (long long) i + (long long) (kb + j) * r; need = lane > j; }
            else {
                off = (KIND == CS3_LU) ? (long long) i + (long long) (kb + j) * r : (long long) (kb + j) + (long long) i * r;
                need = lane < j;
            }
            t[j] = load_if(L, off, j < bw && lane < bw && need);
        }
        if (!FORWARD || KIND == CS3_CHOLESKY) {
#pragma unroll
            for (int j = 0; j < SOLVE_BW; ++j) t[j] *= rdg;
        }
    }
    // KT right-hand sides at once: one pass over the triangle, KT broadcasts per step.  Lanes >= bw of vi must hold
    // zeros (every caller masks its loads).
    template <int KT>
    __device__ __forceinline__ void solve_multi(double (&vi)[KT], int) const
    {
        if (!FORWARD || KIND == CS3_CHOLESKY) {
#pragma unroll
            for (int q = 0; q < KT; ++q) vi[q] *= rdg;
        }
#pragma unroll
        for (int jj = 0; jj < SOLVE_BW; ++jj) {
            const int j = FORWARD ? jj : SOLVE_BW - 1 - jj;
            double xk[KT];
#pragma unroll
            for (int q = 0; q < KT; ++q) xk[q] = bcast_lane(vi[q], j);
#pragma unroll
            for (int q = 0; q < KT; ++q) vi[q] -= t[j] * xk[q];
        }
    }
    __device__ __forceinline__ double solve(double vi, int) const
    {
        if (!FORWARD || KIND == CS3_CHOLESKY) vi *= rdg;
#pragma unroll
        for (int jj = 0; jj < SOLVE_BW; ++jj) {
            const int j = FORWARD ? jj : SOLVE_BW - 1 - jj;
            vi -= t[j] * bcast_lane(vi, j);
        }
        return vi;
    }
};

template <int KIND>
__global__ void __launch_bounds__(256)
k_fwd_blk(const SolveDesc *__restrict__ sd, int first,
          const int *__restrict__ fsrc, const int *__restrict__ ftgt, const int *__restrict__ flong,
          const double *__restrict__ pool_all, double *__restrict__ cv_all, double *__restrict__ X_all,
          int nrhs, long long pool_stride, long long cv_stride, long long x_stride)
{
    extern __shared__ __attribute__((aligned(16))) double v[];       // [r + 1] then y[64]
    const SolveDesc d = sd[first + blockIdx.x];
    const int rhs = blockIdx.z;
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *cv = cv_all + (long long) blockIdx.y * cv_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = d.r, w = d.w;
    const int tid = threadIdx.x, lane = tid & 63;
    double *y = v + r + 1;
    const double *L = pool + d.lpan;
    BlockTriangle<KIND, true> tri;                 // wave 0: the first triangle's loads travel with the assembly's
    if (tid < 64) tri.load(L, r, 0, min(SOLVE_BW, w));
    for (int i = tid; i < r; i += 256) v[i] = 0.0;
    __syncthreads();
    gather_front(d.fasm_begin, d.fasm_count >> 6, tid >> 6, 4, fsrc, ftgt, flong,
                 [&](int q) -> const double * {
                     return (q >= 0) ? cv + (long long) q * nrhs + rhs : X + (long long) (~q) * nrhs + rhs;
                 },
                 [&](int t, double val) { v[t] = val; });
    __syncthreads();
    for (int kb = 0; kb < w; kb += SOLVE_BW) {
        const int bw = min(SOLVE_BW, w - kb);
        if (tid < 64) {                            // triangle [kb, kb + bw): lane = row kb + lane
            const int i = kb + lane;
            if (kb > 0) tri.load(L, r, kb, bw);    // (the first one came in with the assembly)
            const double vi = tri.solve((lane < bw) ? v[i] : 0.0, bw);
            if (lane < bw) v[i] = vi;
            y[lane] = (lane < bw) ? vi : 0.0;      // all 64: the products below run over the whole chunk (0 x stale LDS is not 0)
        }
        __syncthreads();
        for (int i = kb + bw + tid; i < r; i += 256) {      // rows below the chunk, 16 columns of it per round trip
            double acc = 0.0;
            for (int j0 = 0; j0 < bw; j0 += 16) {
                double lv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) lv[j] = load_if(L, i + (long long) (kb + j0 + j) * r, j0 + j < bw);
#pragma unroll
                for (int j = 0; j < 16; ++j) acc += lv[j] * y[j0 + j];
            }
            v[i] -= acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < w; i += 256) X[(long long) (d.c0 + i) * nrhs + rhs] = v[i];
    if (d.parent >= 0)
        for (int i = w + tid; i < r; i += 256) cv[(d.cv + i - w) * nrhs + rhs] = v[i];
}

template <int KIND>
__global__ void __launch_bounds__(256)
k_bwd_blk(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ st_idx,
          const double *__restrict__ pool_all, double *__restrict__ X_all,
          int nrhs, long long pool_stride, long long x_stride)
{
    extern __shared__ __attribute__((aligned(16))) double v[];       // [r + 1] then y[64]
    const SolveDesc d = sd[first + blockIdx.x];
    const int rhs = blockIdx.z;
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *X = X_all + (long long) blockIdx.y * x_stride;
    const int r = d.r, w = d.w, nb = r - w;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int *st = st_idx + d.st;
    double *y = v + r + 1;
    const double *L = pool + d.lpan;
    const double *U = pool + d.upan;
    const int nchunk = (w + SOLVE_BW - 1) / SOLVE_BW;
    BlockTriangle<KIND, false> tri;                // wave 0: the rightmost triangle's loads travel with the front vector's
    if (tid < 64) tri.load(L, r, (nchunk - 1) * SOLVE_BW, w - (nchunk - 1) * SOLVE_BW);
    for (int i = tid; i < r; i += 256) {
        const long long row = (i < w) ? d.c0 + i : st[i];
        v[i] = X[row * nrhs + rhs];
    }
    __syncthreads();
    // v1 -= U12 v2: one wave per pivot row, lanes across the ancestors, fixed-order reduction.  A wave takes RB of its
    // rows and JU strides of 64 ancestors per pass and issues those RB x JU loads together (one round trip instead of
    // one per row); every row still sums its products in ascending order of the ancestors, as before.
    {
        constexpr int RB = 8, JU = 4;
        for (int i0 = wv; i0 < w; i0 += 4 * RB) {
            double acc[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) acc[q] = 0.0;
            for (int j0 = 0; j0 < nb; j0 += 64 * JU) {
                double u[RB][JU];
#pragma unroll
                for (int q = 0; q < RB; ++q)
#pragma unroll
                    for (int t = 0; t < JU; ++t) {
                        const int i = i0 + 4 * q, j = j0 + 64 * t + lane;
                        const long long off = (KIND == CS3_LU) ? (long long) i * d.u_sk + (long long) j * d.u_sj
                                                               : (long long) (w + j) + (long long) i * r;
                        u[q][t] = load_if((KIND == CS3_LU) ? U : L, off, i < w && j < nb);
                    }
#pragma unroll
                for (int t = 0; t < JU; ++t) {
                    const int j = j0 + 64 * t + lane;
                    const double vj = (j < nb) ? v[w + j] : 0.0;
#pragma unroll
                    for (int q = 0; q < RB; ++q) acc[q] += u[q][t] * vj;
                }
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                double a = acc[q];
                for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
                if (lane == 0 && i0 + 4 * q < w) v[i0 + 4 * q] -= a;
            }
        }
    }
    __syncthreads();
    // back substitution, chunks of 64 columns from the right
    for (int c = nchunk - 1; c >= 0; --c) {
        const int kb = c * SOLVE_BW, bw = min(SOLVE_BW, w - kb);
        if (tid < 64) {                            // triangle: lane = row kb + lane, columns kb + bw - 1 .. kb
            const int i = kb + lane;
            if (c < nchunk - 1) tri.load(L, r, kb, bw);     // (the rightmost one came in with the front vector)
            const double vi = tri.solve((lane < bw) ? v[i] : 0.0, bw);
            if (lane < bw) v[i] = vi;
            y[lane] = (lane < bw) ? vi : 0.0;      // all 64 (see k_fwd_blk)
        }
        __syncthreads();
        for (int i = tid; i < kb; i += 256) {      // pivot rows above the chunk
            double acc = 0.0;
            for (int j0 = 0; j0 < bw; j0 += 16) {
                double uv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const long long off = (KIND == CS3_LU) ? (long long) i + (long long) (kb + j0 + j) * r
                                                           : (long long) (kb + j0 + j) + (long long) i * r;
                    uv[j] = load_if(L, off, j0 + j < bw);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) acc += uv[j] * y[j0 + j];
            }
            v[i] -= acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < w; i += 256) X[(long long) (d.c0 + i) * nrhs + rhs] = v[i];
}

// Wide big fronts (w > 64, r > 136): the single-workgroup sweep above is bound by what one CU
// can pull from memory, so these fronts take ONE LAUNCH PER 64-COLUMN CHUNK with many
// workgroups, as the factorisation does: the front vector lives in HBM (bigv); in chunk launch c
// every workgroup solves the 64 x 64 triangle of the chunk on its own (wave 0, rows in registers)
// and then applies the chunk to ITS slice of 64 rows (4 threads per row, 16 columns each,
// partial sums combined in a fixed order).  blockIdx.y = matrix * nrhs + right-hand side.
__global__ void __launch_bounds__(256)
k_fwd_big_gather(const SolveDesc *__restrict__ sd, int first,
                 const int *__restrict__ fsrc, const int *__restrict__ ftgt, const int *__restrict__ flong,
                 const double *__restrict__ cv_all, const double *__restrict__ X_all, double *__restrict__ bigv_all,
                 int nrhs, long long cv_stride, long long x_stride, long long bv_size)
{
    const SolveDesc d = sd[first + blockIdx.z];
    const int b = blockIdx.y / nrhs, rhs = blockIdx.y % nrhs;
    const double *cv = cv_all + (long long) b * cv_stride;
    const double *X = X_all + (long long) b * x_stride;
    double *v = bigv_all + ((long long) b * nrhs + rhs) * bv_size + d.bv;
    // one workgroup per front vector: rows that no child updates have no source in the gather list and start from zero
    // (the zeroing was a hipMemsetAsync before the launch: two fill kernels, 10 us on the overlapped sweep's branch)
    for (int i = threadIdx.x; i < d.r; i += 256) v[i] = 0.0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    gather_front(d.fasm_begin, d.fasm_count >> 6, wave, 4, fsrc, ftgt, flong,
                 [&](int q) -> const double * {
                     return (q >= 0) ? cv + (long long) q * nrhs + rhs : X + (long long) (~q) * nrhs + rhs;
                 },
                 [&](int t, double val) { v[t] = val; });
}

// One launch per chunk of BIG_CW = 128 pivot columns = two blocks of 64.  Every workgroup solves the
// chunk on its own: wave 0 the first block, wave 1 the second after taking the first block's
// solution out of its rows (64 x 64, operands prefetched), then all four waves apply the chunk to
// the workgroup's slice of 64 rows (4 threads per row, 32 columns each, prefetched before the
// triangles; partial sums combined in a fixed order).
// CW = 128 when the sweep is latency-bound (few right-hand sides: half the dependent launches),
// CW = 64 when there are enough right-hand sides to fill the chip (no idle second block).
constexpr int BIG_CW = 2 * SOLVE_BW;

template <int KIND, int CW>
__global__ void __launch_bounds__(256)
k_fwd_big_step(const SolveDesc *__restrict__ sd, int first, int kb,
               const double *__restrict__ pool_all, double *__restrict__ cv_all, double *__restrict__ X_all,
               double *__restrict__ bigv_all, int nrhs, long long pool_stride, long long cv_stride,
               long long x_stride, long long bv_size)
{
    constexpr int SC = CW / 4;                      // chunk columns per wave in the slice update
    __shared__ double y[BIG_CW];
    __shared__ double cpl[64];                      // second block's right-hand side after the coupling (wave 2 -> wave 1)
    __shared__ double part[4][64];
    const SolveDesc d = sd[first + blockIdx.z];
    const int r = d.r, w = d.w;
    if (kb >= w) return;
    const int bw = min(CW, w - kb), ke = kb + bw;
    const int bwa = min(SOLVE_BW, bw), bwb = bw - bwa;
    const int nsl = (r - ke + 63) / 64;
    if ((int) blockIdx.x > 0 && (int) blockIdx.x >= nsl) return;
    const int b = blockIdx.y / nrhs, rhs = blockIdx.y % nrhs;
    const double *L = pool_all + (long long) b * pool_stride + d.lpan;
    double *v = bigv_all + ((long long) b * nrhs + rhs) * bv_size + d.bv;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // my slice of rows below the chunk: wave wv takes columns [SC wv, SC wv + SC) of the chunk
    const int row = ke + blockIdx.x * 64 + lane;
    double ls[SC];
#pragma unroll
    for (int j = 0; j < SC; ++j) {
        const int jj = wv * SC + j;
        ls[j] = load_if(L, (long long) row + (long long) (kb + jj) * r, row < r && jj < bw);
    }
    if (wv == 0) {
        BlockTriangle<KIND, true> ta;
        ta.load(L, r, kb, bwa);
        const double vi = ta.solve(load_if(v, kb + lane, lane < bwa), bwa);
        y[lane] = (lane < bwa) ? vi : 0.0;          // all 64: wave 2 multiplies the whole block (0 x stale LDS is not 0)
        __syncthreads();
        __syncthreads();
    } else if (wv == 1) {                           // second block: its triangle; the first block's solution reaches its
        BlockTriangle<KIND, true> tb;               //   rows through wave 2, so that no wave holds two 64 x 64 operands
        tb.load(L, r, kb + SOLVE_BW, bwb);
        __syncthreads();
        __syncthreads();
        const double vi = tb.solve(cpl[lane], bwb);
        if (lane < bwb) y[SOLVE_BW + lane] = vi;
    } else if (wv == 2) {
        double tm[SOLVE_BW];                        // L(second block row, first block columns)
#pragma unroll
        for (int j = 0; j < SOLVE_BW; ++j)
            tm[j] = load_if(L, (long long) (kb + SOLVE_BW + lane) + (long long) (kb + j) * r, lane < bwb);
        double vi = load_if(v, kb + SOLVE_BW + lane, lane < bwb);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SOLVE_BW; ++j) vi -= tm[j] * y[j];
        cpl[lane] = vi;
        __syncthreads();
    } else {
        __syncthreads();
        __syncthreads();
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < bw)
        X_all[(long long) b * x_stride + (long long) (d.c0 + kb + tid) * nrhs + rhs] = y[tid];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SC; ++j) {
        const int jj = wv * SC + j;
        acc += ls[j] * y[jj < bw ? jj : 0];          // ls[j] = 0 past the chunk
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (tid < 64 && row < r) {
        const double nv = v[row] - (((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]);
        v[row] = nv;
        if (ke >= w && row >= w && d.parent >= 0)           // last chunk: rows below the pivots are final
            cv_all[(long long) b * cv_stride + (d.cv + row - w) * nrhs + rhs] = nv;
    }
}

// Many right-hand sides: the same chunk step (64 columns) for BIG_KT right-hand sides per workgroup.  With
// one right-hand side per workgroup every one of them re-reads the chunk's triangle and its slice of the
// panel from L2 (64 KB each, 64 MB per launch at 128 right-hand sides: that traffic, not latency, set the
// launch time); here the operands are read once and applied to BIG_KT vectors.
constexpr int BIG_KT = 8;

template <int KIND>
__global__ void __launch_bounds__(256)
k_fwd_big_step_multi(const SolveDesc *__restrict__ sd, int first, int kb,
                     const double *__restrict__ pool_all, double *__restrict__ cv_all, double *__restrict__ X_all,
                     double *__restrict__ bigv_all, int nrhs, long long pool_stride, long long cv_stride,
                     long long x_stride, long long bv_size)
{
    __shared__ double y[BIG_KT][SOLVE_BW];
    __shared__ double part[4][BIG_KT][64];
    const SolveDesc d = sd[first + blockIdx.z];
    const int r = d.r, w = d.w;
    if (kb >= w) return;
    const int bw = min(SOLVE_BW, w - kb), ke = kb + bw;
    const int nsl = (r - ke + 63) / 64;
    if ((int) blockIdx.x > 0 && (int) blockIdx.x >= nsl) return;
    const int ntile = (nrhs + BIG_KT - 1) / BIG_KT;
    const int b = blockIdx.y / ntile, rhs0 = (blockIdx.y % ntile) * BIG_KT, nlive = min(BIG_KT, nrhs - rhs0);
    const double *L = pool_all + (long long) b * pool_stride + d.lpan;
    double *v0 = bigv_all + ((long long) b * nrhs + rhs0) * bv_size + d.bv;       // vector q at v0 + q * bv_size
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = ke + blockIdx.x * 64 + lane;
    double ls[16];                                                             // my slice: columns [16 wv, 16 wv + 16)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int jj = wv * 16 + j;
        ls[j] = load_if(L, (long long) row + (long long) (kb + jj) * r, row < r && jj < bw);
    }
    {                                               // every wave solves the triangle for BIG_KT / 4 of the vectors
        constexpr int QW = BIG_KT / 4;
        BlockTriangle<KIND, true> ta;
        ta.load(L, r, kb, bw);
        double vi[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u) {                   // a dead slot reads slot 0: past nrhs there is no vector
            const int q = wv * QW + u;
            vi[u] = load_if(v0 + (long long) (q < nlive ? q : 0) * bv_size, kb + lane, lane < bw && q < nlive);
        }
        ta.template solve_multi<QW>(vi, bw);
#pragma unroll
        for (int u = 0; u < QW; ++u) y[wv * QW + u][lane] = (lane < bw) ? vi[u] : 0.0;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < bw)
        for (int q = 0; q < nlive; ++q)
            X_all[(long long) b * x_stride + (long long) (d.c0 + kb + tid) * nrhs + rhs0 + q] = y[q][tid];
#pragma unroll
    for (int q = 0; q < BIG_KT; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += ls[j] * y[q][wv * 16 + j];              // ls[j] = 0 past the chunk
        part[wv][q][lane] = acc;
    }
    __syncthreads();
    if (tid < 64 && row < r) {
        for (int q = 0; q < nlive; ++q) {
            double *vq = v0 + (long long) q * bv_size;
            const double nv = vq[row] - (((part[0][q][tid] + part[1][q][tid]) + part[2][q][tid]) + part[3][q][tid]);
            vq[row] = nv;
            if (ke >= w && row >= w && d.parent >= 0)
                cv_all[(long long) b * cv_stride + (d.cv + row - w) * nrhs + rhs0 + q] = nv;
        }
    }
}

template <int KIND>
__global__ void __launch_bounds__(256)
k_bwd_big_step_multi(const SolveDesc *__restrict__ sd, int first, int chunk_from_right,
                     const double *__restrict__ pool_all, double *__restrict__ X_all, double *__restrict__ bigv_all,
                     int nrhs, long long pool_stride, long long x_stride, long long bv_size)
{
    __shared__ double y[BIG_KT][SOLVE_BW];
    __shared__ double part[4][BIG_KT][64];
    const SolveDesc d = sd[first + blockIdx.z];
    const int r = d.r, w = d.w;
    const int nchunk = (w + SOLVE_BW - 1) / SOLVE_BW;
    const int c = nchunk - 1 - chunk_from_right;
    if (c < 0) return;
    const int kb = c * SOLVE_BW, bw = min(SOLVE_BW, w - kb);
    const int nsl = (kb + 63) / 64;
    if ((int) blockIdx.x > 0 && (int) blockIdx.x >= nsl) return;
    const int ntile = (nrhs + BIG_KT - 1) / BIG_KT;
    const int b = blockIdx.y / ntile, rhs0 = (blockIdx.y % ntile) * BIG_KT, nlive = min(BIG_KT, nrhs - rhs0);
    const double *L = pool_all + (long long) b * pool_stride + d.lpan;
    double *v0 = bigv_all + ((long long) b * nrhs + rhs0) * bv_size + d.bv;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = blockIdx.x * 64 + lane;
    double us[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int jj = wv * 16 + j;
        const long long off = (KIND == CS3_LU) ? (long long) row + (long long) (kb + jj) * r
                                               : (long long) (kb + jj) + (long long) row * r;
        us[j] = load_if(L, off, row < kb && jj < bw);
    }
    {                                               // every wave solves the triangle for BIG_KT / 4 of the vectors
        constexpr int QW = BIG_KT / 4;
        BlockTriangle<KIND, false> ta;
        ta.load(L, r, kb, bw);
        double vi[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u) {                   // a dead slot reads slot 0: past nrhs there is no vector
            const int q = wv * QW + u;
            vi[u] = load_if(v0 + (long long) (q < nlive ? q : 0) * bv_size, kb + lane, lane < bw && q < nlive);
        }
        ta.template solve_multi<QW>(vi, bw);
#pragma unroll
        for (int u = 0; u < QW; ++u) y[wv * QW + u][lane] = (lane < bw) ? vi[u] : 0.0;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < bw)
        for (int q = 0; q < nlive; ++q)
            X_all[(long long) b * x_stride + (long long) (d.c0 + kb + tid) * nrhs + rhs0 + q] = y[q][tid];
#pragma unroll
    for (int q = 0; q < BIG_KT; ++q) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += us[j] * y[q][wv * 16 + j];
        part[wv][q][lane] = acc;
    }
    __syncthreads();
    if (tid < 64 && row < kb)
        for (int q = 0; q < nlive; ++q) {
            double *vq = v0 + (long long) q * bv_size;
            vq[row] -= ((part[0][q][tid] + part[1][q][tid]) + part[2][q][tid]) + part[3][q][tid];
        }
}

// v = [X rows of my pivots ; X rows of my ancestors], pivot rows minus U12 times the ancestors.
template <int KIND>
__global__ void __launch_bounds__(256)
k_bwd_big_init(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ st_idx,
               const double *__restrict__ pool_all, const double *__restrict__ X_all, double *__restrict__ bigv_all,
               int nrhs, long long pool_stride, long long x_stride, long long bv_size)
{
    extern __shared__ __attribute__((aligned(16))) double v2[];      // [nb]
    const SolveDesc d = sd[first + blockIdx.z];
    const int b = blockIdx.y / nrhs, rhs = blockIdx.y % nrhs;
    const int r = d.r, w = d.w, nb = r - w;
    const double *pool = pool_all + (long long) b * pool_stride;
    const double *X = X_all + (long long) b * x_stride;
    double *v = bigv_all + ((long long) b * nrhs + rhs) * bv_size + d.bv;
    const int *st = st_idx + d.st;
    const int tid = threadIdx.x;
    for (int j = tid; j < nb; j += 256) v2[j] = X[(long long) st[w + j] * nrhs + rhs];
    __syncthreads();
    const double *L = pool + d.lpan;
    const double *U = pool + d.upan;
    // one pivot row per thread; for a fixed ancestor column the rows are contiguous in memory
    for (int i = blockIdx.x * 256 + tid; i < w; i += gridDim.x * 256) {
        double acc = 0.0;
        for (int j0 = 0; j0 < nb; j0 += 16) {
            double u[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int jj = j0 + j;
                const long long off = (KIND == CS3_LU) ? (long long) i * d.u_sk + (long long) jj * d.u_sj
                                                       : (long long) (w + jj) + (long long) i * r;
                u[j] = load_if((KIND == CS3_LU) ? U : L, off, jj < nb);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += u[j] * v2[j0 + j < nb ? j0 + j : 0];
        }
        v[i] = X[(long long) (d.c0 + i) * nrhs + rhs] - acc;
    }
}

template <int KIND, int CW>
__global__ void __launch_bounds__(256)
k_bwd_big_step(const SolveDesc *__restrict__ sd, int first, int chunk_from_right,
               const double *__restrict__ pool_all, double *__restrict__ X_all, double *__restrict__ bigv_all,
               int nrhs, long long pool_stride, long long x_stride, long long bv_size)
{
    constexpr int SC = CW / 4;
    __shared__ double y[BIG_CW];
    __shared__ double cpl[64];                      // second block's right-hand side after the coupling (wave 2 -> wave 1)
    __shared__ double part[4][64];
    const SolveDesc d = sd[first + blockIdx.z];
    const int r = d.r, w = d.w;
    const int nchunk = (w + CW - 1) / CW;
    const int c = nchunk - 1 - chunk_from_right;
    if (c < 0) return;
    const int kb = c * CW, bw = min(CW, w - kb);
    const int bwa = min(SOLVE_BW, bw), bwb = bw - bwa;     // blocks [kb, kb + bwa) and [kb + 64, kb + 64 + bwb)
    const int nsl = (kb + 63) / 64;                        // slices of rows above the chunk
    if ((int) blockIdx.x > 0 && (int) blockIdx.x >= nsl) return;
    const int b = blockIdx.y / nrhs, rhs = blockIdx.y % nrhs;
    const double *L = pool_all + (long long) b * pool_stride + d.lpan;
    double *v = bigv_all + ((long long) b * nrhs + rhs) * bv_size + d.bv;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = blockIdx.x * 64 + lane;
    double us[SC];                                         // U(row, chunk columns [SC wv, SC wv + SC))
#pragma unroll
    for (int j = 0; j < SC; ++j) {
        const int jj = wv * SC + j;
        const long long off = (KIND == CS3_LU) ? (long long) row + (long long) (kb + jj) * r
                                               : (long long) (kb + jj) + (long long) row * r;
        us[j] = load_if(L, off, row < kb && jj < bw);
    }
    if (wv == 0) {                                         // right block first
        if (bwb > 0) {
            BlockTriangle<KIND, false> tb;
            tb.load(L, r, kb + SOLVE_BW, bwb);
            const double vi = tb.solve(load_if(v, kb + SOLVE_BW + lane, lane < bwb), bwb);
            if (lane < bwb) y[SOLVE_BW + lane] = vi;
        }
        __syncthreads();
        __syncthreads();
    } else if (wv == 1) {                                  // left block: its triangle; the right block's solution reaches
        BlockTriangle<KIND, false> ta;                     //   its rows through wave 2 (no wave holds two 64 x 64 operands)
        ta.load(L, r, kb, bwa);
        __syncthreads();
        __syncthreads();
        const double vi = ta.solve(cpl[lane], bwa);
        if (lane < bwa) y[lane] = vi;
    } else if (wv == 2) {
        double tm[SOLVE_BW];                               // U(left block row, right block columns)
#pragma unroll
        for (int j = 0; j < SOLVE_BW; ++j) {
            const long long i = kb + lane, k = kb + SOLVE_BW + j;
            const long long off = (KIND == CS3_LU) ? i + k * r : k + i * r;
            tm[j] = load_if(L, off, lane < bwa && j < bwb);
        }
        double vi = load_if(v, kb + lane, lane < bwa);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SOLVE_BW; ++j) vi -= tm[j] * ((j < bwb) ? y[SOLVE_BW + j] : 0.0);
        cpl[lane] = vi;
        __syncthreads();
    } else {
        __syncthreads();
        __syncthreads();
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < bw)
        X_all[(long long) b * x_stride + (long long) (d.c0 + kb + tid) * nrhs + rhs] = y[tid];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < SC; ++j) {
        const int jj = wv * SC + j;
        acc += us[j] * y[jj < bw ? jj : 0];
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (tid < 64 && row < kb)
        v[row] -= ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
}

// ============================================================ lane = matrix ==
// Batches of 64 or more matrices that share a pattern: the fronts of order <= 32 (nine in ten of all
// fronts, and most of the factor entries of a power-grid matrix) are stored MATRIX-INTERLEAVED -- entry e of the
// dense r x r buffer of a front, for the 64 matrices of a group, is 64 consecutive doubles -- and one wave
// owns one front of one GROUP: lane = matrix.  Every lane runs the same scalar algorithm on its own matrix:
// no cross-lane traffic in the data, every access of the wave is one coalesced 512-byte line set, and the index
// lists (which are the same for all matrices) are read once per 64 matrices instead of once per matrix.  With
// one wave per (front, matrix) these fronts are instruction- and latency-bound (three dependent memory rounds
// and a 32-step padded elimination for a front that holds 100 entries).
//   * assembly: (target, source) pairs sorted by target in which every stored entry of the front appears
//     (Symbolic::ila_pairs); a run is summed in a register and stored once.  Sources in the interleaved block
//     are coalesced; A entries and contribution blocks of larger children are per-matrix (strided over the lanes).
//   * factorisation: right-looking, IL_KB pivots per step, in place in the interleaved buffer (which sits in
//     this CU's L1 / the XCD's L2); registers hold one tile at a time, statically indexed: the diagonal block,
//     then 8-row strips of the block column (the triangular solve X U11 = T by substitution), 8-column strips of
//     the block row (LU), then the 8 x 8 tiles of the trailing matrix with the rank-IL_KB update.
// Dead lanes of the last group shadow the group's first matrix and store into their own (allocated) slots.
constexpr int IL_KB = 4;
constexpr int IL_T = 8;        // rows per strip / tile
constexpr int IL_TJ = 4;       // columns per tile of the trailing update
constexpr int IL_NONE = INT32_MIN;             // pair source: no source (cs3::IL_ZERO)
constexpr int IL_AV = 32;                      // assembly: sources in flight per round trip

template <int KIND>
__global__ void __launch_bounds__(64)
k_front_il(const FrontDesc *__restrict__ fdesc, int first, const int *__restrict__ pairs,
           const double *__restrict__ ax_all, long long nnz_a, IlView il,
           const double *__restrict__ pool_pm_all, long long pm_stride, int batch, double inv_tol, int *status)
{
    const FrontDesc d = fdesc[first + blockIdx.x];
    const int lane = threadIdx.x;
    const long long m = (long long) blockIdx.y * 64 + lane;
    const bool live = m < batch;
    const long long mm = live ? m : (long long) blockIdx.y * 64;
    double *G = il.base + (long long) blockIdx.y * 64 * il.len + lane;        // entry off of my matrix: G[off * 64]
    const double *P = pool_pm_all + mm * pm_stride;                           // per-matrix part, virtual offsets
    const double *ax = ax_all + mm * nnz_a;
    const int r = d.r, w = d.w;
    double *F = G + d.lpan * 64;                                              // element (i, j) of the front: F[(i + j r) 64]
#define EL(i, j) F[((long long) (i) + (long long) (j) * r) * 64]

    // ---- assembly: sum every run of equal targets in a register, store it once
    {
        const int np = d.a_count;                                             // multiple of 16
        const int *pr = pairs + 2 * d.a_begin;
        int cur = -1;
        double acc = 0.0;
        int ptg_n = pr[2 * (lane < min(64, np) ? lane : 0)], psr_n = pr[2 * (lane < min(64, np) ? lane : 0) + 1];
        for (int base = 0; base < np; base += 64) {
            const int have = min(64, np - base);
            const int ptg = ptg_n, psr = psr_n;                               // pair `lane` of this block of 64
            if (base + 64 < np) {                                             // the next block's pairs travel meanwhile
                const int have_n = min(64, np - base - 64);
                const int e_n = base + 64 + (lane < have_n ? lane : 0);
                ptg_n = pr[2 * e_n]; psr_n = pr[2 * e_n + 1];
            }
            for (int c = 0; c < have; c += IL_AV) {              // `have` is a multiple of 16: the tail of a block reads pair 0
                double val[IL_AV];
#pragma unroll
                for (int u = 0; u < IL_AV; ++u) {
                    const int sr = bcast_lane_i(psr, (c + u) & 63);           // wave-uniform
                    const bool none = sr == IL_NONE;
                    const int sc = none ? 0 : sr;
                    const double *p = (sc >= 0) ? ((sc < il.len) ? (const double *) (G + (long long) sc * 64) : P + sc) : ax + ~sc;
                    const double v = *p;
                    val[u] = none ? 0.0 : v;
                }
#pragma unroll
                for (int u = 0; u < IL_AV; ++u) {
                    if (c + u >= have) break;                                 // wave-uniform
                    const int tg = bcast_lane_i(ptg, c + u);
                    if (tg != cur) {                                          // wave-uniform
                        if (cur >= 0) F[(long long) cur * 64] = acc;
                        cur = tg; acc = 0.0;
                    }
                    acc += (tg >= 0) ? val[u] : 0.0;
                }
            }
        }
        if (cur >= 0) F[(long long) cur * 64] = acc;
    }

    // ---- factorisation in place
    bool bad = false;
    int bad_col = 0;
    for (int k0 = 0; k0 < w; k0 += IL_KB) {
        const int kb = min(IL_KB, w - k0), ke = k0 + kb;
        // 1. diagonal block, identity-padded past kb: dd[jj][ii] = element (k0 + ii, k0 + jj)
        double dd[IL_KB][IL_KB], rd[IL_KB];
#pragma unroll
        for (int jj = 0; jj < IL_KB; ++jj)
#pragma unroll
            for (int ii = 0; ii < IL_KB; ++ii) {
                const bool in = ii < kb && jj < kb && (KIND == CS3_LU || ii >= jj);
                const double v = EL(in ? k0 + ii : k0, in ? k0 + jj : k0);
                dd[jj][ii] = in ? v : ((ii == jj) ? 1.0 : 0.0);
            }
#pragma unroll
        for (int kk = 0; kk < IL_KB; ++kk) {
            const double piv = dd[kk][kk];
            double dg;
            pivot_scale<KIND>(piv, dg, rd[kk]);
            bool rej = (KIND == CS3_LU) ? (!(fabs(piv) > 0.0) || !(fabs(piv) < 1.0e300)) : !(piv > 0.0);
            if (KIND == CS3_CHOLESKY) dd[kk][kk] = (piv > 0.0) ? dg : -1.0;
#pragma unroll
            for (int ii = kk + 1; ii < IL_KB; ++ii) {
                dd[kk][ii] *= rd[kk];
                if (KIND == CS3_LU) rej = rej || (ii < kb && !(fabs(dd[kk][ii]) <= inv_tol));
            }
            if (rej && kk < kb && !bad) { bad = true; bad_col = k0 + kk; }
#pragma unroll
            for (int jj = kk + 1; jj < IL_KB; ++jj)
#pragma unroll
                for (int ii = (KIND == CS3_LU ? kk + 1 : jj); ii < IL_KB; ++ii)
                    dd[jj][ii] -= dd[kk][ii] * ((KIND == CS3_LU) ? dd[jj][kk] : dd[kk][jj]);
        }
#pragma unroll
        for (int jj = 0; jj < IL_KB; ++jj)
#pragma unroll
            for (int ii = 0; ii < IL_KB; ++ii)
                if (ii < kb && jj < kb && (KIND == CS3_LU || ii >= jj)) EL(k0 + ii, k0 + jj) = dd[jj][ii];
        // 2. block column below the block, 8 rows at a time:  x U11 = t  (Cholesky: x L11' = t) by substitution
        for (int i0 = ke; i0 < r; i0 += IL_T) {
            double t[IL_T][IL_KB];
#pragma unroll
            for (int a = 0; a < IL_T; ++a)
#pragma unroll
                for (int jj = 0; jj < IL_KB; ++jj) {
                    const bool in = i0 + a < r && jj < kb;
                    const double v = EL(in ? i0 + a : i0, in ? k0 + jj : k0);
                    t[a][jj] = in ? v : 0.0;
                }
#pragma unroll
            for (int a = 0; a < IL_T; ++a)
#pragma unroll
                for (int jj = 0; jj < IL_KB; ++jj) {
                    double x = t[a][jj];
#pragma unroll
                    for (int kk = 0; kk < jj; ++kk) x -= t[a][kk] * ((KIND == CS3_LU) ? dd[jj][kk] : dd[kk][jj]);
                    x *= rd[jj];
                    t[a][jj] = x;
                    if (KIND == CS3_LU && jj < kb && i0 + a < r && !(fabs(x) <= inv_tol) && !bad) { bad = true; bad_col = k0 + jj; }
                }
#pragma unroll
            for (int a = 0; a < IL_T; ++a)
#pragma unroll
                for (int jj = 0; jj < IL_KB; ++jj)
                    if (i0 + a < r && jj < kb) EL(i0 + a, k0 + jj) = t[a][jj];
        }
        // 3. block row right of the block (LU), 8 columns at a time:  L11 u = t, unit lower
        if (KIND == CS3_LU) {
            for (int j0 = ke; j0 < r; j0 += IL_T) {
                double u[IL_T][IL_KB];
#pragma unroll
                for (int b = 0; b < IL_T; ++b)
#pragma unroll
                    for (int kk = 0; kk < IL_KB; ++kk) {
                        const bool in = j0 + b < r && kk < kb;
                        const double v = EL(in ? k0 + kk : k0, in ? j0 + b : k0);
                        u[b][kk] = in ? v : 0.0;
                    }
#pragma unroll
                for (int b = 0; b < IL_T; ++b)
#pragma unroll
                    for (int kk = 1; kk < IL_KB; ++kk) {
                        double x = u[b][kk];
#pragma unroll
                        for (int k2 = 0; k2 < kk; ++k2) x -= dd[k2][kk] * u[b][k2];
                        u[b][kk] = x;
                    }
#pragma unroll
                for (int b = 0; b < IL_T; ++b)
#pragma unroll
                    for (int kk = 1; kk < IL_KB; ++kk)
                        if (j0 + b < r && kk < kb) EL(k0 + kk, j0 + b) = u[b][kk];
            }
        }
        // 4. trailing matrix, 8 x 4 tiles:  C -= L(I, block) U(block, J)   (Cholesky: lower tiles, L(J, block)')
        for (int j0 = ke; j0 < r; j0 += IL_TJ) {
            double uc[IL_TJ][IL_KB];
#pragma unroll
            for (int b = 0; b < IL_TJ; ++b)
#pragma unroll
                for (int kk = 0; kk < IL_KB; ++kk) {
                    const bool in = j0 + b < r && kk < kb;
                    const double v = (KIND == CS3_LU) ? EL(in ? k0 + kk : k0, in ? j0 + b : k0) : EL(in ? j0 + b : k0, in ? k0 + kk : k0);
                    uc[b][kk] = in ? v : 0.0;
                }
            // Cholesky: only rows i >= j0 matter; start at the 8-row strip that holds row j0
            const int i_first = (KIND == CS3_LU) ? ke : ke + ((j0 - ke) / IL_T) * IL_T;
            for (int i0 = i_first; i0 < r; i0 += IL_T) {
                double lr[IL_T][IL_KB], c[IL_TJ][IL_T];
#pragma unroll
                for (int a = 0; a < IL_T; ++a)
#pragma unroll
                    for (int kk = 0; kk < IL_KB; ++kk) {
                        const bool in = i0 + a < r && kk < kb;
                        const double v = EL(in ? i0 + a : k0, in ? k0 + kk : k0);
                        lr[a][kk] = in ? v : 0.0;
                    }
#pragma unroll
                for (int b = 0; b < IL_TJ; ++b)
#pragma unroll
                    for (int a = 0; a < IL_T; ++a) {
                        const bool in = i0 + a < r && j0 + b < r;
                        c[b][a] = EL(in ? i0 + a : k0, in ? j0 + b : k0);
                    }
#pragma unroll
                for (int kk = 0; kk < IL_KB; ++kk)
#pragma unroll
                    for (int b = 0; b < IL_TJ; ++b)
#pragma unroll
                        for (int a = 0; a < IL_T; ++a) c[b][a] -= lr[a][kk] * uc[b][kk];
#pragma unroll
                for (int b = 0; b < IL_TJ; ++b)
#pragma unroll
                    for (int a = 0; a < IL_T; ++a)
                        if (i0 + a < r && j0 + b < r) EL(i0 + a, j0 + b) = c[b][a];
            }
        }
    }
#undef EL
    if (bad && live) flag_column(status, d.c0 + bad_col);
}

// Sweeps of the interleaved fronts, lane = matrix.  The front vector of every matrix sits in LDS (one column per
// lane: conflict-free); X and the contribution vectors stay per-matrix (a few values per front: strided over the
// lanes).  A wave is alone with its 64 matrices, so what it waits for is memory latency: both sweeps make ONE pass of w
// steps over the panel (w = pivots of the front), and the panel entries of step k + 1 -- a column of L forward, a row of
// U (Cholesky: a column of L again) backward, 512 coalesced bytes per entry -- are in flight while step k computes.
// (Round 2a walked the backward sweep by columns: r steps, each waiting for its own loads -- 68 us for one front of
// order 30.)  RMAX bounds the front order (LDS: RMAX * 512 bytes per wave).
template <int KIND, int RMAX>
__global__ void __launch_bounds__(64, 2)
k_fwd_il(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ rl, IlView il,
         double *cv_all, double *X_all, int nrhs, long long cv_stride, long long x_stride, int batch)
{
    __shared__ double vl[RMAX * 64];
    const SolveDesc d = sd[first + blockIdx.x];
    const int lane = threadIdx.x;
    const long long m = (long long) blockIdx.y * 64 + lane;
    const bool live = m < batch;
    const long long mm = live ? m : (long long) blockIdx.y * 64;
    const int r = d.r, w = d.w;
    const double *L = il.base + (long long) blockIdx.y * 64 * il.len + lane + d.lpan * 64;     // (i, k): L[(i + k r) 64]
    double *cv = cv_all + mm * cv_stride;
    double *X = X_all + mm * x_stride;
    double *v = vl + lane;                                                                      // entry t: v[t * 64]
    for (int q = 0; q < nrhs; ++q) {
#pragma unroll 8
        for (int t = 0; t < w; ++t) v[t * 64] = X[(long long) (d.c0 + t) * nrhs + q];
        for (int t = w; t < r; ++t) v[t * 64] = 0.0;
        {
            const int np = d.rl_count;                             // multiple of 16
            const int *pr = rl + 2 * d.rl_begin;
            int cur = -1;
            double acc = 0.0;
            for (int base = 0; base < np; base += 64) {
                const int have = min(64, np - base);
                const int e = base + (lane < have ? lane : 0);
                const int ptg = pr[2 * e], psr = pr[2 * e + 1];
                for (int c = 0; c < have; c += 16) {
                    double val[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) val[u] = cv[(long long) bcast_lane_i(psr, c + u) * nrhs + q];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int tg = bcast_lane_i(ptg, c + u);
                        if (tg != cur) {
                            if (cur >= 0) v[cur * 64] += acc;
                            cur = tg; acc = 0.0;
                        }
                        acc += (tg >= 0) ? val[u] : 0.0;
                    }
                }
            }
            if (cur >= 0) v[cur * 64] += acc;
        }
        // two register buffers, two columns per trip: while column k updates the vector, column k + 1 is in flight
        double ca[RMAX + 1], cb[RMAX + 1];                         // (+ 1: a power-of-two array becomes a vector value, see k_fwd_rhs)
        auto fetch = [&](int k, double (&c)[RMAX + 1]) {           // rows k .. r - 1 of column k (the diagonal included)
            const double *Lk = L + (long long) k * r * 64;
#pragma unroll
            for (int i = 0; i < RMAX; ++i) { c[i] = 1.0; if (i >= k && i < r && k < w) c[i] = Lk[(long long) i * 64]; }
        };
        auto apply = [&](int k, const double (&c)[RMAX + 1]) {
            double vk = v[k * 64], dg = 1.0;
            if (KIND == CS3_CHOLESKY) {
#pragma unroll
                for (int i = 0; i < RMAX; ++i) if (i == k) dg = c[i];
                vk *= fast_rcp(dg); v[k * 64] = vk;
            }
#pragma unroll
            for (int i0 = 0; i0 < RMAX; i0 += 8) {
                if (i0 + 8 > k + 1 && i0 < r) {                      // groups of 8 rows: at most 8 LDS values in flight
#pragma unroll
                    for (int i = i0; i < i0 + 8; ++i)
                        if (i > k && i < r) v[i * 64] -= c[i] * vk;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        fetch(0, ca);
        for (int k = 0; k < w; k += 2) {
            fetch(k + 1, cb);
            apply(k, ca);
            if (k + 1 < w) {
                fetch(k + 2, ca);
                apply(k + 1, cb);
            }
        }
        if (live) {
#pragma unroll 8
            for (int t = 0; t < w; ++t) X[(long long) (d.c0 + t) * nrhs + q] = v[t * 64];
            if (d.parent >= 0)
                for (int t = w; t < r; ++t) cv[(d.cv + t - w) * nrhs + q] = v[t * 64];
        }
    }
}

template <int KIND, int RMAX>
__global__ void __launch_bounds__(64, (RMAX <= 16) ? 4 : 2)
k_bwd_il(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ st_idx, IlView il,
         double *X_all, int nrhs, long long x_stride, int batch)
{
    __shared__ double xl[RMAX * 64];
    const SolveDesc d = sd[first + blockIdx.x];
    const int lane = threadIdx.x;
    const long long m = (long long) blockIdx.y * 64 + lane;
    const bool live = m < batch;
    const long long mm = live ? m : (long long) blockIdx.y * 64;
    const int r = d.r, w = d.w;
    const double *L = il.base + (long long) blockIdx.y * 64 * il.len + lane + d.lpan * 64;
    double *X = X_all + mm * x_stride;
    double *x = xl + lane;
    // row of X behind front position `lane` (the same for every matrix)
    const int myrow = (lane < w) ? d.c0 + lane : st_idx[d.st + (lane < r ? lane : 0)];
    // row i of M = [U11 U12] (Cholesky: [L11' L21'], i.e. column i of L): M(i, t) at L[(i + t r) 64] (Cholesky: (t + i r) 64)
    const long long ms = (KIND == CS3_LU) ? (long long) r * 64 : 64;
    auto row_of = [&](int i) -> const double * { return (KIND == CS3_LU) ? L + (long long) i * 64 : L + (long long) i * r * 64; };
    for (int q = 0; q < nrhs; ++q) {
        double rowv[RMAX + 1], nxt[RMAX + 1];              // (+ 1: see k_fwd_rhs)
        {   // the last pivot row goes out first
            const double *M = row_of(w - 1);
#pragma unroll
            for (int t = 0; t < RMAX; ++t) { rowv[t] = 0.0; if (t >= w - 1 && t < r) rowv[t] = M[(long long) t * ms]; }
        }
#pragma unroll 8
        for (int t = 0; t < r; ++t) x[t * 64] = X[(long long) bcast_lane_i(myrow, t) * nrhs + q];
        for (int i = w - 1; i >= 0; --i) {
            if (i > 0) {                                            // row i - 1: its diagonal and everything to the right
                const double *M = row_of(i - 1);
#pragma unroll
                for (int t = 0; t < RMAX; ++t) { nxt[t] = 0.0; if (t >= i - 1 && t < r) nxt[t] = M[(long long) t * ms]; }
            }
            // x_i = (x_i - sum_{t > i} M(i, t) x_t) / M(i, i): the x_t are final (pivot rows done above, ancestors' rows given)
            double acc = x[i * 64], dg = 1.0;
#pragma unroll
            for (int t = 0; t < RMAX; ++t) {
                if (t == i) dg = rowv[t];
                if (t > i && t < r) acc -= rowv[t] * x[t * 64];
            }
            x[i * 64] = acc * fast_rcp(dg);
#pragma unroll
            for (int t = 0; t < RMAX; ++t) rowv[t] = nxt[t];
        }
        if (live) {
#pragma unroll 8
            for (int t = 0; t < w; ++t) X[(long long) (d.c0 + t) * nrhs + q] = x[t * 64];
        }
    }
}

// ---------------------------------------- many right-hand sides: GEMM sweeps --
// With 16 or more right-hand sides the fronts of order > 64 (the top of the tree, the dense root) sweep as
// GEMMs on the matrix cores instead of 64-step substitutions: the 64 x 64 diagonal blocks of L and U are inverted
// once per sweep (k_inv_diag, one workgroup per block, all in one launch), and a chunk of 64 pivots then costs two
// 64 x 64 x 64 products per workgroup and tile of 64 right-hand sides,
//      Y = Linv_cc V_c      and      V_rows -= L(rows, chunk) Y         (backward: Uinv_cc, U(rows, chunk)),
// on v_mfma_f64_16x16x4 with both operands staged in LDS.  The front vector is row-major [row][rhs] (gv buffer),
// like X and the contribution vectors, so every global access of a tile is 512 contiguous bytes per row.
// A chunk step is still one launch (chunk c + 1 needs what chunk c did to the rows below it), but takes a few
// microseconds instead of the 15 to 35 of the substitution kernels, and fronts with up to 64 pivots need one.
// Every workgroup recomputes Y for its tile (cheap) rather than wait for another one.
constexpr int GC = 64;                         // pivots per chunk
constexpr int GLD = 80;                        // LDS row stride of the 64 x 64 operand tiles: 80 = 16 mod 32 doubles, so the
                                               //   two 16-lane rows of a half-wave's ds_read_b64 fall on disjoint banks
constexpr size_t GEMM_LDS = 2 * (size_t) GC * GLD * sizeof(double);     // 80 KB: two workgroups per CU

template <int KIND>
__global__ void __launch_bounds__(256)
k_inv_diag(const SolveDesc *__restrict__ sd, const int *__restrict__ tasks, int task0,
           const double *__restrict__ pool_all, long long pool_stride, double *__restrict__ dinv_all, long long dinv_stride)
{
    const int t = tasks[2 * (task0 + blockIdx.x)], c = tasks[2 * (task0 + blockIdx.x) + 1];
    const SolveDesc d = sd[t];
    const double *L = pool_all + (long long) blockIdx.y * pool_stride + d.lpan;
    double *out = dinv_all + (long long) blockIdx.y * dinv_stride + d.dinv + (long long) c * 2 * GC * GC;
    const int r = d.r, w = d.w, kb = c * GC, bw = min(GC, w - kb);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int QW = 16;                      // identity columns per wave
    {   // Linv: L x = e_j, lane = row; column j of the inverse goes to out[j * 64 + row], zero outside bw x bw
        BlockTriangle<KIND, true> ta;
        ta.load(L, r, kb, bw);
        double vi[QW];
#pragma unroll
        for (int q = 0; q < QW; ++q) vi[q] = (lane == QW * wv + q && lane < bw) ? 1.0 : 0.0;
        ta.template solve_multi<QW>(vi, bw);
#pragma unroll
        for (int q = 0; q < QW; ++q) out[(QW * wv + q) * GC + lane] = (lane < bw && QW * wv + q < bw) ? vi[q] : 0.0;
    }
    {   // Uinv (Cholesky: the inverse of L')
        BlockTriangle<KIND, false> tb;
        tb.load(L, r, kb, bw);
        double vi[QW];
#pragma unroll
        for (int q = 0; q < QW; ++q) vi[q] = (lane == QW * wv + q && lane < bw) ? 1.0 : 0.0;
        tb.template solve_multi<QW>(vi, bw);
#pragma unroll
        for (int q = 0; q < QW; ++q) out[GC * GC + (QW * wv + q) * GC + lane] = (lane < bw && QW * wv + q < bw) ? vi[q] : 0.0;
    }
}

// Front vector of the GEMM fronts, row-major [row][rhs]: one wave per row and tile of 64 right-hand sides (a row of the
// tile is one coalesced 512-byte access).  Row t = its own row of X (pivot rows, through the row map when the
// permutation is fused) + the sources of its slot rounds in order (SolveDesc::rl_begin); every row of the vector is
// written, so the buffer needs no zeroing.  Workgroup = 16 rows, 4 per wave.
__global__ void __launch_bounds__(256)
k_gemm_gather(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ slots,
              const double *__restrict__ cv_all, const double *__restrict__ X_all, double *__restrict__ gv_all,
              int nrhs, long long cv_stride, long long x_stride, long long gv_stride, XMap xm, int batch)
{
    const SolveDesc d = sd[first + blockIdx.z / batch];
    const int b = blockIdx.z % batch;
    const int r = d.r, w = d.w, rounds = d.rl_count, stride = (r + 15) & ~15;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + lane;
    const bool live = col < nrhs;
    const long long lo = live ? col : 0;
    const double *cv = cv_all + (long long) b * cv_stride;
    const double *X = (xm.src ? xm.src : X_all) + (long long) b * x_stride;
    double *V = gv_all + (long long) b * gv_stride + d.gv * (long long) nrhs;
    const int *sl = slots + d.rl_begin;
    const int t0 = blockIdx.x * 16 + wv * 4;
    int rowx[4], sv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                               // both index loads of my four rows in one round trip
        const int t = t0 + u;
        rowx[u] = (t < w) ? (xm.src ? xm.q[d.c0 + t] : d.c0 + t) : -1;
        sv[u] = (t < r && lane < rounds) ? sl[lane * stride + t] : -1;          // lane j: source of round j
    }
    double acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = load_if(X, (long long) rowx[u] * nrhs + lo, rowx[u] >= 0);
    for (int j0 = 0; j0 < rounds; j0 += 64) {
        if (j0 > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) sv[u] = (t0 + u < r && j0 + lane < rounds) ? sl[(j0 + lane) * stride + t0 + u] : -1;
        }
        const int nj = min(64, rounds - j0);
        for (int jb = 0; jb < nj; jb += 4) {
            double val[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sx = bcast_lane_i(sv[u], (jb + j) & 63);
                    val[u][j] = load_if(cv, (long long) sx * nrhs + lo, sx >= 0);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[u] += val[u][j];
        }
    }
    if (!live) return;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (t0 + u < r) V[(long long) (t0 + u) * nrhs + col] = acc[u];
}

// 64 x 64 x 64 product on the matrix cores, operands in LDS: acc(i, n) += sgn * A(i, k) B(k, n), with
// A at As[k * GLD + i], B at Bs[k * GLD + n]; wave wv owns rows 16 wv .. 16 wv + 15, acc[nt][v] = entry
// (16 wv + mq + 4 v, 16 nt + mi) in the register layout of v_mfma_f64_16x16x4.
template <bool NEG>
__device__ __forceinline__ void gemm64(const double *As, const double *Bs, double4_t (&acc)[4], int wv, int mi, int mq)
{
#pragma unroll
    for (int k0 = 0; k0 < GC; k0 += 4) {
        const double a0 = As[(k0 + mq) * GLD + 16 * wv + mi];
        const double a = NEG ? -a0 : a0;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[(k0 + mq) * GLD + 16 * nt + mi], acc[nt], 0, 0, 0);
    }
}

// The same product by EIGHT waves (two per SIMD: a wave issues one f64 MFMA per 64 cycles, two interleave to 32): wave
// (rb, ch) owns rows 16 rb .. 16 rb + 15 and right-hand sides 32 ch .. 32 ch + 31, acc[t][v] = entry
// (16 rb + mq + 4 v, 16 (2 ch + t) + mi).
template <bool NEG>
__device__ __forceinline__ void gemm64h(const double *As, const double *Bs, double4_t (&acc)[2], int rb, int ch, int mi, int mq)
{
#pragma unroll
    for (int k0 = 0; k0 < GC; k0 += 4) {
        const double a0 = As[(k0 + mq) * GLD + 16 * rb + mi];
        const double a = NEG ? -a0 : a0;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[(k0 + mq) * GLD + 16 * (2 * ch + t) + mi], acc[t], 0, 0, 0);
    }
}

template <int KIND>
__global__ void __launch_bounds__(512)
k_gemm_fwd(const SolveDesc *__restrict__ sd, int first, int c,
           const double *__restrict__ pool_all, const double *__restrict__ dinv_all, double *__restrict__ cv_all,
           double *__restrict__ X_all, double *__restrict__ gv_all, int nrhs, long long pool_stride, long long dinv_stride,
           long long cv_stride, long long x_stride, long long gv_stride, int batch)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    double *As = gsm, *Bs = gsm + GC * GLD;
    const SolveDesc d = sd[first + blockIdx.z / batch];
    const int b = blockIdx.z % batch;
    const int r = d.r, w = d.w;
    if (c * GC >= w) return;
    const int kb = c * GC, bw = min(GC, w - kb), ke = kb + bw;
    const int nsl = (r - ke + GC - 1) / GC;
    if ((int) blockIdx.x > 0 && (int) blockIdx.x >= nsl) return;
    const int n0 = blockIdx.y * GC, nlive = min(GC, nrhs - n0);
    const double *L = pool_all + (long long) b * pool_stride + d.lpan;
    const double *Linv = dinv_all + (long long) b * dinv_stride + d.dinv + (long long) c * 2 * GC * GC;
    double *V = gv_all + (long long) b * gv_stride + d.gv * (long long) nrhs;                 // V[row * nrhs + rhs]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, mi = lane & 15, mq = lane >> 4, rbk = wv & 3, ch = wv >> 2;
    // every global load of the workgroup goes out first (one round trip): the inverse, the chunk's rows of V, my
    // slice of the panel and my rows of V
    const bool has_rows = nsl > 0;
    const int row0 = ke + blockIdx.x * GC;
    double ra[8], rb[8], rl[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q, k = e >> 6, i = e & 63;
        ra[q] = Linv[e];
        rb[q] = load_if(V, (long long) (kb + k) * nrhs + n0 + i, k < bw && i < nlive);
        rl[q] = load_if(L, (long long) (row0 + i) + (long long) (kb + k) * r, has_rows && row0 + i < r && k < bw);
    }
    double4_t acc2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = row0 + 16 * rbk + mq + 4 * v, n = 16 * (2 * ch + t) + mi;
            acc2[t][v] = load_if(V, (long long) row * nrhs + n0 + n, has_rows && row < r && n < nlive);
        }
    // ---- Y = Linv_cc V_c
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q, k = e >> 6, i = e & 63;
        As[k * GLD + i] = ra[q];
        Bs[k * GLD + i] = rb[q];
    }
    __syncthreads();
    double4_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm64h<false>(As, Bs, acc, rbk, ch, mi, mq);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) Bs[(16 * rbk + mq + 4 * v) * GLD + 16 * (2 * ch + t) + mi] = acc[t][v];     // Y(k, n)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q, k = e >> 6, i = e & 63;
        As[k * GLD + i] = rl[q];                                                                     // L(row0 + i, kb + k)
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        double *X = X_all + (long long) b * x_stride;
        for (int e = tid; e < GC * GC; e += 512) {
            const int k = e >> 6, n = e & 63;
            if (k < bw && n < nlive) X[(long long) (d.c0 + kb + k) * nrhs + n0 + n] = Bs[k * GLD + n];
        }
    }
    if (!has_rows) return;
    // ---- my 64 rows below the chunk:  V_rows -= L(rows, chunk) Y
    gemm64h<true>(As, Bs, acc2, rbk, ch, mi, mq);
    const bool last = ke >= w && d.parent >= 0;             // rows below the pivots are final: the parent's input
    double *cv = cv_all + (long long) b * cv_stride;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = row0 + 16 * rbk + mq + 4 * v, n = 16 * (2 * ch + t) + mi;
            if (row < r && n < nlive) {
                V[(long long) row * nrhs + n0 + n] = acc2[t][v];
                if (last) cv[(d.cv + row - w) * nrhs + n0 + n] = acc2[t][v];
            }
        }
}

// V(pivot rows) = X(pivot rows) - U12 X(ancestors' rows): one workgroup per 64 pivot rows and tile of 64 right-hand sides
template <int KIND>
__global__ void __launch_bounds__(512)
k_gemm_bwd_init(const SolveDesc *__restrict__ sd, int first, const int *__restrict__ st_idx,
                const double *__restrict__ pool_all, const double *__restrict__ X_all, double *__restrict__ gv_all,
                int nrhs, long long pool_stride, long long x_stride, long long gv_stride, int batch)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    double *As = gsm, *Bs = gsm + GC * GLD;
    const SolveDesc d = sd[first + blockIdx.z / batch];
    const int b = blockIdx.z % batch;
    const int r = d.r, w = d.w, nb = r - w;
    const int i0 = blockIdx.x * GC;
    if (i0 >= w) return;
    const int n0 = blockIdx.y * GC, nlive = min(GC, nrhs - n0);
    const double *pool = pool_all + (long long) b * pool_stride;
    const double *X = X_all + (long long) b * x_stride;
    double *V = gv_all + (long long) b * gv_stride + d.gv * (long long) nrhs;
    const int *st = st_idx + d.st;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, mi = lane & 15, mq = lane >> 4, rbk = wv & 3, ch = wv >> 2;
    double4_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + 16 * rbk + mq + 4 * v, n = 16 * (2 * ch + t) + mi;
            acc[t][v] = load_if(X, (long long) (d.c0 + i) * nrhs + n0 + n, i < w && n < nlive);
        }
    for (int jb = 0; jb < nb; jb += GC) {
        // all 32 global loads of a thread go out together (the ancestors' row numbers one round trip ahead of their rows)
        int arow[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = (tid >> 6) + 8 * q;
            arow[q] = st[(jb + k < nb) ? w + jb + k : 0];
        }
        double ua[8], xb[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 512 * q;
            {   // As[k][i] = U(i0 + i, w + jb + k)
                const int k = (KIND == CS3_LU) ? e >> 6 : e & 63, i = (KIND == CS3_LU) ? e & 63 : e >> 6;
                const bool in = i0 + i < w && jb + k < nb;
                const long long off = (KIND == CS3_LU) ? d.upan + (long long) (i0 + i) * d.u_sk + (long long) (jb + k) * d.u_sj
                                                       : d.lpan + (long long) (w + jb + k) + (long long) (i0 + i) * r;
                ua[q] = load_if(pool, off, in);
            }
            {   // Bs[k][n] = X(row of ancestor w + jb + k, n0 + n)
                const int k = e >> 6, n = e & 63;
                xb[q] = load_if(X, (long long) arow[q] * nrhs + n0 + n, jb + k < nb && n < nlive);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 512 * q;
            const int k = (KIND == CS3_LU) ? e >> 6 : e & 63, i = (KIND == CS3_LU) ? e & 63 : e >> 6;
            As[k * GLD + i] = ua[q];
            Bs[(e >> 6) * GLD + (e & 63)] = xb[q];
        }
        __syncthreads();
        gemm64h<true>(As, Bs, acc, rbk, ch, mi, mq);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + 16 * rbk + mq + 4 * v, n = 16 * (2 * ch + t) + mi;
            if (i < w && n < nlive) V[(long long) i * nrhs + n0 + n] = acc[t][v];
        }
}

template <int KIND>
__global__ void __launch_bounds__(512)
k_gemm_bwd(const SolveDesc *__restrict__ sd, int first, int chunk_from_right,
           const double *__restrict__ pool_all, const double *__restrict__ dinv_all, double *__restrict__ X_all,
           double *__restrict__ gv_all, int nrhs, long long pool_stride, long long dinv_stride, long long x_stride,
           long long gv_stride, int batch, XMap xm)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    double *As = gsm, *Bs = gsm + GC * GLD;
    const SolveDesc d = sd[first + blockIdx.z / batch];
    const int b = blockIdx.z % batch;
    const int r = d.r, w = d.w;
    const int nchunk = (w + GC - 1) / GC;
    const int c = nchunk - 1 - chunk_from_right;
    if (c < 0) return;
    const int kb = c * GC, bw = min(GC, w - kb);
    const int nsl = kb / GC;                                    // 64-row blocks of pivot rows above the chunk
    if ((int) blockIdx.x > 0 && (int) blockIdx.x >= nsl) return;
    const int n0 = blockIdx.y * GC, nlive = min(GC, nrhs - n0);
    const double *L = pool_all + (long long) b * pool_stride + d.lpan;
    const double *Uinv = dinv_all + (long long) b * dinv_stride + d.dinv + (long long) c * 2 * GC * GC + GC * GC;
    double *V = gv_all + (long long) b * gv_stride + d.gv * (long long) nrhs;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, mi = lane & 15, mq = lane >> 4, rbk = wv & 3, ch = wv >> 2;
    const bool has_rows = nsl > 0;
    const int row0 = blockIdx.x * GC;
    double ra[8], rb[8], rl[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q;
        {
            const int k = e >> 6, i = e & 63;
            ra[q] = Uinv[e];
            rb[q] = load_if(V, (long long) (kb + k) * nrhs + n0 + i, k < bw && i < nlive);
        }
        {   // U(row0 + i, kb + k); Cholesky: L(kb + k, row0 + i), k contiguous in memory
            const int k = (KIND == CS3_LU) ? e >> 6 : e & 63, i = (KIND == CS3_LU) ? e & 63 : e >> 6;
            const long long off = (KIND == CS3_LU) ? (long long) (row0 + i) + (long long) (kb + k) * r
                                                   : (long long) (kb + k) + (long long) (row0 + i) * r;
            rl[q] = load_if(L, off, has_rows && row0 + i < kb && k < bw);
        }
    }
    double4_t acc2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = row0 + 16 * rbk + mq + 4 * v, n = 16 * (2 * ch + t) + mi;
            acc2[t][v] = load_if(V, (long long) row * nrhs + n0 + n, has_rows && row < kb && n < nlive);
        }
    // ---- Y = Uinv_cc V_c
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q, k = e >> 6, i = e & 63;
        As[k * GLD + i] = ra[q];
        Bs[k * GLD + i] = rb[q];
    }
    __syncthreads();
    double4_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm64h<false>(As, Bs, acc, rbk, ch, mi, mq);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) Bs[(16 * rbk + mq + 4 * v) * GLD + 16 * (2 * ch + t) + mi] = acc[t][v];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q;
        const int k = (KIND == CS3_LU) ? e >> 6 : e & 63, i = (KIND == CS3_LU) ? e & 63 : e >> 6;
        As[k * GLD + i] = rl[q];
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        double *X = X_all + (long long) b * x_stride;
        double *Xo = xm.dst ? xm.dst + (long long) b * x_stride : nullptr;     // fused permutation: row q[k] of the caller's array
        for (int e = tid; e < GC * GC; e += 512) {
            const int k = e >> 6, n = e & 63;
            if (k < bw && n < nlive) {
                X[(long long) (d.c0 + kb + k) * nrhs + n0 + n] = Bs[k * GLD + n];
                if (Xo) Xo[(long long) xm.q[d.c0 + kb + k] * nrhs + n0 + n] = Bs[k * GLD + n];
            }
        }
    }
    if (!has_rows) return;
    // ---- my 64 pivot rows above the chunk:  V_rows -= U(rows, chunk) Y
    gemm64h<true>(As, Bs, acc2, rbk, ch, mi, mq);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = row0 + 16 * rbk + mq + 4 * v, n = 16 * (2 * ch + t) + mi;
            if (row < kb && n < nlive) V[(long long) row * nrhs + n0 + n] = acc2[t][v];
        }
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

// Everything a (re)factorisation needs before its first front kernel, in ONE launch instead of four:
// the status word, zeros in the big-front buffers (the gather writes only touched entries), the copy of
// the caller's values, and -- for the fused factor + solve call -- the permuted right-hand sides.
// One flat index space, cut into the four jobs.
__global__ void __launch_bounds__(256)
k_prologue(int *status, double *__restrict__ pool, long long big_begin, long long nzero, long long pool_stride,
           long long batch, const double *__restrict__ ax_src, double *__restrict__ ax_dst, long long nax,
           const double *__restrict__ x_src, double *__restrict__ xp, const int *__restrict__ q, long long n, int nrhs,
           const int *__restrict__ f_src, double *__restrict__ axf, long long nf, long long nnz_a)
{
    const long long t0 = (long long) blockIdx.x * blockDim.x + threadIdx.x, stride = (long long) gridDim.x * blockDim.x;
    if (t0 == 0) { status[0] = 0x7f7f7f7f; status[1] = 0; status[2] = 0; }      // ... and the two hand-over words of the fused step
    const long long z_all = nzero * batch, x_all = x_src ? n * nrhs * batch : 0;
    // the values of the bottom forest's fronts in THEIR order (forest.hip reads them without an index indirection)
    for (long long t = t0; t < nf * batch; t += stride) axf[t] = ax_src[(t / nf) * nnz_a + f_src[t % nf]];
    for (long long t = t0; t < z_all + nax + x_all; t += stride) {
        if (t < z_all) {
            pool[(t / nzero) * pool_stride + big_begin + t % nzero] = 0.0;
        } else if (t < z_all + nax) {
            const long long e = t - z_all;
            ax_dst[e] = ax_src[e];
        } else {
            const long long e = t - z_all - nax, per = n * nrhs;
            const long long b = e / per, k = (e % per) / nrhs, c = e % nrhs;
            xp[e] = x_src[b * per + (long long) q[k] * nrhs + c];
        }
    }
}

// out[p] = map[p] < 0 ? 1.0 : value at the virtual pool offset map[p] of this matrix
__global__ void __launch_bounds__(256)
k_extract(const double *__restrict__ vals, const double *__restrict__ vals_il, long long il_len,
          const long long *__restrict__ map, double *__restrict__ out, long long count)
{
    for (long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x; p < count;
         p += (long long) gridDim.x * blockDim.x) {
        const long long o = map[p];
        out[p] = o < 0 ? 1.0 : (o < il_len ? vals_il[o * 64] : vals[o]);
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
        for (int p = Rp[i]; p < Rp[i + 1]; ++p) {
#pragma clang fp contract(off)                   // the reference rounds the product, then the sum
            const double prod = Rx[p] * X[(long long) Rj[p] * nrhs + t];
            y = y + prod;
        }
        Y[e] = y;
    }
}

// R = B - A X with A in the CSR-ordered view of the handle's CSC arrays (Rmap: CSR entry -> entry of Ax): the product
// row is summed as csc_mat_vec_ff does (ascending column, separate multiply and add roundings, csc_numba.py:309-328),
// then subtracted from b.  One thread per (row, right-hand side); fixed order, reproducible.
__global__ void __launch_bounds__(256)
k_residual_rows(const int *__restrict__ Rp, const int *__restrict__ Rj, const int *__restrict__ Rmap,
                const double *__restrict__ Ax_all, const double *__restrict__ X_all, const double *__restrict__ B_all,
                double *__restrict__ R_all, long long n, int nrhs, long long nnz_a)
{
    const double *Ax = Ax_all + (long long) blockIdx.y * nnz_a;
    const double *X = X_all + (long long) blockIdx.y * n * nrhs;
    const double *B = B_all ? B_all + (long long) blockIdx.y * n * nrhs : nullptr;      // null: R = A X (the SpMV alone)
    double *R = R_all + (long long) blockIdx.y * n * nrhs;
    const long long total = n * nrhs;
    for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long) gridDim.x * blockDim.x) {
        const long long i = e / nrhs;
        const int t = (int) (e - i * nrhs);
        double y = 0.0;
        for (int p = Rp[i]; p < Rp[i + 1]; ++p) {
#pragma clang fp contract(off)
            const double prod = Ax[Rmap[p]] * X[(long long) Rj[p] * nrhs + t];
            y = y + prod;
        }
        R[e] = B ? B[e] - y : y;
    }
}

// X += D; out[0] = max |D| over everything (atomicMax on the bit pattern of a non-negative double is exact)
__global__ void __launch_bounds__(256)
k_axpy_max(double *__restrict__ X, const double *__restrict__ D, long long total, unsigned long long *maxbits)
{
    double m = 0.0;
    for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long) gridDim.x * blockDim.x) {
        const double dv = D[e];
        if (X) X[e] += dv;
        const double a = fabs(dv);
        m = (a > m || a != a) ? a : m;
    }
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(m, off); m = (o > m || o != o) ? o : m; }
    if ((threadIdx.x & 63) == 0 && maxbits) atomicMax(maxbits, (unsigned long long) __double_as_longlong(m));
}

hipError_t launch_residual(const int *Rp, const int *Rj, const int *Rmap, const double *Ax, const double *X, const double *B,
                           double *R, long long n, int nrhs, long long nnz_a, long long batch, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned) std::min<long long>((n * nrhs + 255) / 256, 4096), (unsigned) batch);
    hipLaunchKernelGGL(k_residual_rows, grid, dim3(256), 0, st, Rp, Rj, Rmap, Ax, X, B, R, n, nrhs, nnz_a);
    return hipGetLastError();
}

hipError_t launch_axpy_max(double *X, const double *D, long long total, unsigned long long *maxbits, hipStream_t st)
{
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_axpy_max, dim3((unsigned) std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, st, X, D, total, maxbits);
    return hipGetLastError();
}

// 2 x 2 block stacking [[A, B], [C, D]] in CSC (power-flow Jacobian assembly), the layout of
// csc_stack_4_by_4_ff (csc_numba.py:640-720): output column j < an is A(:,j) followed by C(:,j)
// with rows shifted by am; column an + j is B(:,j) followed by D(:,j) shifted by bm.  Column
// pointers are closed-form sums of the inputs' pointers, so no scan is needed.  One wave per column.
__global__ void __launch_bounds__(256)
k_stack_4_by_4(int an, int bn, int am, int bm,
               const int *__restrict__ Ap, const int *__restrict__ Ai, const double *__restrict__ Ax,
               const int *__restrict__ Bp, const int *__restrict__ Bi, const double *__restrict__ Bx,
               const int *__restrict__ Cp, const int *__restrict__ Ci, const double *__restrict__ Cx,
               const int *__restrict__ Dp, const int *__restrict__ Di, const double *__restrict__ Dx,
               int *__restrict__ Pp, int *__restrict__ Pi, double *__restrict__ Px, int *__restrict__ map)
{
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (col >= an + bn) return;
    const bool left = col < an;
    const int j = left ? col : col - an;
    const int *Tp = left ? Ap : Bp, *Ti = left ? Ai : Bi;
    const int *Bp2 = left ? Cp : Dp, *Bi2 = left ? Ci : Di;
    const double *Tx = left ? Ax : Bx, *Bx2 = left ? Cx : Dx;
    const int shift = left ? am : bm;
    const int base = left ? 0 : Ap[an] + Cp[an];
    const int out0 = base + Tp[j] + Bp2[j];
    const int n1 = Tp[j + 1] - Tp[j], n2 = Bp2[j + 1] - Bp2[j];
    // position of the source entry in the concatenation A | B | C | D of the value arrays (the restack map)
    const int cat1 = left ? 0 : Ap[an], cat2 = left ? Ap[an] + Bp[bn] : Ap[an] + Bp[bn] + Cp[an];
    for (int k = lane; k < n1; k += 64) {
        Pi[out0 + k] = Ti[Tp[j] + k]; Px[out0 + k] = Tx[Tp[j] + k];
        if (map) map[out0 + k] = cat1 + Tp[j] + k;
    }
    for (int k = lane; k < n2; k += 64) {
        Pi[out0 + n1 + k] = Bi2[Bp2[j] + k] + shift; Px[out0 + n1 + k] = Bx2[Bp2[j] + k];
        if (map) map[out0 + n1 + k] = cat2 + Bp2[j] + k;
    }
    if (lane == 0) {
        Pp[col + 1] = out0 + n1 + n2;
        if (col == 0) Pp[0] = 0;
    }
}

hipError_t launch_stack_4_by_4(int an, int bn, int am, int bm, const int *Ap, const int *Ai, const double *Ax,
                               const int *Bp, const int *Bi, const double *Bx, const int *Cp, const int *Ci,
                               const double *Cx, const int *Dp, const int *Di, const double *Dx,
                               int *Pp, int *Pi, double *Px, int *map, hipStream_t st)
{
    const int ncol = an + bn;
    if (ncol == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stack_4_by_4, dim3((ncol + 3) / 4), dim3(256), 0, st, an, bn, am, bm, Ap, Ai, Ax, Bp, Bi, Bx,
                       Cp, Ci, Cx, Dp, Di, Dx, Pp, Pi, Px, map);
    hipError_t e = hipGetLastError();
    return e;
}

// Px[p] = value at position map[p] of the concatenation A | B | C | D: the Newton-loop restack (pattern unchanged)
__global__ void __launch_bounds__(256)
k_restack_values(long long nnz, const int *__restrict__ map, long long na, long long nb, long long nc,
                 const double *__restrict__ Ax, const double *__restrict__ Bx, const double *__restrict__ Cx,
                 const double *__restrict__ Dx, double *__restrict__ Px)
{
    for (long long p = (long long) blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (long long) gridDim.x * blockDim.x) {
        const long long s = map[p];
        Px[p] = (s < na) ? Ax[s] : (s < na + nb) ? Bx[s - na] : (s < na + nb + nc) ? Cx[s - na - nb] : Dx[s - na - nb - nc];
    }
}

hipError_t launch_restack_values(long long nnz, const int *map, long long na, long long nb, long long nc, const double *Ax,
                                 const double *Bx, const double *Cx, const double *Dx, double *Px, hipStream_t st)
{
    if (nnz == 0) return hipSuccess;
    hipLaunchKernelGGL(k_restack_values, dim3((unsigned) std::min<long long>((nnz + 255) / 256, 4096)), dim3(256), 0, st, nnz, map,
                       na, nb, nc, Ax, Bx, Cx, Dx, Px);
    return hipGetLastError();
}

// ================================================================ launchers ==
constexpr int RHS_LANES_MIN = 16;          // from this many right-hand sides on, SK_SMALL fronts run lane = right-hand side
                                           //   and the fronts of order > 64 sweep as GEMMs
#define CS3_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

static int grid_for(long long work, int block, int cap = 4096)
{
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int) g;
}

// A group of big fronts = one gather launch, then launches 0 .. nblk of the block step (nblk = closing).
static int big_group_blocks(const LaunchGroup &g) { return (g.max_w + BIG_NB - 1) / BIG_NB; }

static hipError_t launch_big_gather(const DeviceFactor &D, const LaunchGroup &g, hipStream_t st)
{
    const int gx = std::max(1, std::min(64, (g.max_r + 7) / 8));           // workgroups per front: eight columns each, at most 64
    hipLaunchKernelGGL(k_big_gather, dim3(gx, (unsigned) D.batch, g.count), dim3(256), 0, st, D.fdesc, g.first, D.kind,
                       AsmLists{D.fa_tgt, D.fa_src, D.ch_tab, D.rel_idx}, D.ax, D.pool_pm, D.nnz_a, D.pm_stride, IlView{D.pool_il, D.il_len});
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

template <int KIND>
static hipError_t launch_big_block(const DeviceFactor &D, const LaunchGroup &g, int blk, double inv_tol, hipStream_t st)
{
    const unsigned batch = (unsigned) D.batch;
    const int kb = blk * BIG_NB;
    // trailing region starts at ke >= kb + 1 for a front with a panel here, and at
    // w >= kb - BIG_NB + 1 for a front whose closing launch this is
    const int start = std::max(1, kb - BIG_NB + 1);
    const int rem = g.max_r - std::min(start, g.max_r);        // largest trailing order over the group
    const int tiles = 1 + (rem + 63) / 64;
    hipLaunchKernelGGL((k_big_step<KIND>), dim3(tiles, tiles, g.count * batch), dim3(256), 0, st, D.fdesc,
                       g.first, kb, D.pool_pm, D.pm_stride, D.dbuf, D.dbuf_size, inv_tol, D.status, (int) batch, D.tbuf);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

// One workgroup per (big front, matrix) when the batch fills the chip by itself and the panels fit the LDS.
static int wg_panel_ld(const LaunchGroup &g) { return (g.max_r + 1) | 1; }
// pivots per block step: 16 keeps the kernel within 128 registers, i.e. two workgroups per CU (its phases are serial and
// latency-bound: 512 matrices took 809 us with one workgroup per CU, in two rounds; the 32-pivot form of round 2a is gone)
constexpr int WG_NB = 16;
static int wg_nb() { return WG_NB; }
static size_t wg_lds_bytes(int kind, const LaunchGroup &g)
{
    // (Cholesky, left-looking: the block column + the block's own rows of the factor, 16 x max_w)
    return ((size_t) wg_nb() * 33 + (size_t) (kind == CS3_LU ? 2 : 1) * wg_nb() * wg_panel_ld(g) +
            (kind == CS3_LU ? 0 : (size_t) 16 * (size_t) g.max_w)) * sizeof(double);
}
bool big_group_in_one_workgroup(int kind, long long batch, const LaunchGroup &g)
{
    static const long long min_batch = getenv("CS3_WG_MIN_BATCH") ? atoll(getenv("CS3_WG_MIN_BATCH")) : 48;    // (32 matrices: 1.01 ms in one workgroup each, 0.95 block step by block step; 64: 1.21 / 1.22; 128: 1.56 / 1.70)
    return g.cls == FC_BIG && batch >= min_batch && wg_lds_bytes(kind, g) <= 150 * 1024;
}

template <int KIND>
static hipError_t launch_front_group(const DeviceFactor &D, const LaunchGroup &g, double inv_tol, hipStream_t st)
{
    const unsigned batch = (unsigned) D.batch;
    if (g.cls == FC_SUB) return launch_sub_factor(D, g.first, D.fwd_in_factor, inv_tol, st);     // a tier of the bottom forest
    if (big_group_in_one_workgroup(KIND, D.batch, g)) {
        hipLaunchKernelGGL((k_front_wg<KIND, WG_NB>), dim3((unsigned) g.count, batch), dim3(512), wg_lds_bytes(KIND, g), st, D.fdesc,
                           g.first, AsmLists{D.fa_tgt, D.fa_src, D.ch_tab, D.rel_idx}, D.ax, D.pool_pm, D.nnz_a, D.pm_stride,
                           IlView{D.pool_il, D.il_len}, inv_tol, D.status, wg_panel_ld(g), D.tbuf);
        CS3_LAUNCH_CHECK();
        return hipSuccess;
    }
    if (g.cls == FC_IL) {
        hipLaunchKernelGGL((k_front_il<KIND>), dim3((unsigned) g.count, (unsigned) D.ngroups), dim3(64), 0, st, D.fdesc, g.first,
                           D.ila_pairs, D.ax, D.nnz_a, IlView{D.pool_il, D.il_len}, D.pool_pm, D.pm_stride, (int) D.batch,
                           inv_tol, D.status);
        CS3_LAUNCH_CHECK();
        return hipSuccess;
    }
    if (g.cls == FC_BIG) {
        hipError_t e = launch_big_gather(D, g, st);
        // one launch per block of BIG_NB pivots, plus the closing launch (last update + last parked block)
        for (int blk = 0; e == hipSuccess && blk <= big_group_blocks(g); ++blk) e = launch_big_block<KIND>(D, g, blk, inv_tol, st);
        return e;
    }
    dim3 grid((unsigned) g.count, batch);
    const size_t ld = (size_t) (g.max_r | 1);
    // (front_wave_body, the path of every front with <= 32 pivots, packs the lower triangle for Cholesky; the 16 x 16 thread
    //  grid that takes the others keeps the full image)
    const bool packed = KIND == CS3_CHOLESKY && g.max_w <= 32;
    const size_t lds = ((packed ? (size_t) g.max_r * (size_t) (g.max_r + 1) / 2 : ld * (size_t) g.max_r) + 4 * (size_t) g.max_r + 6) * sizeof(double);
#define CS3_FRONT_ARGS D.fdesc, g.first, AsmLists{D.fa_tgt, D.fa_src, D.ch_tab, D.rel_idx}, D.ax, D.pool_pm, D.nnz_a, D.pm_stride, IlView{D.pool_il, D.il_len}, inv_tol, D.status, D.tbuf
    switch (g.cls) {
    case FC_R16:
    case FC_R32:      // only present when the analysis split the small fronts off (batched handles)
        hipLaunchKernelGGL((k_front_wave<KIND>), grid, dim3(64), lds, st, CS3_FRONT_ARGS); break;
    case FC_R64: {    // at most 32 pivots: panel by one wave + Schur complement by MFMA; more: the 16 x 16 thread grid
        // a batch fills the chip with fronts, not with waves per front: two waves (panel columns / pivot rows, then the
        // tiles) instead of four double the fronts per CU -- the image, not the registers, then bounds the occupancy
        const unsigned threads = (D.batch >= 16 && g.max_w <= 32) ? 128 : 256;
        hipLaunchKernelGGL((k_front_mix<KIND>), grid, dim3(threads), lds, st, CS3_FRONT_ARGS); break;
    }
    default:
        // 16 pivots per block step: the one-wave elimination of a block costs NBK^2 column updates, the MFMA update that
        // follows is cheap, so narrow blocks win (measured: 32 -> 16 took 4 % off the batched config, neutral on config 3;
        // 8 equal to 16)
        // the image: r x (r | 1) for LU, the packed lower triangle for Cholesky, one zero entry behind it
        const size_t blds = ((KIND == CS3_LU) ? ld * (size_t) g.max_r : (size_t) g.max_r * (size_t) (g.max_r + 1) / 2) * sizeof(double) + 16;
        // Cholesky has no block-row waves: four waves do all the eliminations of a block step, and a batch prefers more
        // workgroups per CU to more waves per front
        if (KIND == CS3_CHOLESKY && D.batch >= 16)
            hipLaunchKernelGGL((k_front_block<CS3_CHOLESKY, 16, 4>), grid, dim3(256), blds, st, CS3_FRONT_ARGS);
        else
            hipLaunchKernelGGL((k_front_block<KIND, 16>), grid, dim3(512), blds, st, CS3_FRONT_ARGS);
        break;
    }
#undef CS3_FRONT_ARGS
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

// Diagnostics: fill the LDS of every CU with NaN bit patterns (the LDS is not cleared between kernels), so that a kernel
// which multiplies a masked-to-zero operand by an LDS word nobody wrote shows up as NaN in the parity tests.
__global__ void __launch_bounds__(256) k_poison_lds(int words)
{
    extern __shared__ __attribute__((aligned(16))) double pz[];
    for (int i = threadIdx.x; i < words; i += 256) pz[i] = __longlong_as_double(0x7ff8dead0000beefll);
    __syncthreads();
    if (pz[(threadIdx.x * 97) % words] == 0.0) pz[0] = 1.0;      // (keeps the stores)
}

hipError_t launch_poison_lds(hipStream_t st)
{
    const int bytes = 160 * 1024;
    hipError_t e = hipFuncSetAttribute((const void *) k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(256), bytes, st, bytes / 8);      // four rounds over 256 CUs
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t prepare_kernels()
{
    // the largest LDS-resident class needs more than the default 64 KiB of dynamic LDS
    const int big = 160 * 1024;
    hipError_t e;
    const void *block_fns[] = {(const void *) k_front_block<CS3_LU, 16>, (const void *) k_front_block<CS3_CHOLESKY, 16>,
                               (const void *) k_front_block<CS3_CHOLESKY, 16, 4>,
                               (const void *) k_front_wg<CS3_LU, WG_NB>, (const void *) k_front_wg<CS3_CHOLESKY, WG_NB>};
    for (const void *f : block_fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        if (e != hipSuccess) return e;
    }
    const void *gemm_fns[] = {(const void *) k_gemm_fwd<CS3_LU>, (const void *) k_gemm_fwd<CS3_CHOLESKY>,
                              (const void *) k_gemm_bwd<CS3_LU>, (const void *) k_gemm_bwd<CS3_CHOLESKY>,
                              (const void *) k_gemm_bwd_init<CS3_LU>, (const void *) k_gemm_bwd_init<CS3_CHOLESKY>};
    for (const void *f : gemm_fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        if (e != hipSuccess) return e;
    }
    const void *solve_fns[] = {
        (const void *) k_fwd_blk<CS3_LU>, (const void *) k_fwd_blk<CS3_CHOLESKY>,
        (const void *) k_bwd_blk<CS3_LU>, (const void *) k_bwd_blk<CS3_CHOLESKY>,
        (const void *) k_bwd_big_init<CS3_LU>, (const void *) k_bwd_big_init<CS3_CHOLESKY>};
    for (const void *f : solve_fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t ForkJoin::init()
{
    for (int i = 0; i < NSIDE; ++i) {
        hipError_t e = hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    return hipStreamCreateWithFlags(&aux, hipStreamNonBlocking);
}

void ForkJoin::destroy()
{
    for (int i = 0; i < NSIDE; ++i) if (side[i]) { (void) hipStreamDestroy(side[i]); side[i] = nullptr; }
    if (aux) { (void) hipStreamDestroy(aux); aux = nullptr; }
    for (hipEvent_t e : events) (void) hipEventDestroy(e);
    events.clear();
    next = 0;
}

hipError_t ForkJoin::event(hipEvent_t *out)
{
    if (next == events.size()) {
        hipEvent_t e;
        hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (rc != hipSuccess) return rc;
        events.push_back(e);
    }
    *out = events[next++];
    return hipSuccess;
}

// Run the launch groups of one tree level concurrently: the first on `st`, the others on side
// streams that fork from `st` and join it again.  `launch(group, stream)` enqueues one group.
template <class Launch>
static hipError_t run_level(const std::vector<LaunchGroup> &groups, size_t g0, size_t g1, hipStream_t st,
                            ForkJoin &fj, bool parallel, Launch launch)
{
    const size_t ng = g1 - g0;
    if (ng <= 1) return ng ? launch(groups[g0], st) : hipSuccess;
    if (!parallel) {
        for (size_t i = g0; i < g1; ++i) { hipError_t e = launch(groups[i], st); if (e != hipSuccess) return e; }
        return hipSuccess;
    }
    hipEvent_t fork;
    hipError_t e = fj.event(&fork);
    if (e != hipSuccess) return e;
    if ((e = hipEventRecord(fork, st)) != hipSuccess) return e;
    // the heaviest group (last: big fronts / block kernels sort last) stays on the main stream and is captured FIRST: the
    // runtime keeps a node's first captured dependent in the node's own hardware queue (launch_factor_with_forward)
    if ((e = launch(groups[g1 - 1], st)) != hipSuccess) return e;
    for (size_t i = 0; i + 1 < ng; ++i) {
        hipStream_t s = fj.side[i % ForkJoin::NSIDE];
        if (i < (size_t) ForkJoin::NSIDE) { if ((e = hipStreamWaitEvent(s, fork, 0)) != hipSuccess) return e; }
        if ((e = launch(groups[g0 + i], s)) != hipSuccess) return e;
    }
    for (size_t i = 0; i < std::min(ng - 1, (size_t) ForkJoin::NSIDE); ++i) {
        hipEvent_t j;
        if ((e = fj.event(&j)) != hipSuccess) return e;
        if ((e = hipEventRecord(j, fj.side[i])) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st, j, 0)) != hipSuccess) return e;
    }
    return hipSuccess;
}

static std::vector<LaunchGroup> factor_groups(const DeviceFactor &D, const std::vector<LaunchGroup> &groups);

hipError_t launch_factor_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &all_groups,
                                double inv_tol, hipStream_t st, ForkJoin &fj)
{
    const std::vector<LaunchGroup> groups = factor_groups(D, all_groups);
    fj.rewind();                         // (the big-front buffers were zeroed by launch_prologue)
    for (size_t g0 = 0; g0 < groups.size(); ) {
        size_t g1 = g0;
        while (g1 < groups.size() && groups[g1].level == groups[g0].level) ++g1;
        hipError_t e = run_level(groups, g0, g1, st, fj, true, [&](const LaunchGroup &g, hipStream_t s) {
            return (D.kind == CS3_LU) ? launch_front_group<CS3_LU>(D, g, inv_tol, s)
                                      : launch_front_group<CS3_CHOLESKY>(D, g, inv_tol, s);
        });
        if (e != hipSuccess) return e;
        g0 = g1;
    }
    return hipSuccess;
}

// Sweeps over a group of wide big fronts: a preparation launch, then one launch per chunk of columns.
struct BigSweepPlan {
    bool wide, multi;       // two 64-blocks per launch (latency-bound); BIG_KT right-hand sides per workgroup
    int cw, nchunk, slices;
    unsigned by, by_multi;
};

static BigSweepPlan big_sweep_plan(const DeviceFactor &D, const LaunchGroup &g, int nrhs)
{
    BigSweepPlan pl;
    pl.by = (unsigned) (D.batch * nrhs);
    pl.wide = (long long) D.batch * nrhs < 8;
    pl.multi = nrhs >= BIG_KT;
    pl.by_multi = (unsigned) (D.batch * ((nrhs + BIG_KT - 1) / BIG_KT));
    pl.cw = pl.wide ? BIG_CW : SOLVE_BW;
    pl.nchunk = (g.max_w + pl.cw - 1) / pl.cw;
    pl.slices = std::max(1, (g.max_r + 63) / 64);
    return pl;
}

static hipError_t launch_fwd_big_pre(const DeviceFactor &D, const LaunchGroup &g, double *X, int nrhs, hipStream_t st)
{
    const BigSweepPlan pl = big_sweep_plan(D, g, nrhs);
    hipLaunchKernelGGL(k_fwd_big_gather, dim3(1, pl.by, g.count), dim3(256), 0, st, D.sd(), g.first, D.fasm_src,
                       D.fasm_tgt, D.flong_src, D.cv, X, D.bigv, nrhs, D.cv_size * (long long) nrhs, D.n * (long long) nrhs, D.bv_size);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

template <int KIND>
static hipError_t launch_fwd_big_chunk(const DeviceFactor &D, const LaunchGroup &g, double *X, int nrhs, int c, hipStream_t st)
{
    const BigSweepPlan pl = big_sweep_plan(D, g, nrhs);
    const long long xs = D.n * (long long) nrhs, cvs = D.cv_size * (long long) nrhs;
    if (pl.wide)
        hipLaunchKernelGGL((k_fwd_big_step<KIND, BIG_CW>), dim3(pl.slices, pl.by, g.count), dim3(256), 0, st, D.sd(),
                           g.first, c * pl.cw, D.pool_pm, D.cv, X, D.bigv, nrhs, D.pm_stride, cvs, xs, D.bv_size);
    else if (pl.multi)
        hipLaunchKernelGGL((k_fwd_big_step_multi<KIND>), dim3(pl.slices, pl.by_multi, g.count), dim3(256), 0, st, D.sd(),
                           g.first, c * pl.cw, D.pool_pm, D.cv, X, D.bigv, nrhs, D.pm_stride, cvs, xs, D.bv_size);
    else
        hipLaunchKernelGGL((k_fwd_big_step<KIND, SOLVE_BW>), dim3(pl.slices, pl.by, g.count), dim3(256), 0, st, D.sd(),
                           g.first, c * pl.cw, D.pool_pm, D.cv, X, D.bigv, nrhs, D.pm_stride, cvs, xs, D.bv_size);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

template <int KIND>
static hipError_t launch_bwd_big_pre(const DeviceFactor &D, const LaunchGroup &g, double *X, int nrhs, hipStream_t st)
{
    const BigSweepPlan pl = big_sweep_plan(D, g, nrhs);
    const size_t lds = (size_t) std::max(1, g.max_r) * sizeof(double);
    hipLaunchKernelGGL((k_bwd_big_init<KIND>), dim3(2, pl.by, g.count), dim3(256), lds, st, D.sd(), g.first,
                       D.st_idx, D.pool_pm, X, D.bigv, nrhs, D.pm_stride, D.n * (long long) nrhs, D.bv_size);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

template <int KIND>
static hipError_t launch_bwd_big_chunk(const DeviceFactor &D, const LaunchGroup &g, double *X, int nrhs, int c, hipStream_t st)
{
    const BigSweepPlan pl = big_sweep_plan(D, g, nrhs);
    const long long xs = D.n * (long long) nrhs;
    if (pl.wide)
        hipLaunchKernelGGL((k_bwd_big_step<KIND, BIG_CW>), dim3(pl.slices, pl.by, g.count), dim3(256), 0, st, D.sd(),
                           g.first, c, D.pool_pm, X, D.bigv, nrhs, D.pm_stride, xs, D.bv_size);
    else if (pl.multi)
        hipLaunchKernelGGL((k_bwd_big_step_multi<KIND>), dim3(pl.slices, pl.by_multi, g.count), dim3(256), 0, st, D.sd(),
                           g.first, c, D.pool_pm, X, D.bigv, nrhs, D.pm_stride, xs, D.bv_size);
    else
        hipLaunchKernelGGL((k_bwd_big_step<KIND, SOLVE_BW>), dim3(pl.slices, pl.by, g.count), dim3(256), 0, st, D.sd(),
                           g.first, c, D.pool_pm, X, D.bigv, nrhs, D.pm_stride, xs, D.bv_size);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}


// GEMM sweeps of a group of fronts of order > 64 (SK_WAVE, SK_BLOCK, SK_BIG) for many right-hand sides.
// Inverted 64 x 64 diagonal blocks of the fronts of order > 64: tasks [t0, t1) of the list (solve-schedule order).
template <int KIND>
static hipError_t launch_inv_tasks(const DeviceFactor &D, int t0, int t1, hipStream_t st)
{
    if (t1 <= t0) return hipSuccess;
    hipLaunchKernelGGL((k_inv_diag<KIND>), dim3((unsigned) (t1 - t0), (unsigned) D.batch), dim3(256), 0, st, D.sdesc, D.inv_tasks, t0,
                       D.pool_pm, D.pm_stride, D.dinv, D.dinv_size);
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

// The permutations can ride on the sweeps when every pivot row of X is read (forward) and written (backward) by a kernel
// that knows the row map: the lane = right-hand-side kernels and the GEMM sweeps, i.e. 16 or more right-hand sides, no
// interleaved batch, GEMM sweeps on.
bool permutation_can_fuse(const DeviceFactor &D, int nrhs)
{
    static const bool on = !(getenv("CS3_NO_GEMM_SWEEPS") && getenv("CS3_NO_GEMM_SWEEPS")[0] == '1');
    // (measured on config 4: the extra row-map round trip per front costs more than the two permutation kernels below a
    // few hundred right-hand sides; at 1024 the fused form saves 0.14 ms of 2.3)
    constexpr int min_rhs = 256;
    return on && nrhs >= std::max(min_rhs, RHS_LANES_MIN) && D.il_len == 0;
}

hipError_t launch_diag_inverses(const DeviceFactor &D, hipStream_t st)
{
    return (D.kind == CS3_LU) ? launch_inv_tasks<CS3_LU>(D, 0, D.n_inv_tasks, st) : launch_inv_tasks<CS3_CHOLESKY>(D, 0, D.n_inv_tasks, st);
}

// with_inverse: compute this group's inverted diagonal blocks first (the fused factor + solve graph, where they cannot
// be prepared ahead of the factorisation); otherwise the caller has run launch_diag_inverses since the last factorisation.
template <int KIND>
static hipError_t launch_gemm_group(const DeviceFactor &D, const LaunchGroup &g, double *X, int nrhs, bool forward, bool with_inverse,
                                    hipStream_t st)
{
    const long long xs = D.n * (long long) nrhs, cvs = D.cv_size * (long long) nrhs, gvs = D.gv_size * (long long) nrhs;
    const unsigned batch = (unsigned) D.batch, tiles = (unsigned) ((nrhs + GC - 1) / GC);
    // the inverted diagonal blocks of this group's fronts: tasks are in solve-schedule order
    int t0 = 0, t1 = 0;
    {
        const std::vector<int> &T = D.inv_tasks_host;
        const int nt = (int) (T.size() / 2);
        while (t0 < nt && T[2 * t0] < g.first) ++t0;
        t1 = t0;
        while (t1 < nt && T[2 * t1] < g.first + g.count) ++t1;
    }
    if (with_inverse) {
        hipError_t ie = launch_inv_tasks<KIND>(D, t0, t1, st);
        if (ie != hipSuccess) return ie;
    }
    const int nchunk = (g.max_w + GC - 1) / GC;
    const unsigned slices = (unsigned) std::max(1, (g.max_r + GC - 1) / GC);
    if (forward) {
        hipLaunchKernelGGL(k_gemm_gather, dim3((unsigned) ((g.max_r + 15) / 16), tiles, g.count * batch), dim3(256), 0, st, D.sd(),
                           g.first, D.sl_src, D.cv, X, D.gv, nrhs, cvs, xs, gvs, D.xm, (int) batch);
        CS3_LAUNCH_CHECK();
        for (int c = 0; c < nchunk; ++c) {
            hipLaunchKernelGGL((k_gemm_fwd<KIND>), dim3(slices, tiles, g.count * batch), dim3(512), GEMM_LDS, st, D.sd(), g.first, c,
                               D.pool_pm, D.dinv, D.cv, X, D.gv, nrhs, D.pm_stride, D.dinv_size, cvs, xs, gvs, (int) batch);
            CS3_LAUNCH_CHECK();
        }
    } else {
        hipLaunchKernelGGL((k_gemm_bwd_init<KIND>), dim3((unsigned) nchunk, tiles, g.count * batch), dim3(512), GEMM_LDS, st, D.sd(),
                           g.first, D.st_idx, D.pool_pm, X, D.gv, nrhs, D.pm_stride, xs, gvs, (int) batch);
        CS3_LAUNCH_CHECK();
        for (int c = 0; c < nchunk; ++c) {
            hipLaunchKernelGGL((k_gemm_bwd<KIND>), dim3((unsigned) std::max(1, nchunk), tiles, g.count * batch), dim3(512), GEMM_LDS, st,
                               D.sd(), g.first, c, D.pool_pm, D.dinv, X, D.gv, nrhs, D.pm_stride, D.dinv_size, xs, gvs, (int) batch, D.xm);
            CS3_LAUNCH_CHECK();
        }
    }
    return hipSuccess;
}

template <int KIND, int RMAX>
static void launch_rhs_sweep(const DeviceFactor &D, int first, int count, double *X, int nrhs, bool forward, hipStream_t st)
{
    if (count <= 0) return;
    const long long xs = D.n * (long long) nrhs, cvs = D.cv_size * (long long) nrhs;
    dim3 grid((unsigned) count, (unsigned) D.batch, (unsigned) ((nrhs + 63) / 64));
    if (forward)
        hipLaunchKernelGGL((k_fwd_rhs<KIND, RMAX>), grid, dim3(64), 0, st, D.sd(), first, D.sl_src, D.pool_pm, D.cv, X,
                           nrhs, D.pm_stride, cvs, xs, D.xm);
    else
        hipLaunchKernelGGL((k_bwd_rhs<KIND, RMAX>), grid, dim3(64), 0, st, D.sd(), first, D.st_idx, D.pool_pm, X,
                           nrhs, D.pm_stride, xs, D.xm);
}

template <int KIND>
static hipError_t launch_solve_group(const DeviceFactor &D, const LaunchGroup &g, double *X, int nrhs,
                                     bool forward, hipStream_t st)
{
    const long long xs = D.n * (long long) nrhs;
    const long long cvs = D.cv_size * (long long) nrhs;
    if (g.cls == SK_SUB) return (nrhs == 1) ? launch_sub_sweep(D, g.first, X, forward, st) : hipErrorInvalidValue;
    static const bool use_gemm = !(getenv("CS3_NO_GEMM_SWEEPS") && getenv("CS3_NO_GEMM_SWEEPS")[0] == '1');
    if (use_gemm && nrhs >= RHS_LANES_MIN && (g.cls == SK_WAVE || g.cls == SK_BLOCK || g.cls == SK_BIG))
        return launch_gemm_group<KIND>(D, g, X, nrhs, forward, forward && D.inverses_in_sweep, st);
    if (g.cls == SK_IL) {
        // the fronts of order <= 16 come first in the group: half the LDS per wave (the front vector of 64 matrices),
        // twice the waves per CU -- these sweeps wait on one round trip per pivot
        const int n16 = g.n16;
        if (n16 > 0) {
            dim3 grid((unsigned) n16, (unsigned) D.ngroups);
            if (forward)
                hipLaunchKernelGGL((k_fwd_il<KIND, 16>), grid, dim3(64), 0, st, D.sd(), g.first, D.rl_pairs,
                                   IlView{D.pool_il, D.il_len}, D.cv, X, nrhs, cvs, xs, (int) D.batch);
            else
                hipLaunchKernelGGL((k_bwd_il<KIND, 16>), grid, dim3(64), 0, st, D.sd(), g.first, D.st_idx,
                                   IlView{D.pool_il, D.il_len}, X, nrhs, xs, (int) D.batch);
        }
        if (g.count > n16) {
            dim3 grid((unsigned) (g.count - n16), (unsigned) D.ngroups);
            if (forward)
                hipLaunchKernelGGL((k_fwd_il<KIND, IL_RMAX>), grid, dim3(64), 0, st, D.sd(), g.first + n16, D.rl_pairs,
                                   IlView{D.pool_il, D.il_len}, D.cv, X, nrhs, cvs, xs, (int) D.batch);
            else
                hipLaunchKernelGGL((k_bwd_il<KIND, IL_RMAX>), grid, dim3(64), 0, st, D.sd(), g.first + n16, D.st_idx,
                                   IlView{D.pool_il, D.il_len}, X, nrhs, xs, (int) D.batch);
        }
    } else if (g.cls == SK_SMALL && nrhs >= RHS_LANES_MIN) {
        // one instance by the group's largest order (fronts of order <= 32 only: the analysis sends the others to the GEMM
        // sweeps; a separate launch for the fronts of order <= 16 of a group paid while the assembly went through LDS and
        // stopped paying with the slot rounds: 256 right-hand sides 1.01 ms split, 0.945 not)
        if (g.max_r <= 16) launch_rhs_sweep<KIND, 16>(D, g.first, g.count, X, nrhs, forward, st);
        else if (g.max_r <= 24) launch_rhs_sweep<KIND, 24>(D, g.first, g.count, X, nrhs, forward, st);
        else if (g.max_r <= 32) launch_rhs_sweep<KIND, 32>(D, g.first, g.count, X, nrhs, forward, st);
        else return hipErrorInvalidValue;
    } else if (g.cls == SK_SMALL || g.cls == SK_WAVE) {
        if (nrhs == 1) {
            dim3 grid((unsigned) ((g.count + 3) / 4), (unsigned) D.batch, 1);
            if (forward)
                hipLaunchKernelGGL((k_fwd_wave<KIND, 1>), grid, dim3(256), 0, st, D.sd(), g.first, g.count, D.fasm_src,
                                   D.fasm_tgt, D.flong_src, D.pool_pm, D.cv, X, nrhs, D.pm_stride, cvs, xs);
            else
                hipLaunchKernelGGL((k_bwd_wave<KIND, 1>), grid, dim3(256), 0, st, D.sd(), g.first, g.count, D.st_idx,
                                   D.pool_pm, X, nrhs, D.pm_stride, xs);
        } else {
            constexpr int KT = 8;                   // right-hand sides per wave: the panel is read once per tile
            dim3 grid((unsigned) ((g.count + 3) / 4), (unsigned) D.batch, (unsigned) ((nrhs + KT - 1) / KT));
            if (forward)
                hipLaunchKernelGGL((k_fwd_wave<KIND, KT>), grid, dim3(256), 0, st, D.sd(), g.first, g.count, D.fasm_src,
                                   D.fasm_tgt, D.flong_src, D.pool_pm, D.cv, X, nrhs, D.pm_stride, cvs, xs);
            else
                hipLaunchKernelGGL((k_bwd_wave<KIND, KT>), grid, dim3(256), 0, st, D.sd(), g.first, g.count, D.st_idx,
                                   D.pool_pm, X, nrhs, D.pm_stride, xs);
        }
    } else if (g.cls == SK_BIG) {
        const BigSweepPlan pl = big_sweep_plan(D, g, nrhs);
        hipError_t e = forward ? launch_fwd_big_pre(D, g, X, nrhs, st) : launch_bwd_big_pre<KIND>(D, g, X, nrhs, st);
        for (int c = 0; e == hipSuccess && c < pl.nchunk; ++c)
            e = forward ? launch_fwd_big_chunk<KIND>(D, g, X, nrhs, c, st) : launch_bwd_big_chunk<KIND>(D, g, X, nrhs, c, st);
        return e;
    } else {
        dim3 grid((unsigned) g.count, (unsigned) D.batch, (unsigned) nrhs);
        const size_t lds = (size_t) (g.max_r + 1 + SOLVE_BW) * sizeof(double);
        if (forward)
            hipLaunchKernelGGL((k_fwd_blk<KIND>), grid, dim3(256), lds, st, D.sd(), g.first, D.fasm_src,
                               D.fasm_tgt, D.flong_src, D.pool_pm, D.cv, X, nrhs, D.pm_stride, cvs, xs);
        else
            hipLaunchKernelGGL((k_bwd_blk<KIND>), grid, dim3(256), lds, st, D.sd(), g.first, D.st_idx,
                               D.pool_pm, X, nrhs, D.pm_stride, xs);
    }
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

// With few right-hand sides SK_SMALL and SK_WAVE fronts share the one-wave-per-front kernels:
// the two groups of a level are adjacent in the schedule and go out as one launch.
// A handful of small fronts beside the level's workgroup-per-front group (the top of the tree) join that group: a second
// launch for four fronts costs a dependent launch (sweeps, 11 us) or a fork and a join across hardware queues
// (factorisation, 13 us), the workgroup kernels take fronts of any order.
static int promote_max()
{
    return 16;                                   // (8 .. 200 measured identical on config 3: only the top three levels hold such a group)
}

static void absorb_small_group(std::vector<LaunchGroup> &out, const LaunchGroup &g, int small_cls, int wg_cls)
{
    if (!out.empty() && out.back().level == g.level && out.back().cls == small_cls && g.cls == wg_cls &&
        out.back().count <= promote_max() && out.back().first + out.back().count == g.first) {
        LaunchGroup &m = out.back();
        m.cls = wg_cls; m.count += g.count;
        m.max_r = std::max(m.max_r, g.max_r); m.max_w = std::max(m.max_w, g.max_w);
        m.n16 = 0;
    } else {
        out.push_back(g);
    }
}

static std::vector<LaunchGroup> sweep_groups(const std::vector<LaunchGroup> &groups, int nrhs, long long batch)
{
    if (nrhs >= RHS_LANES_MIN) return groups;
    std::vector<LaunchGroup> merged, out;
    for (const LaunchGroup &g : groups) {
        if (!merged.empty() && merged.back().level == g.level && merged.back().cls == SK_SMALL && g.cls == SK_WAVE &&
            merged.back().first + merged.back().count == g.first) {
            LaunchGroup &m = merged.back();
            m.cls = SK_WAVE; m.count += g.count;
            m.max_r = std::max(m.max_r, g.max_r); m.max_w = std::max(m.max_w, g.max_w);
        } else {
            merged.push_back(g);
        }
    }
    if (batch != 1) return merged;       // (a batch fills the chip with waves: one per front stays cheaper)
    for (const LaunchGroup &g : merged) absorb_small_group(out, g, SK_WAVE, SK_BLOCK);
    return out;
}

// the same for the factorisation of a single matrix: few fronts of order <= 64 beside the level's LDS-image fronts
static std::vector<LaunchGroup> factor_groups(const DeviceFactor &D, const std::vector<LaunchGroup> &groups)
{
    if (D.batch != 1) return groups;
    std::vector<LaunchGroup> out;
    for (const LaunchGroup &g : groups) absorb_small_group(out, g, FC_R64, FC_LDS);
    return out;
}

hipError_t launch_solve_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &all_groups,
                               double *X, int nrhs, bool forward, hipStream_t st, ForkJoin &fj)
{
    const std::vector<LaunchGroup> groups = sweep_groups(all_groups, nrhs, D.batch);
    fj.rewind();
    // one right-hand side: the per-level launches are short, fork/join costs more than it hides (measured), so they stay
    // in line.  Many right-hand sides: the lane = right-hand-side group and the GEMM group of a level take 10-40 us each
    // and are independent -- side by side they save 60 us of 1.73 ms at 1024 right-hand sides, 24 us of 0.83 at 128.
    // (re-measured with the fronts of order 33-64 on the GEMM sweeps: the fork now pays from 512 right-hand sides on only
    //  -- 128: 0.82 ms forked, 0.77 in line; 256: equal; 1024: equal to 0.5 %)
    const bool solve_parallel = nrhs >= 512;
    auto launch = [&](const LaunchGroup &g, hipStream_t s) {
        return (D.kind == CS3_LU) ? launch_solve_group<CS3_LU>(D, g, X, nrhs, forward, s)
                                  : launch_solve_group<CS3_CHOLESKY>(D, g, X, nrhs, forward, s);
    };
    if (forward) {
        for (size_t g0 = 0; g0 < groups.size(); ) {
            size_t g1 = g0;
            while (g1 < groups.size() && groups[g1].level == groups[g0].level) ++g1;
            hipError_t e = run_level(groups, g0, g1, st, fj, solve_parallel, launch);
            if (e != hipSuccess) return e;
            g0 = g1;
        }
    } else {
        for (size_t g1 = groups.size(); g1 > 0; ) {
            size_t g0 = g1;
            while (g0 > 0 && groups[g0 - 1].level == groups[g1 - 1].level) --g0;
            hipError_t e = run_level(groups, g0, g1, st, fj, solve_parallel, launch);
            if (e != hipSuccess) return e;
            g1 = g0;
        }
    }
    return hipSuccess;
}

// Rough cost of a launch group in dependent-launch units, to place the fork below.
static int factor_group_cost(const LaunchGroup &g)
{
    return (g.cls == FC_BIG) ? 4 * ((g.max_w + BIG_NB - 1) / BIG_NB + 2) : 2;
}
static int sweep_group_cost(const LaunchGroup &g)
{
    return (g.cls == SK_BIG) ? 2 + (g.max_w + BIG_CW - 1) / BIG_CW : 1;
}

// Factorisation with the forward sweep partly hidden behind it.  The sweep of levels 0..K needs only
// the panels of those levels, so once level K is factorised it runs on fj.aux beside the
// factorisation of levels K+1.. (the tail of the tree: few, large fronts, many dependent launches).
// K is the last level whose tail is still long enough to cover the sweep; one fork, one join -- a
// fork per level costs more than it hides (measured).  Same kernels, same operands: same bits.
hipError_t launch_factor_with_forward(const DeviceFactor &D, const std::vector<LaunchGroup> &all_fgroups,
                                      const std::vector<LaunchGroup> &all_sgroups, double inv_tol, double *X, int nrhs,
                                      hipStream_t st, ForkJoin &fj)
{
    hipError_t e;
    if (!D.sub_tiers.empty() && !D.sd_active) {
        // a bottom forest under the factorisation, but sweeps on the level schedule of the whole tree (several right-hand
        // sides): the two number their levels differently, so nothing is overlapped
        if ((e = launch_factor_levels(D, all_fgroups, inv_tol, st, fj)) != hipSuccess) return e;
        return launch_solve_levels(D, all_sgroups, X, nrhs, true, st, fj);
    }
    const std::vector<LaunchGroup> fgroups = factor_groups(D, all_fgroups);
    const std::vector<LaunchGroup> sgroups = sweep_groups(all_sgroups, nrhs, D.batch);
    fj.rewind();
    const int nlevels = fgroups.empty() ? 0 : fgroups.back().level + 1;
    std::vector<long long> tail(nlevels + 1, 0), head(nlevels + 1, 0);   // factor cost of levels >= l; sweep cost of levels < l
    for (const LaunchGroup &g : fgroups) tail[g.level] += factor_group_cost(g);
    for (int l = nlevels - 1; l >= 0; --l) tail[l] += tail[l + 1];
    // (a tier of the bottom forest whose factor launch carries the forward sweep has nothing left to sweep)
    for (const LaunchGroup &g : sgroups) head[g.level + 1] += (g.cls == SK_SUB && D.fwd_in_factor) ? 0 : sweep_group_cost(g);
    for (int l = 0; l < nlevels; ++l) head[l + 1] += head[l];
    int fork_level = -1;
    static const bool overlap = !(getenv("CS3_NO_OVERLAP") && getenv("CS3_NO_OVERLAP")[0] == '1');
    for (int l = 0; overlap && l + 1 < nlevels; ++l)
        if (tail[l + 1] >= head[l + 1]) fork_level = l;

    auto sweep = [&](int lo, int hi, hipStream_t s) -> hipError_t {      // forward sweep of levels lo..hi
        for (const LaunchGroup &g : sgroups) {
            if (g.level < lo || g.level > hi) continue;
            if (g.cls == SK_SUB && D.fwd_in_factor) continue;
            hipError_t se = (D.kind == CS3_LU) ? launch_solve_group<CS3_LU>(D, g, X, nrhs, true, s)
                                               : launch_solve_group<CS3_CHOLESKY>(D, g, X, nrhs, true, s);
            if (se != hipSuccess) return se;
        }
        return hipSuccess;
    };
    // BATCHES: every level is a launch of 100 us or more, so a cross-queue edge (5-10 us) per level is cheap and the
    // forward sweep follows the factorisation level by level on fj.aux -- sweep level l as soon as level l is factorised --
    // instead of waiting for the fork level (round 3; kernel trace of 512 matrices: the sweep of levels 1..9, 400 us of
    // small launches, ran after the root with nothing beside it).  Sweep level l is CAPTURED after factor level l + 1, so
    // that the factor chain stays the first captured dependent of its own nodes (see below).
    if (overlap && D.batch >= 16 && nlevels >= 2) {
        hipEvent_t pending = nullptr;
        int pending_level = -1;
        auto flush = [&]() -> hipError_t {
            if (!pending) return hipSuccess;
            hipError_t fe;
            if ((fe = hipStreamWaitEvent(fj.aux, pending, 0)) != hipSuccess) return fe;
            fe = sweep(pending_level, pending_level, fj.aux);
            pending = nullptr;
            return fe;
        };
        for (size_t f0 = 0; f0 < fgroups.size(); ) {
            const int level = fgroups[f0].level;
            size_t f1 = f0;
            while (f1 < fgroups.size() && fgroups[f1].level == level) ++f1;
            e = run_level(fgroups, f0, f1, st, fj, true, [&](const LaunchGroup &g, hipStream_t s2) {
                return (D.kind == CS3_LU) ? launch_front_group<CS3_LU>(D, g, inv_tol, s2)
                                          : launch_front_group<CS3_CHOLESKY>(D, g, inv_tol, s2);
            });
            if (e != hipSuccess) return e;
            if ((e = flush()) != hipSuccess) return e;
            if ((e = fj.event(&pending)) != hipSuccess) return e;
            if ((e = hipEventRecord(pending, st)) != hipSuccess) return e;
            pending_level = level;
            f0 = f1;
        }
        if ((e = flush()) != hipSuccess) return e;
        hipEvent_t done;
        if ((e = fj.event(&done)) != hipSuccess) return e;
        if ((e = hipEventRecord(done, fj.aux)) != hipSuccess) return e;
        return hipStreamWaitEvent(st, done, 0);
    }
    // The last level, when it is one group of wide big fronts (the dense root): its forward sweep does not wait
    // for the end of its factorisation either.  Column block b of the factor is final in F after block launch
    // b + 1 (k_big_step), so chunk c of the sweep follows on fj.aux as soon as its blocks are home, while the
    // later blocks are still being factorised.
    const LaunchGroup *rootf = nullptr, *roots = nullptr;
    // (with many right-hand sides the root sweeps as GEMMs through launch_solve_group like every other group)
    if (nrhs < RHS_LANES_MIN && nlevels >= 2 && fork_level == nlevels - 2) {
        int nf = 0, ns = 0;
        for (const LaunchGroup &g : fgroups) if (g.level == nlevels - 1) { ++nf; rootf = &g; }
        for (const LaunchGroup &g : sgroups) if (g.level == nlevels - 1) { ++ns; roots = &g; }
        if (nf != 1 || ns != 1 || rootf->cls != FC_BIG || roots->cls != SK_BIG || rootf->count != roots->count)
            rootf = roots = nullptr;
    }
    // The side branch is CAPTURED after the next launch of the chain it forks from: the runtime keeps a node's FIRST
    // captured dependent in the node's own hardware queue and moves the others to another queue, whose first dispatch
    // then waits 40-60 us (measured: profiles/r02_timeline_fused_step.json) -- the late start must land on the sweep,
    // which has slack, not on the factorisation.  (Tried and removed, DESIGN.md: the sweep captured first; the block chain
    // waiting for the side branch before the release; hand-overs through memory words with a waiting kernel.)
    hipEvent_t swept = nullptr, ready_deferred = nullptr;
    int root_rest = 0;                             // first chunk of the root's sweep that is still to do after the join
    for (size_t f0 = 0; f0 < fgroups.size(); ) {
        const int level = fgroups[f0].level;
        size_t f1 = f0;
        while (f1 < fgroups.size() && fgroups[f1].level == level) ++f1;
        if (rootf && level == nlevels - 1) {
            const BigSweepPlan pl = big_sweep_plan(D, *roots, nrhs);
            const int nblk = big_group_blocks(*rootf), cwb = pl.cw / BIG_NB;
            if ((e = launch_big_gather(D, *rootf, st)) != hipSuccess) return e;
            // ONE release (every cross-stream edge costs the block chain about 10 us): the first k chunks go to
            // fj.aux after block launch k * cwb, k the largest count that the remaining launches still cover (one launch
            // of slack: the side queue starts a chunk 10-15 us after its release); the other chunks follow on st after
            // the join.
            int k = 0;
            while (k < pl.nchunk && (k + 1) * cwb <= nblk && nblk - (k + 1) * cwb >= k + 2) ++k;
            hipEvent_t home = nullptr;
            bool side_started = false;
            auto start_side = [&]() -> hipError_t {                        // sweep of the lower levels + the root's gather
                side_started = true;
                if (ready_deferred) {
                    hipError_t se;
                    if ((se = hipStreamWaitEvent(fj.aux, ready_deferred, 0)) != hipSuccess) return se;
                    if ((se = sweep(0, fork_level, fj.aux)) != hipSuccess) return se;
                }
                return launch_fwd_big_pre(D, *roots, X, nrhs, fj.aux);
            };
            auto release = [&]() -> hipError_t {                           // chunks 0 .. k - 1 hang off block launch k * cwb
                hipError_t se;
                if ((se = hipStreamWaitEvent(fj.aux, home, 0)) != hipSuccess) return se;
                for (int c = 0; c < k; ++c) {
                    se = (D.kind == CS3_LU) ? launch_fwd_big_chunk<CS3_LU>(D, *roots, X, nrhs, c, fj.aux)
                                            : launch_fwd_big_chunk<CS3_CHOLESKY>(D, *roots, X, nrhs, c, fj.aux);
                    if (se != hipSuccess) return se;
                }
                home = nullptr;
                return hipSuccess;
            };
            for (int blk = 0; blk <= nblk; ++blk) {
                e = (D.kind == CS3_LU) ? launch_big_block<CS3_LU>(D, *rootf, blk, inv_tol, st)
                                       : launch_big_block<CS3_CHOLESKY>(D, *rootf, blk, inv_tol, st);
                if (e != hipSuccess) return e;
                if (!side_started && (e = start_side()) != hipSuccess) return e;      // after the chain's first block is captured
                if (home && (e = release()) != hipSuccess) return e;                  // after the block that follows the release point
                if (k > 0 && blk == k * cwb) {                         // blocks 0 .. blk - 1 are home
                    if ((e = fj.event(&home)) != hipSuccess) return e;
                    if ((e = hipEventRecord(home, st)) != hipSuccess) return e;
                }
            }
            if (home && (e = release()) != hipSuccess) return e;
            root_rest = k;
            if ((e = fj.event(&swept)) != hipSuccess) return e;           // replaces the join recorded at the fork
            if ((e = hipEventRecord(swept, fj.aux)) != hipSuccess) return e;
            f0 = f1;
            continue;
        }
        e = run_level(fgroups, f0, f1, st, fj, true, [&](const LaunchGroup &g, hipStream_t s) {
            return (D.kind == CS3_LU) ? launch_front_group<CS3_LU>(D, g, inv_tol, s)
                                      : launch_front_group<CS3_CHOLESKY>(D, g, inv_tol, s);
        });
        if (e != hipSuccess) return e;
        if (ready_deferred && !rootf) {            // the level above the fork has been captured: now the side branch
            if ((e = hipStreamWaitEvent(fj.aux, ready_deferred, 0)) != hipSuccess) return e;
            if ((e = sweep(0, fork_level, fj.aux)) != hipSuccess) return e;
            if ((e = fj.event(&swept)) != hipSuccess) return e;
            if ((e = hipEventRecord(swept, fj.aux)) != hipSuccess) return e;
            ready_deferred = nullptr;
        }
        if (level == fork_level) {
            hipEvent_t ready;                      // panels of levels 0..fork_level are final
            if ((e = fj.event(&ready)) != hipSuccess) return e;
            if ((e = hipEventRecord(ready, st)) != hipSuccess) return e;
            ready_deferred = ready;                // captured after the chain above has begun (see the root's chain)
        }
        f0 = f1;
    }
    if (swept && (e = hipStreamWaitEvent(st, swept, 0)) != hipSuccess) return e;
    if (!rootf) return sweep(fork_level + 1, nlevels, st);
    const BigSweepPlan pl = big_sweep_plan(D, *roots, nrhs);
    for (int c = root_rest; c < pl.nchunk; ++c) {
        e = (D.kind == CS3_LU) ? launch_fwd_big_chunk<CS3_LU>(D, *roots, X, nrhs, c, st)
                               : launch_fwd_big_chunk<CS3_CHOLESKY>(D, *roots, X, nrhs, c, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_prologue(const DeviceFactor &D, const double *ax_src, const double *x_src, int nrhs, hipStream_t st)
{
    const long long nzero = D.zero_big ? D.vals_size - D.big_begin : 0;     // k_front_wg zeroes its own buffer
    const long long nax = (ax_src && ax_src != D.ax) ? D.batch * D.nnz_a : 0;
    const long long nf = ax_src ? D.n_sub_a : 0;
    const long long total = std::max(nzero * D.batch + nax + (x_src ? D.n * (long long) nrhs * D.batch : 0), nf * D.batch);
    hipLaunchKernelGGL(k_prologue, dim3(grid_for(std::max<long long>(total, 1), 256)), dim3(256), 0, st, D.status, D.pool_pm,
                       D.big_begin, nzero, D.pm_stride, D.batch, ax_src, D.ax, nax, x_src, D.xp, D.q, D.n, nrhs,
                       D.sub_a_src, D.axf, nf, D.nnz_a);
    CS3_LAUNCH_CHECK();
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

hipError_t launch_extract(const double *vals, const double *vals_il, long long il_len, const long long *map, double *out,
                          long long count, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_extract, dim3(grid_for(count, 256)), dim3(256), 0, st, vals, vals_il, il_len, map, out, count);
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
