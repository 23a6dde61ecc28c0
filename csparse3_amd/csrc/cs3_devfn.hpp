// Small device functions shared by the kernel translation units (kernels.hip, forest.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cs3 {

// Load-then-select: a predicated `cond ? p[i] : 0` makes hipcc branch around the
// load and wait for it alone; loading from a safe address keeps loads in flight.
__device__ __forceinline__ double load_if(const double *__restrict__ p, long long off, bool ok)
{
    const double v = p[ok ? off : 0];
    return ok ? v : 0.0;
}

// 1 / x on the critical path of every pivot: v_rcp_f64 and two Newton steps (about 1 ulp) instead of the
// IEEE division sequence (div_scale / fmas / fixup, three times the dependent instructions).  A zero
// or non-finite pivot still yields inf / nan, which the pivot checks reject.
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

// sqrt(x) and 1 / sqrt(x) together, for the Cholesky pivots: v_rsq_f64, two coupled Newton steps (Goldschmidt: g -> sqrt x,
// h -> 1 / (2 sqrt x)), one correction of the root (about 1 ulp both).  The library sqrt followed by fast_rcp is some forty
// dependent instructions on the critical path of every pivot (650 cycles per pivot measured on a one-wave panel);
// these are ten.  x <= 0 or non-finite yields nan / inf, which the pivot checks reject (they test the pivot itself).
__device__ __forceinline__ void sqrt_and_rsqrt(double x, double &s, double &rs)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double e = fma(-g, h, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-g, h, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    s = fma(fma(-g, g, x), h, g);
    rs = h + h;
}

// The pivot's diagonal entry of the factor and its reciprocal: LU keeps the pivot, Cholesky its root.
template <int KIND>
__device__ __forceinline__ void pivot_scale(double piv, double &dg, double &rp)
{
    if (KIND == CS3_CHOLESKY) sqrt_and_rsqrt(piv, dg, rp);
    else { dg = piv; rp = fast_rcp(piv); }
}

// 1 / diagonal entry `i` of an r-row panel: the sweeps multiply by it (one division per lane instead
// of one per pivot step executed by the whole wave)
__device__ __forceinline__ double recip_diag(const double *__restrict__ L, long long i, long long r, bool ok)
{
    const double dg = L[ok ? i * (r + 1) : 0];
    return 1.0 / (ok ? dg : 1.0);
}

__device__ __forceinline__ int bcast_lane_i(int x, int k) { return __builtin_amdgcn_readlane(x, k); }   // k wave-uniform

__device__ __forceinline__ double bcast_lane(double x, int k)     // k wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void flag_column(int *status, int col)
{
    atomicMin(status, col);
}

// ---- multipliers handed from wave to wave through LDS (eliminate_pair / eliminate_parts, the shared fronts of forest.hip) --
// The producer stores a vector of multipliers, then the number of pivots handed over so far; the consumer polls that number.
// LDS operations of one wave complete in order, so no fence sits on the producer's pivot chain (a release fence per pivot
// measured 350 instead of 275 cycles per pivot); the counter is accessed with relaxed workgroup-scope atomics so that the
// contract with the compiler is explicit, the multipliers through volatile LDS pointers.
typedef volatile __attribute__((address_space(3))) double *lds_vdouble_ptr;
typedef __attribute__((address_space(3))) int *lds_int_ptr;

// CS3_DEBUG_WITHHOLD=1 (cs3_debug_withhold_handover): producers keep their counters back, so that every consumer runs
// into its time-out -- the test of the give-up path.  One copy per translation unit (no relocatable device code in this
// build): kernels.hip and forest.hip each export a setter.
static __device__ int g_withhold_handover = 0;

__device__ __forceinline__ void handover_publish(lds_int_ptr ready, int value, bool withhold)
{
    if (!withhold) __hip_atomic_store(ready, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Waits until the counter exceeds `need`; false when the wait was given up.  A consumer that gives up raises status[3]:
// cs3_factor_status then reports the step as failed (CS3_ERR_STATE) instead of letting a wrong factor pass.
__device__ __forceinline__ bool handover_wait(lds_int_ptr ready, int need, int *status, bool withhold)
{
    const int limit = withhold ? (1 << 8) : (1 << 22);
    for (int it = 0; __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= need; ++it) {
        if (it >= limit) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// Pins a value to the place where it was computed.  The waits split a consumer's pivot steps into basic blocks of their
// own; an FMA whose result is only read later is then SUNK past them, block after block, while the lane reads that
// feed it cannot move: their scalar results pile up (440 of them spilled to vector lanes and read back, in the Cholesky
// consumer of a four-wave front) and the FMAs all run at the end.
__device__ __forceinline__ void keep_here(double &x) { asm volatile("" : "+v"(x)); }

// ---- one wave eliminates a front of order <= 32: lane = row, 32 register columns (the forest's tasks and the level
// kernels' one-wave fronts share it).  Elimination only: column k of the registers is column k of the front for the whole
// loop (no stores, no shifting inside it); the reciprocal of the next pivot is issued right after the first column update
// of a step, behind which its latency hides.  Pivot checks only raise `suspect` (the caller then looks for the column, a
// rare path); RHS carries one vector column along (the fused forward sweep).
//
// PANEL form (UT, LU only): the caller holds only the w pivot columns in `row` (r = w here: no column beyond them is
// touched) and, in `ut`, the pivot ROWS transposed -- lane = a column to the right of the block, ut[i] = F(i, that
// column).  Step k then also applies its multipliers to them, ut[i] -= L(i, k) ut[k] (L(i, k) sits in lane i of the
// multiplier register), which leaves U12 in `ut`; the trailing matrix is left to schur_tiles (MFMA).
template <int KIND, bool RHS, bool UT = false>
__device__ __forceinline__ void sub_eliminate(double (&row)[32], double &rhs, int r, int w, double inv_tol, bool &suspect,
                                              double (&ut)[32])
{
    constexpr int NC = 32;
    const int lane = threadIdx.x & 63;
    double piv = bcast_lane(row[0], 0);
    double dg, rp;
    pivot_scale<KIND>(piv, dg, rp);
#pragma unroll
    for (int k0 = 0; k0 < NC; k0 += 8) {
      if (k0 < w) {                                             // (eight steps skipped at once past the last pivot)
#pragma unroll
       for (int k = k0; k < k0 + 8; ++k) {
        if (k < w) {
            const bool below = lane > k;
            const double l = below ? row[k] * rp : 0.0;         // multiplier, zero on and above the pivot row
            if (below) row[k] = l;
            if (KIND == CS3_CHOLESKY && lane == k) row[k] = (piv > 0.0) ? dg : -1.0;
            if (KIND == CS3_LU) suspect = (int) suspect | (int) !(fabs(l) <= inv_tol) | (int) !(fabs(piv) > 0.0) | (int) !(fabs(piv) < 1.0e300);
            else suspect = (int) suspect | (int) !(piv > 0.0);
            // (the vector is scaled by what the separate forward sweep multiplies by, bit for bit: fast_rcp of the stored root)
            const double rpk = (RHS && KIND == CS3_CHOLESKY) ? fast_rcp(dg) : rp;
            if (k + 1 < NC) {
                if (KIND == CS3_LU) row[k + 1] -= l * bcast_lane(row[k + 1], k);
                else { const double lj = bcast_lane(row[k], k + 1); row[k + 1] -= l * lj; }
                piv = bcast_lane(row[k + 1], k + 1);
                pivot_scale<KIND>(piv, dg, rp);
            }
            if (RHS) {                                          // forward substitution: y_k final, rows below take it
                if (KIND == CS3_CHOLESKY && lane == k) rhs *= rpk;
                rhs -= l * bcast_lane(rhs, k);
            }
#pragma unroll
            for (int j0 = (k + 2) & ~7; j0 < NC; j0 += 8) {
                if (j0 < r) {                                   // skip register groups beyond the front
                    double bc[8];
#pragma unroll
                    for (int j = (j0 > k + 2 ? j0 : k + 2); j < j0 + 8; ++j)
                        bc[j - j0] = (KIND == CS3_LU) ? bcast_lane(row[j], k) : bcast_lane(row[k], j);
#pragma unroll
                    for (int j = (j0 > k + 2 ? j0 : k + 2); j < j0 + 8; ++j) row[j] -= l * bc[j - j0];
                }
            }
            if (UT && KIND == CS3_LU) {
#pragma unroll
                for (int i0 = (k + 1) & ~7; i0 < NC; i0 += 8) {
                    if (i0 < w) {
                        double bc[8];
#pragma unroll
                        for (int i = (i0 > k + 1 ? i0 : k + 1); i < i0 + 8; ++i) bc[i - i0] = bcast_lane(l, i);
#pragma unroll
                        for (int i = (i0 > k + 1 ? i0 : k + 1); i < i0 + 8; ++i) ut[i] -= bc[i - i0] * ut[k];
                    }
                }
            }
        }
       }
      }
    }
}
template <int KIND, bool RHS>
__device__ __forceinline__ void sub_eliminate(double (&row)[32], double &rhs, int r, int w, double inv_tol, bool &suspect)
{
    double none[32];
    sub_eliminate<KIND, RHS, false>(row, rhs, r, w, inv_tol, suspect, none);
}

// The pivot ROWS of an LU front by themselves, transposed: lane = a column of the front (all r of them, the pivot columns
// included), ut[i] = F(i, column) for the w pivot rows.  Row k is final when its turn comes; the rows below take
// ut[i] -= (F(i, k) / pivot) ut[k] with the multiplier formed from lane k of ut[i] -- a wave-uniform value, so this wave needs
// nothing from the wave that factors the pivot COLUMNS (sub_eliminate), and the two run side by side.  Every entry sees
// the same operations in the same order as in the one-wave panel form: U11 comes out identical in both waves (only the
// columns to the right of the block, U12, are taken from here).
__device__ __forceinline__ void rows_eliminate_lu(double (&ut)[32], int w)
{
    constexpr int NC = 32;
    double piv = bcast_lane(ut[0], 0);
    double rp = fast_rcp(piv);
#pragma unroll
    for (int k0 = 0; k0 < NC; k0 += 8) {
      if (k0 < w) {
#pragma unroll
       for (int k = k0; k < k0 + 8; ++k) {
        if (k < w) {
            const double rpk = rp;
            if (k + 1 < NC) {
                const double l1 = bcast_lane(ut[k + 1], k) * rpk;
                ut[k + 1] -= l1 * ut[k];
                piv = bcast_lane(ut[k + 1], k + 1);
                rp = fast_rcp(piv);
            }
#pragma unroll
            for (int i0 = (k + 2) & ~7; i0 < NC; i0 += 8) {
                if (i0 < w) {
                    double li[8];
#pragma unroll
                    for (int i = (i0 > k + 2 ? i0 : k + 2); i < i0 + 8; ++i) li[i - i0] = bcast_lane(ut[i], k) * rpk;
#pragma unroll
                    for (int i = (i0 > k + 2 ? i0 : k + 2); i < i0 + 8; ++i) ut[i] -= li[i - i0] * ut[k];
                }
            }
        }
       }
      }
    }
}

// ---- the trailing matrix of a front whose image lives in LDS (column-major, leading dimension ld), by
// v_mfma_f64_16x16x4 (entry (i, j) of the image at F[at(i, j)]: column-major, or the packed lower triangle for Cholesky):
//   C(i, j) - sum_{k < w} L(i, k) U(k, j)   for i, j in [w, r), in 16 x 16 tiles dealt to `nwaves`
// waves; every entry of the result is handed to out(i, j, value) exactly once (Cholesky: the tiles on and below the
// diagonal; the caller drops i < j inside a diagonal tile).  L(i, k) = F[i + k ld]; U(k, j) = F[k + j ld] (LU), or
// L(j, k) (Cholesky).  The sum runs in pivot order with fused multiply-adds, like the column-by-column updates it
// replaces (bit-equal: measured with tools/probes/mfma_f64.hip), at 64 cycles per 1024 of them instead of 490 (two lane
// reads and one FMA per column and pivot).  Tiles are computed TRANSPOSED (A operand = U, B operand = -L), so that the 16
// lanes of an output register hold 16 consecutive ROWS of one column: stores to a column-major block coalesce.
typedef double cs3_double4 __attribute__((ext_vector_type(4)));
template <int KIND, class At, class Out>
__device__ __forceinline__ void schur_tiles(const double *F, At at, int r, int w, int wave, int nwaves, Out out)
{
    const int lane = threadIdx.x & 63, mi = lane & 15, mq = lane >> 4;
    const int nt = (r - w + 15) >> 4;
    int t = 0;
    for (int tj = 0; tj < nt; ++tj) {
        for (int ti = (KIND == CS3_CHOLESKY) ? tj : 0; ti < nt; ++ti, ++t) {
            if (t % nwaves != wave) continue;                   // (wave-uniform)
            const int i = w + 16 * ti + mi;                     // my row: B operand and the outputs' lane & 15
            const int ca = w + 16 * tj + mi;                    // my column of the A operand
            const int ic = min(i, r - 1), cc = min(ca, r - 1);  // (rows / columns past the front: a copy of the last one --
                                                                //  finite values that only reach outputs nobody takes)
            cs3_double4 acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = F[at(ic, min(w + 16 * tj + mq + 4 * v, r - 1))];
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += 4) {
                if (k0 < w) {
                    const int k = k0 + mq;
                    const bool kin = k < w;
                    const int kc = kin ? k : 0;
                    const double au = F[(KIND == CS3_LU) ? at(kc, cc) : at(cc, kc)];
                    const double bl = F[at(ic, kc)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kin ? au : 0.0, kin ? -bl : 0.0, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = w + 16 * tj + mq + 4 * v;
                if (i < r && c < r) out(i, c, acc[v]);
            }
        }
    }
}

}  // namespace cs3
