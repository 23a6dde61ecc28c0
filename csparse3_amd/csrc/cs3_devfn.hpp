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

// ---- one wave eliminates a front of order <= 32: lane = row, 32 register columns (the forest's tasks and the level
// kernels' one-wave fronts share it).  Elimination only: column k of the registers is column k of the front for the whole
// loop (no stores, no shifting inside it); the reciprocal of the next pivot is issued right after the first column update
// of a step, behind which its latency hides.  Pivot checks only raise `suspect` (the caller then looks for the column, a
// rare path); RHS carries one vector column along (the fused forward sweep).
template <int KIND, bool RHS>
__device__ __forceinline__ void sub_eliminate(double (&row)[32], double &rhs, int r, int w, double inv_tol, bool &suspect)
{
    constexpr int NC = 32;
    const int lane = threadIdx.x & 63;
    double piv = bcast_lane(row[0], 0);
    double dg = (KIND == CS3_CHOLESKY) ? sqrt(piv) : piv;
    double rp = fast_rcp(dg);
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
            const double rpk = rp;
            if (k + 1 < NC) {
                if (KIND == CS3_LU) row[k + 1] -= l * bcast_lane(row[k + 1], k);
                else { const double lj = bcast_lane(row[k], k + 1); row[k + 1] -= l * lj; }
                piv = bcast_lane(row[k + 1], k + 1);
                dg = (KIND == CS3_CHOLESKY) ? sqrt(piv) : piv;
                rp = fast_rcp(dg);
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
                    for (int j = (j0 > k + 2 ? j0 : k + 2); j < j0 + 8; ++j) {
                        if (KIND == CS3_LU) row[j] -= l * bc[j - j0];
                        else row[j] -= l * bc[j - j0];
                    }
                }
            }
        }
       }
      }
    }
}

}  // namespace cs3
