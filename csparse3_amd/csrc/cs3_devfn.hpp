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


}  // namespace cs3
