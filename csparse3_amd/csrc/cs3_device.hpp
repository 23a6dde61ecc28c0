// Device-side descriptors shared by kernels.hip and api.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "cs3_internal.hpp"

namespace cs3 {

// One front (supernode).  Offsets are element offsets into the per-matrix
// vals / cb / cv pools and into the rel_idx / st_idx index arrays.
struct FrontMeta {
    long long lpan, upan, cb, cv, rel, st;
    int c0, r, w;
    int child_begin, child_end;   // range in child_idx
    int parent;                   // -1 for a root
};

// Everything the kernels read, resident in HBM for the life of the handle.
struct DeviceFactor {
    int kind = CS3_LU;
    long long n = 0, nnz_a = 0, batch = 1;
    long long vals_size = 0, cb_size = 0, cv_size = 0;
    FrontMeta *meta = nullptr;
    int *sched = nullptr, *child_idx = nullptr, *rel_idx = nullptr, *st_idx = nullptr;
    int *vsrc = nullptr;          // [vals_size] entry of A feeding each panel slot, or -1
    int *q = nullptr;             // [n] pivot order
    double *vals = nullptr;       // [batch][vals_size]
    double *cb = nullptr;         // [batch][cb_size]
    double *cv = nullptr;         // [batch][cv_size * nrhs_cap]
    double *xp = nullptr;         // [batch][n * nrhs_cap] right-hand sides in pivot order
    long long nrhs_cap = 0;
    int *status = nullptr;        // [1] first failing pivot column, INT_MAX when clean
};

hipError_t prepare_kernels();
hipError_t launch_assemble(const DeviceFactor &D, const double *Ax_dev, hipStream_t st);
hipError_t launch_factor_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &groups,
                                double inv_tol, hipStream_t st);
hipError_t launch_solve_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &groups,
                               double *X, int nrhs, bool forward, hipStream_t st);
hipError_t launch_permute(const DeviceFactor &D, const double *src, double *dst, int nrhs, bool scatter,
                          hipStream_t st);
hipError_t launch_extract(const double *vals, const long long *map, double *out, long long count,
                          hipStream_t st);
hipError_t launch_tri_level(const int *rows, int nrows, const int *Rp, const int *Rj, const long long *Rmap,
                            const long long *diag, const double *Gx, double *X, int nrhs, hipStream_t st);
hipError_t launch_matvec_rows(const int *Rp, const int *Rj, const double *Rx, const double *X, double *Y,
                              long long m, int nrhs, hipStream_t st);

}  // namespace cs3
