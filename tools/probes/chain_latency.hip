// Probe: dependent-chain cost of the instructions on the pivot chain of the one-wave eliminations (gfx950).
// One wave, N dependent repetitions of each pattern, shader-clock cycles per repetition.
//   hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.hip && ./chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double bcast_lane(double x, int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(hi, lo);
}

__global__ void probe(double *out, long long *cyc, double seed)
{
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0 + 1e-9 * lane;
    long long t0, t1;
    constexpr int N = 1024;
    // 0: dependent v_fma_f64
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fma(x, y, 1e-9);
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[0] = (t1 - t0);
    // 1: dependent v_rcp_f64
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x) + 1.0;
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[1] = (t1 - t0);
    // 2: readlane (lane-to-scalar) feeding an fma, dependent
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fma(bcast_lane(x, i & 31), y, 1e-9);
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[2] = (t1 - t0);
    // 3: the pivot chain: broadcast, rcp + two Newton steps, scale
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        const double p = bcast_lane(x, i & 31);
        double r = __builtin_amdgcn_rcp(p);
        r = fma(fma(-p, r, 1.0), r, r);
        r = fma(fma(-p, r, 1.0), r, r);
        x = x * r + 1.5;
    }
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[3] = (t1 - t0);
    // 4: independent fmas (8 accumulators): issue rate
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = x + j;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fma(a[j], y, 1e-9);
    }
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[4] = (t1 - t0);
    // 5: lane-masked region around an fma (exec save / restore), dependent
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { if (lane > (i & 31)) x = fma(x, y, 1e-9); }
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[5] = (t1 - t0);
    // 6: LDS write + read back (volatile), dependent
    __shared__ double sh[64];
    volatile double *vs = sh;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { vs[lane] = x; x = vs[lane ^ 1] + 1e-9; }
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[6] = (t1 - t0);
    // 7: sqrt chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = sqrt(x) + 2.0;
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[7] = (t1 - t0);
    // 8: the lean pivot step of the shared fronts: 8 columns, lane-masked region, hand-over through LDS
    {
        __shared__ double lmm[8 * 64];
        __shared__ int rdy;
        volatile double *lm = lmm; volatile int *ready = &rdy;
        double d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = 1.0 + 0.01 * ((lane * 7 + j * 3) % 11) + (lane == j ? 8.0 : 0.0);
        t0 = __builtin_amdgcn_s_memtime();
        for (int rep = 0; rep < 128; ++rep) {
            double rp = 1.0 / bcast_lane(d[0], 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (lane > k) {
                    d[k] *= rp;
                    lm[k * 64 + lane] = d[k];
                    if (lane == 63) *ready = rep * 8 + k + 1;
                    if (k + 1 < 8) {
                        d[k + 1] -= d[k] * bcast_lane(d[k + 1], k);
                        const double p = bcast_lane(d[k + 1], k + 1);
                        double r = __builtin_amdgcn_rcp(p);
                        r = fma(fma(-p, r, 1.0), r, r);
                        rp = fma(fma(-p, r, 1.0), r, r);
                    }
                    double bc[8];
#pragma unroll
                    for (int j = k + 2; j < 8; ++j) bc[j] = bcast_lane(d[j], k);
#pragma unroll
                    for (int j = k + 2; j < 8; ++j) d[j] -= d[k] * bc[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = d[j] * 1e-3 + 1.0 + (lane == j ? 8.0 : 0.0);
        }
        t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) cyc[8] = (t1 - t0);
#pragma unroll
        for (int j = 0; j < 8; ++j) x += d[j];
    }
    double s = x;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[lane] = s;
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&cyc, 16 * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char *name[8] = {"dependent v_fma_f64", "dependent v_rcp_f64 (+ add)", "readlane pair -> fma, dependent", "pivot chain: bcast, rcp, 2 Newton, scale",
                           "independent fma x8 (per fma)", "lane-masked fma (exec save/restore)", "LDS write + read back", "sqrt (+ add) chain"};
    for (int i = 0; i < 8; ++i) printf("%-45s %8.1f cycles per repetition\n", name[i], h[i] / 1024.0);
    printf("%-45s %8.1f cycles per pivot\n", "lean 8-column pivot step with LDS hand-over", h[8] / 1024.0);
    return 0;
}
