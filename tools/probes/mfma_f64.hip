// Probe: layout and cost of v_mfma_f64_16x16x4_f64 on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_f64.hip -o /tmp/mfma_f64 && /tmp/mfma_f64
// D(16x16) = A(16x4) B(4x16) + C.  Expected layout (MI355X_MICROARCH.md): A: row = lane & 15, k = lane >> 4;
// B: col = lane & 15, k = lane >> 4;  C/D register i: col = lane & 15, row = (lane >> 4) + 4 i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double double4v __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double *A, const double *B, const double *C, double *D)
{
    const int lane = threadIdx.x;
    const double a = A[(lane & 15) + 16 * (lane >> 4)];          // A[m][k] stored m + 16 k
    const double b = B[(lane >> 4) + 4 * (lane & 15)];           // B[k][n] stored k + 4 n
    double4v c;
    for (int i = 0; i < 4; ++i) c[i] = C[((lane >> 4) + 4 * i) + 16 * (lane & 15)];     // C[m][n] stored m + 16 n
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((lane >> 4) + 4 * i) + 16 * (lane & 15)] = c[i];
}

// chains: NCH independent accumulators, each updated `iters` times
template <int NCH>
__global__ void k_time(double *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
    double4v c[NCH];
    for (int n = 0; n < NCH; ++n) for (int i = 0; i < 4; ++i) c[n][i] = n + i;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NCH; ++n) c[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[n], 0, 0, 0);
    }
    double s = 0;
    for (int n = 0; n < NCH; ++n) for (int i = 0; i < 4; ++i) s += c[n][i];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the VALU equivalent of one 16x16x4 product on 64 rows x 16 columns x 1 pivot ... here: 16 columns, lane = row,
// 2 lane reads + 1 FMA per column and pivot (what the elimination loops issue today): per 1024 FMAs = 16 column updates
__global__ void k_valu(double *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    double d[16];
    for (int j = 0; j < 16; ++j) d[j] = 1.0 + j + lane * 1e-6;
    double l = 1e-3 * lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int k = it & 63;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int lo = __builtin_amdgcn_readlane((int) __double2loint(d[j]), k), hi = __builtin_amdgcn_readlane((int) __double2hiint(d[j]), k);
            d[j] -= l * __hiloint2double(hi, lo);
        }
    }
    double s = 0;
    for (int j = 0; j < 16; ++j) s += d[j];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    std::vector<double> A(64), B(64), C(256), D(256), R(256);
    for (int i = 0; i < 64; ++i) { A[i] = sin(1.0 + i); B[i] = cos(2.0 + 3 * i); }
    for (int i = 0; i < 256; ++i) C[i] = sin(0.1 * i);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double s = C[m + 16 * n];
        for (int k = 0; k < 4; ++k) s = fma(A[m + 16 * k], B[k + 4 * n], s);
        R[m + 16 * n] = s;
    }
    double *dA, *dB, *dC, *dD; long long *dcyc;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 1 << 22); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    double err = 0; int bitwise = 0;
    for (int i = 0; i < 256; ++i) { err = fmax(err, fabs(D[i] - R[i])); bitwise += (D[i] == R[i]); }
    printf("layout: max |D - ref| = %.3e, %d / 256 entries bit-equal to the k = 0..3 fma chain\n", err, bitwise);
    const int iters = 4096;
    long long cyc;
    auto report = [&](const char *what, double per) { hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost); printf("%-60s %8.1f cycles each\n", what, (double) cyc / per); };
    k_time<1><<<1, 64>>>(dD, dcyc, iters); report("mfma f64 16x16x4, one dependent chain, 1 wave", iters);
    k_time<4><<<1, 64>>>(dD, dcyc, iters); report("mfma f64 16x16x4, four chains, 1 wave", 4.0 * iters);
    k_time<4><<<1, 256>>>(dD, dcyc, iters); report("mfma f64 16x16x4, four chains, 4 waves (one per SIMD)", 4.0 * iters);
    k_time<4><<<1, 512>>>(dD, dcyc, iters); report("mfma f64 16x16x4, four chains, 8 waves (two per SIMD)", 4.0 * iters);
    k_valu<<<1, 64>>>(dD, dcyc, iters); report("VALU: 16 x (2 readlane + fma) = 1024 FMAs, 1 wave", iters);
    k_valu<<<1, 512>>>(dD, dcyc, iters); report("VALU: the same, 8 waves (two per SIMD)", iters);
    return 0;
}
