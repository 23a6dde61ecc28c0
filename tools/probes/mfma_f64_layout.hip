// Probe: operand layout of v_mfma_f64_16x16x4_f64 on gfx950.  Assumed (and checked here):
//   A[i][k] in lane i + 16 k,  B[k][j] in lane j + 16 k,  D[4 (lane / 16) + v][lane % 16] in register v.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_f64_layout.hip -o mfma_probe ; prints "layout ok" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k(const double *A, const double *B, double *D)
{
    const int lane = threadIdx.x;
    const int i = lane % 16, kk = lane / 16;
    const double a = A[i * 4 + kk];          // A[i][k], row-major 16 x 4
    const double b = B[kk * 16 + i];         // B[k][j], row-major 4 x 16, j = lane % 16
    double4_t c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[lane * 4 + v] = c[v];          // raw: [lane][register]
}

int main()
{
    double hA[64], hB[64], hD[256], ref[256];
    srand(1);
    for (int t = 0; t < 64; ++t) { hA[t] = rand() / (double) RAND_MAX; hB[t] = rand() / (double) RAND_MAX; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int q = 0; q < 4; ++q) s += hA[i * 4 + q] * hB[q * 16 + j];
        ref[i * 16 + j] = s;
    }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // discover where each (lane, register) value sits in the reference product
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int v = 0; v < 4; ++v) {
        int hit = -1;
        for (int t = 0; t < 256; ++t) if (fabs(hD[lane * 4 + v] - ref[t]) < 1e-13) { hit = t; break; }
        if (lane < 20 || lane % 16 == 0) printf("lane %2d reg %d -> D[%2d][%2d]\n", lane, v, hit / 16, hit % 16);
        const int ei = 4 * (lane / 16) + v, ej = lane % 16;              // first guess
        const int fi = (lane / 16) + 4 * v, fj = lane % 16;              // second guess
        if (hit != fi * 16 + fj) bad |= 2;
        if (hit != ei * 16 + ej) bad |= 1;
    }
    printf("guess i = 4 (lane/16) + v: %s;  guess i = lane/16 + 4 v: %s\n", (bad & 1) ? "no" : "YES", (bad & 2) ? "no" : "YES");
    return 0;
}
