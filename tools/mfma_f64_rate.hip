// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double* out, long long* cyc, int n) {
    d4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_fma(double* out, long long* cyc, int n) {
    double a[8]; for (int q = 0; q < 8; ++q) a[q] = q;
    double x = 1.0 + threadIdx.x * 1e-9, y = threadIdx.x * 1e-7;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = fma(a[q], x, y);
    }
    long long t1 = clock64();
    double s = 0; for (int q = 0; q < 8; ++q) s += a[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    long long h[8];
    for (int waves = 1; waves <= 2; ++waves) {
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64 * 4 * waves), 0, 0, out, cyc, 1000);
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4: %d wave(s)/SIMD: %.1f cycles per MFMA per wave\n", waves, h[0] / 4000.0);
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(64 * 4 * waves), 0, 0, out, cyc, 1000);
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_fma_f64       : %d wave(s)/SIMD: %.2f cycles per FMA per wave\n", waves, h[0] / 8000.0);
    }
    return 0;
}
