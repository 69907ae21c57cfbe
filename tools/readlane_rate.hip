// micro-benchmark: the inner pattern of the register Cholesky (two v_readlane_b32 + one
// v_fma_f64 with the broadcast value as scalar operand), 1 / 2 waves per SIMD on one CU.
//   hipcc --offload-arch=gfx950 -O3 tools/readlane_rate.hip -o tools/readlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
__device__ __forceinline__ double bcast(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
template <int BATCH>
__global__ void k(double* out, long long* t, int n) {
    const int wave = threadIdx.x >> 6;
    double A[32];
    for (int q = 0; q < 32; ++q) A[q] = q + threadIdx.x * 1e-3;
    double l = 1.0 + threadIdx.x * 1e-6;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += BATCH) {
            double lk[BATCH];
#pragma unroll
            for (int r = 0; r < BATCH; ++r) lk[r] = bcast(l, k0 + r);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < BATCH; ++r) A[k0 + r] = fma(-l, lk[r], A[k0 + r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        l = A[0] * 1e-9 + 1.0;
    }
    const long long t1 = clock64();
    double s = 0; for (int q = 0; q < 32; ++q) s += A[q];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { t[2 * wave] = t0; t[2 * wave + 1] = t1; }
}
template <int BATCH> void run(double* out, long long* t) {
    long long h[64];
    const int n = 2000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int nw = 4 * wps;
        hipLaunchKernelGGL(k<BATCH>, dim3(1), dim3(64 * nw), 0, 0, out, t, n);
        hipLaunchKernelGGL(k<BATCH>, dim3(1), dim3(64 * nw), 0, 0, out, t, n);
        (void)hipMemcpy(h, t, 16 * nw, hipMemcpyDeviceToHost);
        long long first = h[0], last = h[1];
        for (int w = 0; w < nw; ++w) { first = std::min(first, h[2 * w]); last = std::max(last, h[2 * w + 1]); }
        printf("batch %d, %d wave(s)/SIMD: %.2f cycles per (2 readlane + fma) per SIMD\n", BATCH, wps,
               double(last - first) / (32.0 * n * wps));
    }
}
int main() {
    double* out; long long* t;
    (void)hipMalloc(&out, 1 << 16); (void)hipMalloc(&t, 4096);
    run<1>(out, t); run<8>(out, t);
    return 0;
}
