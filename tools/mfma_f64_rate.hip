// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950,
// one workgroup on one CU, 1 / 2 / 4 waves per SIMD.  Every wave reports its own
// start and end (s_memtime); the figure printed is (last end - first start) per
// instruction issued BY ONE SIMD, so that an unfair arbiter cannot hide contention.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: MFMA in every wave; 1: FMA in every wave; 2: waves on even slots MFMA, odd slots FMA;
// mode 3: v_mfma_f64_4x4x4 (four 4x4x4 blocks) in every wave
__global__ void k_rate(double* out, long long* t, int n, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0) || (mode == 2 && ((wave >> 2) & 1) == 0);
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double f[8];
    for (int q = 0; q < 8; ++q) f[q] = q;
    double x = 1.0 + threadIdx.x * 1e-9, y = threadIdx.x * 1e-7;
    __syncthreads();
    const long long t0 = clock64();
    if (mode == 3) {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int i = 0; i < n; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, y, c3, 0, 0, 0);
        }
        f[0] = c0 + c1 + c2 + c3;
    } else if (do_mfma) {
        for (int i = 0; i < n; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < 8 * n; ++i) {      // 32 FMAs per MFMA-loop trip
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = fma(f[q], x, y);
        }
    }
    const long long t1 = clock64();
    double s = a0[0] + a1[1] + a2[2] + a3[3];
    for (int q = 0; q < 8; ++q) s += f[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { t[2 * wave] = t0; t[2 * wave + 1] = t1; }
}

int main() {
    double* out; long long* t;
    hipMalloc(&out, 1 << 20); hipMalloc(&t, 4096);
    long long h[64];
    const int n = 2000;
    const char* names[4] = {"mfma_f64_16x16x4 in every wave", "v_fma_f64 in every wave", "mfma in waves 0-3, fma in waves 4-7", "mfma_f64_4x4x4 (4 blocks) in every wave"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int wps = (mode == 2 ? 2 : 1); wps <= (mode == 2 ? 2 : (mode == 3 ? 2 : 4)); wps *= 2) {
            const int nw = 4 * wps;
            hipLaunchKernelGGL(k_rate, dim3(1), dim3(64 * nw), 0, 0, out, t, n, mode);   // warm-up
            hipLaunchKernelGGL(k_rate, dim3(1), dim3(64 * nw), 0, 0, out, t, n, mode);
            hipMemcpy(h, t, 16 * nw, hipMemcpyDeviceToHost);
            long long first = h[0], last = h[1];
            for (int w = 0; w < nw; ++w) { first = std::min(first, h[2 * w]); last = std::max(last, h[2 * w + 1]); }
            const double span = double(last - first);
            printf("%-36s %d wave(s)/SIMD: span %9.0f cycles;", names[mode], wps, span);
            if (mode == 0 || mode == 3) printf("  %.1f cycles per MFMA per SIMD\n", span / (4.0 * n * wps));
            else if (mode == 1) printf("  %.2f cycles per FMA per SIMD\n", span / (64.0 * n * wps));
            else {
                printf("  per-wave spans:");
                for (int w = 0; w < nw; ++w) printf(" %lld", h[2 * w + 1] - h[2 * w]);
                printf("  (alone: mfma %.0f, fma %.0f expected from the lines above)\n", 0.0, 0.0);
            }
        }
    }
    return 0;
}
