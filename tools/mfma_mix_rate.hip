// micro-benchmark for the fused pass of the lock-step kernel: cycles per row group of
//   mode 0: 12 x v_mfma_f32_16x16x4_f32 (independent accumulators)
//   mode 1: + 4 x v_mfma_f64_4x4x4
//   mode 2: + the vector work of one group (2 v_cvt_f32_f64, 8 v_mul_f32) from fresh registers
//   mode 3: mode 1 with the 4x4x4 forms first
//   mode 4: 16 x f32 MFMA (no f64)
// one workgroup, one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/mfma_mix_rate.hip -o tools/mfma_mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float g4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k_mix(double* out, long long* t, int n, const double* src) {
    const int wave = threadIdx.x >> 6;
    g4 acc[16];
    for (int q = 0; q < 16; ++q) acc[q] = g4{0, 0, 0, 0};
    double hp[4] = {0, 0, 0, 0};
    double f0 = src[threadIdx.x], f1 = src[threadIdx.x + 64], hh = src[threadIdx.x + 128];
    float w[4] = {(float)src[1], (float)src[2], (float)src[3], (float)src[4]};
    float ff0 = (float)f0, ff1 = (float)f1;
    float a[4][2];
    for (int c = 0; c < 4; ++c) { a[c][0] = ff0 * w[c]; a[c][1] = ff1 * w[c]; }
    float z[8] = {w[0], w[1], w[2], w[3], ff0, ff1, 1.0f, 2.0f};
    double zd[4] = {f0, f1, hh, 1.0};
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 3)
            for (int q = 0; q < 4; ++q) hp[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(f0, hh, hp[q], 0, 0, 0);
        if (MODE == 2) {
            ff0 = (float)f0; ff1 = (float)f1;
            for (int c = 0; c < 4; ++c) { a[c][0] = ff0 * w[c]; a[c][1] = ff1 * w[c]; }
            f0 += 1.0;      // keeps the conversions in the loop
        }
        for (int c = 0; c < 4; ++c) {
            acc[3 * c + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][0], ff0, acc[3 * c + 0], 0, 0, 0);
            acc[3 * c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][0], ff1, acc[3 * c + 1], 0, 0, 0);
            acc[3 * c + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][1], ff1, acc[3 * c + 2], 0, 0, 0);
        }
        if (MODE == 5 || MODE == 6) {
            // independent vector work (nothing the MFMAs read): does it issue in their shadow?
#pragma unroll
            for (int q = 0; q < (MODE == 5 ? 12 : 24); ++q) { z[q & 7] = z[q & 7] * w[q & 3] + 1.0f; }
        }
        if (MODE == 7) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { zd[q & 3] = fma(zd[q & 3], hh, f1); }
        }
        if (MODE == 4)
            for (int q = 12; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][0], ff1, acc[q], 0, 0, 0);
        if (MODE == 1 || MODE == 2)
            for (int q = 0; q < 4; ++q) hp[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(f0, hh, hp[q], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = hp[0] + hp[1] + hp[2] + hp[3];
    for (int q = 0; q < 16; ++q) s += acc[q][0] + acc[q][3];
    for (int q = 0; q < 8; ++q) s += z[q];
    for (int q = 0; q < 4; ++q) s += zd[q];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { t[2 * wave] = t0; t[2 * wave + 1] = t1; }
}

int main() {
    double* out; long long* t; double* src;
    hipMalloc(&out, 1 << 16); hipMalloc(&t, 4096); hipMalloc(&src, 4096);
    hipMemset(src, 0, 4096);
    long long h[64];
    const int n = 4000;
    const char* names[8] = {"12 x mfma_f32_16x16x4", "12 x f32 + 4 x mfma_f64_4x4x4", "12 x f32 + 4 x f64_4x4x4 + cvt/mul of a group",
                            "4 x f64_4x4x4 first, then 12 x f32", "16 x mfma_f32_16x16x4",
                            "12 x f32 MFMA + 12 independent v_fma_f32", "12 x f32 MFMA + 24 independent v_fma_f32", "12 x f32 MFMA + 8 independent v_fma_f64"};
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_mix<0>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
                case 1: hipLaunchKernelGGL(k_mix<1>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
                case 2: hipLaunchKernelGGL(k_mix<2>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
                case 3: hipLaunchKernelGGL(k_mix<3>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
                case 4: hipLaunchKernelGGL(k_mix<4>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
                case 5: hipLaunchKernelGGL(k_mix<5>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
                case 6: hipLaunchKernelGGL(k_mix<6>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
                default: hipLaunchKernelGGL(k_mix<7>, dim3(1), dim3(256), 0, 0, out, t, n, src); break;
            }
        }
        hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        long long first = h[0], last = h[1];
        for (int w = 0; w < 4; ++w) { first = std::min(first, h[2 * w]); last = std::max(last, h[2 * w + 1]); }
        printf("%-48s %8.1f cycles per group\n", names[mode], double(last - first) / n);
    }
    return 0;
}
