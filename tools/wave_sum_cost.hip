// micro-benchmark: a wave-wide sum of one binary64 value per lane, result in every lane, on a lone wave of gfx950 -- three ways:
//   (a) four DPP steps inside the rows of 16 lanes + the four row results by v_readlane (wave_sum until round 4),
//   (b) four DPP steps + v_permlane16_swap + v_permlane32_swap,
//   (c) two v_mfma_f64_16x16x4_f64 with a matrix of ones (lanes i, i + 16, i + 32, i + 48 in the first, the four row sums in the second).
// Cycles (s_memtime) per reduction in a chain where every reduction needs the one before it, and the results against a host sum.
//   hipcc --offload-arch=gfx950 -O3 tools/wave_sum_cost.hip -o tools/wave_sum_cost && tools/wave_sum_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ double dpp_row(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rows16(double x) {
    x += dpp_row<0xB1>(x); x += dpp_row<0x4E>(x); x += dpp_row<0x141>(x); x += dpp_row<0x140>(x);
    return x;
}
__device__ __forceinline__ double sum_a(double x) {
    x = rows16(x);
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ double sum_b(double x) {
    x = rows16(x);
    {
        const unsigned lo = __double2loint(x), hi = __double2hiint(x);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double sum_c(double x) {
    // A (16 x 4): lane l supplies A[l % 16][l / 16]; B = ones (4 x 16); D[i][j] = x_i + x_(i+16) + x_(i+32) + x_(i+48): lane l holds
    // rows 4 (l / 16) .. + 3 of column l % 16
    const d4 z = {0.0, 0.0, 0.0, 0.0};
    const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 1.0, z, 0, 0, 0);
    const double s = (d[0] + d[1]) + (d[2] + d[3]);          // the sum over rows 4 g .. 4 g + 3 (g = l / 16), the same in all 16 lanes of the group
    const d4 e = __builtin_amdgcn_mfma_f64_16x16x4f64(s, 1.0, z, 0, 0, 0);   // A'[i][k] = s_k: D'[i][j] = s_0 + s_1 + s_2 + s_3
    return e[0];
}

__global__ void k_sum(double* out, long long* t, int mode, int n) {
    double x = 1.0 + 1e-3 * threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        double s = mode == 0 ? sum_a(x) : mode == 1 ? sum_b(x) : sum_c(x);
        x = fma(s, 1e-9, 1.0 + 1e-3 * threadIdx.x);           // (the next reduction needs this one)
    }
    const long long t1 = clock64();
    const double y = 1.0 + 1e-3 * threadIdx.x;
    out[threadIdx.x] = mode == 0 ? sum_a(y) : mode == 1 ? sum_b(y) : sum_c(y);
    out[64 + threadIdx.x] = x;
    if (threadIdx.x == 0) { t[0] = t0; t[1] = t1; }
}

int main() {
    double* out; long long* t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64);
    const char* names[3] = {"(a) DPP rows + 8 v_readlane", "(b) DPP rows + permlane16_swap + permlane32_swap", "(c) two v_mfma_f64_16x16x4 with ones"};
    double want = 0; for (int l = 0; l < 64; ++l) want += 1.0 + 1e-3 * l;
    for (int mode = 0; mode < 3; ++mode) {
        const int n = 2000;
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, 0, out, t, mode, 10);
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, 0, out, t, mode, n);
        hipDeviceSynchronize();
        long long h[2]; double r[64];
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); hipMemcpy(r, out, 512, hipMemcpyDeviceToHost);
        double err = 0; for (int l = 0; l < 64; ++l) err = fmax(err, fabs(r[l] - want));
        printf("%-55s %7.1f ticks per reduction (+ one fma), max |result - host sum| over the lanes %.1e\n", names[mode], (double)(h[1] - h[0]) / n, err);
    }
    return 0;
}
