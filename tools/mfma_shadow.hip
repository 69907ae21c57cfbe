// micro-benchmark: does a vector instruction issue in the shadow of the wave's own MFMA on gfx950?
// hand-placed instruction order (inline asm), one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shadow.hip -o tools/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float g4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, long long* t, int n, const float* src) {
    g4 acc[12];
    for (int q = 0; q < 12; ++q) acc[q] = g4{0, 0, 0, 0};
    float a = src[threadIdx.x], b = src[threadIdx.x + 64];
    float z0 = a, z1 = b, z2 = a + 1, z3 = b + 1, z4 = a + 2, z5 = b + 2, z6 = a + 3, z7 = b + 3, m = src[1], c = src[2];
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %12, %13, %0\n\t"
            "v_fma_f32 %14, %14, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %1, %12, %13, %1\n\t"
            "v_fma_f32 %15, %15, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %12, %13, %2\n\t"
            "v_fma_f32 %16, %16, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %3, %12, %13, %3\n\t"
            "v_fma_f32 %17, %17, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %4, %12, %13, %4\n\t"
            "v_fma_f32 %18, %18, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %5, %12, %13, %5\n\t"
            "v_fma_f32 %19, %19, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %6, %12, %13, %6\n\t"
            "v_fma_f32 %20, %20, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %7, %12, %13, %7\n\t"
            "v_fma_f32 %21, %21, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %8, %12, %13, %8\n\t"
            "v_fma_f32 %14, %14, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %9, %12, %13, %9\n\t"
            "v_fma_f32 %15, %15, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %10, %12, %13, %10\n\t"
            "v_fma_f32 %16, %16, %22, %23\n\t"
            "v_mfma_f32_16x16x4_f32 %11, %12, %13, %11\n\t"
            "v_fma_f32 %17, %17, %22, %23\n\t"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(a), "+v"(b), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3), "+v"(z4), "+v"(z5), "+v"(z6), "+v"(z7) : "v"(m), "v"(c));
        if (MODE == 1) asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %12, %13, %0\n\t"
            "v_mfma_f32_16x16x4_f32 %1, %12, %13, %1\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %12, %13, %2\n\t"
            "v_mfma_f32_16x16x4_f32 %3, %12, %13, %3\n\t"
            "v_mfma_f32_16x16x4_f32 %4, %12, %13, %4\n\t"
            "v_mfma_f32_16x16x4_f32 %5, %12, %13, %5\n\t"
            "v_mfma_f32_16x16x4_f32 %6, %12, %13, %6\n\t"
            "v_mfma_f32_16x16x4_f32 %7, %12, %13, %7\n\t"
            "v_mfma_f32_16x16x4_f32 %8, %12, %13, %8\n\t"
            "v_mfma_f32_16x16x4_f32 %9, %12, %13, %9\n\t"
            "v_mfma_f32_16x16x4_f32 %10, %12, %13, %10\n\t"
            "v_mfma_f32_16x16x4_f32 %11, %12, %13, %11\n\t"
            "v_fma_f32 %14, %14, %22, %23\n\t"
            "v_fma_f32 %15, %15, %22, %23\n\t"
            "v_fma_f32 %16, %16, %22, %23\n\t"
            "v_fma_f32 %17, %17, %22, %23\n\t"
            "v_fma_f32 %18, %18, %22, %23\n\t"
            "v_fma_f32 %19, %19, %22, %23\n\t"
            "v_fma_f32 %20, %20, %22, %23\n\t"
            "v_fma_f32 %21, %21, %22, %23\n\t"
            "v_fma_f32 %14, %14, %22, %23\n\t"
            "v_fma_f32 %15, %15, %22, %23\n\t"
            "v_fma_f32 %16, %16, %22, %23\n\t"
            "v_fma_f32 %17, %17, %22, %23\n\t"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(a), "+v"(b), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3), "+v"(z4), "+v"(z5), "+v"(z6), "+v"(z7) : "v"(m), "v"(c));
        if (MODE == 2) asm volatile(
            "v_mfma_f32_16x16x4_f32 %0, %12, %13, %0\n\t"
            "v_mfma_f32_16x16x4_f32 %1, %12, %13, %1\n\t"
            "v_mfma_f32_16x16x4_f32 %2, %12, %13, %2\n\t"
            "v_mfma_f32_16x16x4_f32 %3, %12, %13, %3\n\t"
            "v_mfma_f32_16x16x4_f32 %4, %12, %13, %4\n\t"
            "v_mfma_f32_16x16x4_f32 %5, %12, %13, %5\n\t"
            "v_mfma_f32_16x16x4_f32 %6, %12, %13, %6\n\t"
            "v_mfma_f32_16x16x4_f32 %7, %12, %13, %7\n\t"
            "v_mfma_f32_16x16x4_f32 %8, %12, %13, %8\n\t"
            "v_mfma_f32_16x16x4_f32 %9, %12, %13, %9\n\t"
            "v_mfma_f32_16x16x4_f32 %10, %12, %13, %10\n\t"
            "v_mfma_f32_16x16x4_f32 %11, %12, %13, %11\n\t"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(a), "+v"(b), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3), "+v"(z4), "+v"(z5), "+v"(z6), "+v"(z7) : "v"(m), "v"(c));
    }
    const long long t1 = clock64();
    float s = z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7;
    for (int q = 0; q < 12; ++q) s += acc[q][0];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { t[2 * (threadIdx.x >> 6)] = t0; t[2 * (threadIdx.x >> 6) + 1] = t1; }
}
int main() {
    float* out; long long* t; float* src;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&t, 4096); (void)hipMalloc(&src, 4096); (void)hipMemset(src, 0, 4096);
    long long h[8]; const int n = 4000;
    const char* names[3] = {"12 x (mfma_f32_16x16x4, v_fma_f32) interleaved", "12 x mfma_f32_16x16x4, then 12 x v_fma_f32", "12 x mfma_f32_16x16x4 alone"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, t, n, src);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, out, t, n, src);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, out, t, n, src);
        }
        (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        long long first = h[0], last = h[1];
        for (int w = 0; w < 4; ++w) { first = std::min(first, h[2 * w]); last = std::max(last, h[2 * w + 1]); }
        printf("%-52s %8.1f cycles per trip\n", names[mode], double(last - first) / n);
    }
    return 0;
}
