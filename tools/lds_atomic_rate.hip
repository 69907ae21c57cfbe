// micro-benchmark: LDS floating-point atomics, four waves of a workgroup adding 48 lane-contiguous
// registers each (the tile sums of the fused pass).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_rate.hip -o tools/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* t, int n, const float* src) {
    extern __shared__ double lds[];
    float* Wf = reinterpret_cast<float*>(lds);
    double* Wd = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 48 * 64 * 4; i += 256) Wd[i] = 0.0;
    float v[48];
    for (int q = 0; q < 48; ++q) v[q] = src[threadIdx.x + q];       // zeros at run time, unknown at compile time
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int q = 0; q < 48; ++q) {
            if (MODE == 0) __hip_atomic_fetch_add(Wf + q * 64 + lane, v[q] + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add(Wd + q * 64 + lane, (double)v[q] + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_add(Wf + (wave * 48 + q) * 64 + lane, v[q] + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) __hip_atomic_fetch_add(Wf + q * 64 + lane, v[q] + 1e-40f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // denormal sums
            if (MODE == 4) Wf[(wave * 48 + q) * 64 + lane] = v[q] + 1.0f;      // plain stores, own region
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    if (lane == 0) { t[2 * wave] = t0; t[2 * wave + 1] = t1; }
    out[threadIdx.x] = Wf[threadIdx.x] + (float)Wd[threadIdx.x + 4096];
}
int main() {
    float* out; long long* t; float* src;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&t, 4096); (void)hipMalloc(&src, 8192); (void)hipMemset(src, 0, 8192);
    long long h[8]; const int n = 200;
    const char* names[5] = {"48 x ds_add_f32, four waves, same addresses", "48 x ds_add_f64, four waves, same addresses",
                            "48 x ds_add_f32, own region per wave", "48 x ds_add_f32, denormal sums", "48 x ds_write_b32, own region per wave"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            const size_t sh = 48 * 64 * 4 * 8;
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), sh, 0, out, t, n, src);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), sh, 0, out, t, n, src);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), sh, 0, out, t, n, src);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), sh, 0, out, t, n, src);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), sh, 0, out, t, n, src);
        }
        (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("%-50s %8.1f cycles per trip of 48\n", names[mode], double(h[1] - h[0]) / n);
    }
    return 0;
}
