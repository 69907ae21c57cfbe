// probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1) on gfx950, and its issue rate
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_4x4x1_layout.hip -o tools/mfma_4x4x1_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    const f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], f4{0, 0, 0, 0}, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[r * 64 + l] = d[r];
}
__global__ void rate(float* out, int n) {
    f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = a0[0] + a1[1] + a0[2] + a1[3];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (2.0f * n);
}
int main() {
    float *dA, *dB, *dD, hA[64], hB[64], hD[256];
    (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256); (void)hipMalloc(&dD, 1024);
    for (int la = 0; la < 64; la += 5) {
        for (int q = 0; q < 64; ++q) hA[q] = (q == la) ? 1.0f : 0.0f;
        for (int q = 0; q < 64; ++q) hB[q] = 1.0f + q;
        (void)hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        printf("A lane %2d:", la);
        for (int r = 0; r < 4; ++r) for (int d = 0; d < 64; ++d) if (hD[r * 64 + d] != 0.0f) printf(" B%-2d->D[reg %d][lane %2d]", (int)(hD[r * 64 + d] + 0.5f) - 1, r, d);
        printf("\n");
    }
    float* dO; (void)hipMalloc(&dO, 1024);
    hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, dO, 4096);
    float hO[65]; (void)hipMemcpy(hO, dO, 260, hipMemcpyDeviceToHost);
    printf("cycles per v_mfma_f32_4x4x1_16b_f32, two accumulators alternating, one wave: %.1f\n", hO[64]);
    return 0;
}
