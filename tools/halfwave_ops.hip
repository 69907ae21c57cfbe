// Probe of the two cross-lane operations the 64-lane Gauss-Jordan solve of the chain kernel relies on (gfx950):
//   v_permlane32_swap_b32 (x, x): result [0] = the lower half's values in both halves, [1] = the upper half's
//   ds_swizzle_b32, bit-mask mode, pattern j << 5: lane j of the lane's own group of 32 lanes
//   hipcc --offload-arch=gfx950 -O2 tools/halfwave_ops.hip -o tools/halfwave_ops && tools/halfwave_ops
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    const unsigned x = 100 + threadIdx.x;
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    const int s = __builtin_amdgcn_ds_swizzle((int)x, 5 << 5);
    out[threadIdx.x] = (int)r[0]; out[64 + threadIdx.x] = (int)r[1]; out[128 + threadIdx.x] = s;
}
int main() {
    int* d; int h[192];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (h[l] != 100 + (l & 31)) ++bad;
        if (h[64 + l] != 100 + 32 + (l & 31)) ++bad;
        if (h[128 + l] != 100 + (l & 32) + 5) ++bad;
    }
    printf("swap[0] lane 0 / 40: %d %d   swap[1] lane 0 / 40: %d %d   swizzle lane 0 / 40: %d %d   mismatches: %d\n",
           h[0], h[40], h[64], h[104], h[128], h[168], bad);
    return bad != 0;
}
