// Rates and semantics of the cross-lane operations a register Gauss-Jordan solve can broadcast a matrix row with (gfx950):
//   ds_swizzle_b32 (bit-mask mode; the LDS crossbar, shared by the waves of a CU),
//   v_mov_b32 DPP row_newbcast:n (lane n of every row of 16 lanes; vector ALU, per SIMD),
//   v_permlane16_swap_b32 (x, x): [0] = the even rows' values in both rows of each half, [1] = the odd rows'
//   hipcc --offload-arch=gfx950 -O2 tools/crosslane_rate.hip -o tools/crosslane_rate && tools/crosslane_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int REP = 2000, UN = 16;
template <int MODE>
__global__ void rate(long long* cyc, int* sink) {
    int x[UN];
    for (int u = 0; u < UN; ++u) x[u] = threadIdx.x * 7 + u;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (MODE == 0) x[u] = __builtin_amdgcn_ds_swizzle(x[u], 5 << 5);
            else if (MODE == 1) x[u] = __builtin_amdgcn_update_dpp(0, x[u], 0x150 + 5, 0xf, 0xf, false);
            else { const auto p = __builtin_amdgcn_permlane16_swap((unsigned)x[u], (unsigned)x[u], false, false); x[u] = (int)p[0]; }
        }
    }
    const long long t1 = clock64();
    int s = 0;
    for (int u = 0; u < UN; ++u) s += x[u];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void probe(int* out) {
    const unsigned x = 100 + threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, (int)x, 0x150 + 5, 0xf, 0xf, false);
    const auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    out[64 + threadIdx.x] = (int)p[0]; out[128 + threadIdx.x] = (int)p[1];
}
int main() {
    long long* dc; int* ds;
    hipMalloc(&dc, 4096 * 8); hipMalloc(&ds, 4096 * 512 * 4);
    const char* names[3] = {"ds_swizzle_b32", "v_mov_b32 dpp row_newbcast", "v_permlane16_swap_b32"};
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {1, 4, 8}) {
            // one workgroup per CU (256 workgroups), `waves` wavefronts each: 1 = one SIMD, 4 = one per SIMD, 8 = two per SIMD
            for (int it = 0; it < 2; ++it) {
                if (mode == 0) rate<0><<<256, 64 * waves>>>(dc, ds);
                else if (mode == 1) rate<1><<<256, 64 * waves>>>(dc, ds);
                else rate<2><<<256, 64 * waves>>>(dc, ds);
                hipDeviceSynchronize();
            }
            std::vector<long long> h(256);
            hipMemcpy(h.data(), dc, 256 * 8, hipMemcpyDeviceToHost);
            double m = 0; for (auto v : h) m += v; m /= 256;
            printf("%-28s %d wave(s) per CU: %.1f cycles per instruction per wave, %.2f per CU\n", names[mode], waves,
                   m / (REP * UN), m / (REP * UN) / waves);
        }
    int* d; int h[192];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (h[l] != 100 + (l & 48) + 5) ++bad;                       // lane 5 of the lane's own row of 16
        if (h[64 + l] != 100 + (l & 32) + (l & 15)) ++bad;           // even row of the lane's half
        if (h[128 + l] != 100 + (l & 32) + 16 + (l & 15)) ++bad;     // odd row of the lane's half
    }
    printf("row_newbcast:5 lanes 3 / 20 / 40: %d %d %d   permlane16_swap [0] lanes 3 / 20 / 40: %d %d %d   [1]: %d %d %d   mismatches: %d\n",
           h[3], h[20], h[40], h[67], h[84], h[104], h[131], h[148], h[168], bad);
    return 0;
}
