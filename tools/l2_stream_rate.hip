// What one CU gets out of its XCD's L2 when every CU streams the same small table (V: 224 KB) again and again -- the
// streaming passes of the lock-step kernel -- by access shape (gfx950):
//   0  the fused pass: 8 B per lane, lane (kq = l >> 4, cn = l & 15) reads V[row + kq][16 t + cn], t = 0..3: four 128-B segments
//      512 B apart per wave-instruction
//   1  the row pass: 16 B per lane, lane (ak = l >> 4, m = l & 15) reads Vt[row + ak][2 m .. 2 m + 1]: four 256-B segments a row
//      (4 KB) apart
//   2  16 B per lane, 1 KB contiguous per wave-instruction
//   3  8 B per lane, 512 B contiguous per wave-instruction
// hipcc --offload-arch=gfx950 -O2 tools/l2_stream_rate.hip -o tools/l2_stream_rate && tools/l2_stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int ROWS = 512, NP = 64, REP = 40;          // 512 x 64 doubles = 256 KB
template <int MODE, int NWV, int PAD = 0>
__global__ __launch_bounds__(64 * NWV) void stream(const double* __restrict__ V, long long* cyc, double* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < REP; ++r) {
        if (MODE == 0) {
            const int kq = lane >> 4, cn = lane & 15;
#pragma unroll 4
            for (int g = wave; g < ROWS / 4; g += NWV) {
                const double* p = V + (size_t)(4 * g + kq) * NP + cn;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc += p[16 * t];
            }
        } else if (MODE == 1) {
            // Vt [NP][ROWS]: rows of 4 KB; a chunk = four rows of Vt, a block = 32 omega columns
            const int ak = lane >> 4, m = lane & 15;
#pragma unroll 2
            for (int kc = 0; kc < NP / 4; ++kc) {
#pragma unroll
                for (int b = wave; b < ROWS / 32; b += NWV) {
                    const double2 x = *reinterpret_cast<const double2*>(V + (size_t)(4 * kc + ak) * (ROWS + PAD) + 32 * b + 2 * m);
                    acc += x.x + x.y;
                }
            }
        } else if (MODE == 2) {
#pragma unroll 8
            for (int c = wave; c < ROWS * NP / 128; c += NWV) {
                const double2 x = *reinterpret_cast<const double2*>(V + (size_t)c * 128 + 2 * lane);
                acc += x.x + x.y;
            }
        } else {
#pragma unroll 16
            for (int c = wave; c < ROWS * NP / 64; c += NWV) acc += V[(size_t)c * 64 + lane];
        }
    }
    const long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* dV; long long* dc; double* ds;
    std::vector<double> h((size_t)(ROWS + 64) * NP, 1.0);
    hipMalloc(&dV, h.size() * 8); hipMalloc(&dc, 1024 * 8); hipMalloc(&ds, (size_t)1024 * 1024 * 8);
    hipMemcpy(dV, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const char* names[6] = {"fused pass: 4 x 128 B, 512 B apart", "row pass: 4 x 256 B, 4 KB apart", "1 KB contiguous (16 B / lane)", "512 B contiguous (8 B / lane)",
                            "row pass, rows 4 KB + 256 B apart", "row pass, rows 4 KB + 128 B apart"};
    for (int mode = 0; mode < 6; ++mode)
        for (int nwg : {16, 256, 512})
            for (int nwv : {4, 8, 16}) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                float ms = 0;
                for (int it = 0; it < 3; ++it) {
                    hipEventRecord(e0);
#define L(M, W) stream<(M > 3 ? 1 : M), W, (M == 4 ? 32 : M == 5 ? 16 : 0)><<<nwg, 64 * W>>>(dV, dc, ds)
#define LW(M) do { if (nwv == 4) L(M, 4); else if (nwv == 8) L(M, 8); else L(M, 16); } while (0)
                    if (mode == 0) LW(0); else if (mode == 1) LW(1); else if (mode == 2) LW(2); else if (mode == 3) LW(3); else if (mode == 4) LW(4); else LW(5);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                std::vector<long long> hc(nwg);
                hipMemcpy(hc.data(), dc, nwg * 8, hipMemcpyDeviceToHost);
                double m = 0; for (auto v : hc) m += v; m /= nwg;
                const double bytes = (double)REP * ROWS * NP * 8;
                printf("%-38s %3d workgroups x %2d waves: %6.1f B/clk per workgroup (%.0f cycles per pass of 256 KB), wall %.3f ms = %.1f TB/s\n",
                       names[mode], nwg, nwv, bytes / m, m / REP, ms, nwg * bytes / (ms * 1e-3) / 1e12);
            }
    return 0;
}
