// Which workgroups of a 512-workgroup launch (256 threads, 71 KB of LDS: two per CU) share a CU on gfx950?
// Every workgroup records XCC_ID and HW_ID (SE, CU) and spins ~50 us so that all are resident together.
//   hipcc --offload-arch=gfx950 -O2 tools/wg_placement.hip -o tools/wg_placement && tools/wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void where(unsigned* out) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; lds[0] = 1; }
    const long long t0 = clock64();
    while (clock64() - t0 < 100000) { }
}
int main() {
    const int n = 512;
    unsigned* d; std::vector<unsigned> h(2 * n);
    hipMalloc(&d, h.size() * 4);
    hipFuncSetAttribute((const void*)where, hipFuncAttributeMaxDynamicSharedMemorySize, 71680);
    for (int rep = 0; rep < 2; ++rep) {
        where<<<n, 256, 71680>>>(d);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    }
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("distinct CUs: %zu\n", by_cu.size());
    int shown = 0, pairs_256 = 0, pairs_other = 0;
    for (auto& kv : by_cu) {
        if (kv.second.size() == 2 && kv.second[1] - kv.second[0] == 256) ++pairs_256; else ++pairs_other;
        if (shown++ < 12) { printf("  cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    }
    printf("CUs whose two workgroups are b and b + 256: %d, others: %d\n", pairs_256, pairs_other);
    // partners of the first workgroups
    for (int b = 0; b < 8; ++b)
        for (auto& kv : by_cu) for (int x : kv.second) if (x == b) { printf("workgroup %d shares with:", b); for (int y : kv.second) if (y != b) printf(" %d", y); printf("\n"); }
    return 0;
}
