// probe: operand / result lane layout of v_mfma_f64_4x4x4 (4 blocks) on gfx950
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_4x4x4_layout.hip -o tools/mfma_4x4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
int main() {
    double *dA, *dB, *dD, hA[64], hB[64], hD[64];
    (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 512);
    // for every (A lane, B lane) pair: which D lane receives the product?
    int dest[64][64];
    for (int la = 0; la < 64; ++la) {
        for (int q = 0; q < 64; ++q) hA[q] = (q == la) ? 1.0 : 0.0;
        for (int q = 0; q < 64; ++q) hB[q] = 1.0 + q;            // value identifies the B lane
        (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
        for (int lb = 0; lb < 64; ++lb) dest[la][lb] = -1;
        for (int d = 0; d < 64; ++d) if (hD[d] != 0.0) dest[la][(int)(hD[d] + 0.5) - 1] = d;
    }
    // print for a few A lanes the (B lane -> D lane) pairs
    for (int la = 0; la < 64; la += 1) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (dest[la][lb] >= 0) printf(" B%-2d->D%-2d", lb, dest[la][lb]);
        printf("\n");
    }
    return 0;
}
