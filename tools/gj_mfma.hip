// micro-test: the 32 x 32 Newton system of the home section by a SYMMETRIC block elimination whose rank-2 update is ONE
// v_mfma_f32_32x32x2_f32 per pivot pair (gjm_solve_f32, mxe_kernel.hip.h), against the two-pivot Gauss-Jordan elimination on the
// vector pipe it replaces (gj2_solve64_f32): the same solutions (binary32, against a binary64 solve on the host), cycles per solve
// of a lone wave and of two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gj_mfma.hip -o tools/gj_mfma && tools/gj_mfma
#include "../maxent_amd/csrc/mxe_kernel.hip.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

using namespace mxe;

template <int N>
__global__ void k_old(const float* __restrict__ M, const float* __restrict__ rhs, float* __restrict__ z, long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sys = blockIdx.x * (blockDim.x / 64) + wave;
    const int i = lane & 31, h = lane >> 5;
    const float* Ms = M + (size_t)sys * 32 * 32;
    float zz = 0.0f;
    long long t = 0;
    for (int r = 0; r < reps; ++r) {
        float A[N / 2];
        for (int kk = 0; kk < N / 2; ++kk) A[kk] = (i < N) ? Ms[i * 32 + 2 * kk + h] : ((2 * kk + h) == i ? 1.0f : 0.0f);
        float b = (i < N) ? rhs[sys * 32 + i] + 1e-9f * r : 0.0f;
        bool small;
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" : "+v"(b), "+v"(A[0]), "+v"(A[N / 2 - 1]));
        const long long t0 = clock64();
        asm volatile("" : "+v"(b) : "s"(t0));
        gj2_solve64_f32<N>(A, b, i, zz, small);
        asm volatile("" : "+v"(zz));
        const long long t1 = clock64();
        asm volatile("" : "+v"(zz) : "s"(t1));
        t += t1 - t0;
    }
    if (h == 0 && i < N) z[sys * 32 + i] = zz;
    if (lane == 0) cyc[sys] = t / reps;
}

template <int N>
__global__ void k_new(const float* __restrict__ M, const float* __restrict__ rhs, float* __restrict__ z, long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sys = blockIdx.x * (blockDim.x / 64) + wave;
    const int n = lane & 31, h = lane >> 5;
    const float* Ms = M + (size_t)sys * 32 * 32;
    float zz = 0.0f;
    long long t = 0;
    for (int r = 0; r < reps; ++r) {
        floatx16 D;
        for (int v = 0; v < 16; ++v) {
            const int row = 8 * (v / 4) + 4 * h + (v % 4);
            D[v] = (row < N && n < N) ? Ms[row * 32 + n] : (row == n ? 1.0f : 0.0f);
        }
        float b = (n < N) ? rhs[sys * 32 + n] + 1e-9f * r : 0.0f;
        bool small;
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" : "+v"(b));
        const long long t0 = clock64();
        asm volatile("" : "+v"(b) : "s"(t0));
        gjm_solve_f32<N>(D, b, N, zz, small);
        asm volatile("" : "+v"(zz));
        const long long t1 = clock64();
        asm volatile("" : "+v"(zz) : "s"(t1));
        t += t1 - t0;
    }
    if (h == 0 && n < N) z[sys * 32 + n] = zz;
    if (lane == 0) cyc[sys] = t / reps;
}

__global__ void k_probe(int* bad)
{
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    floatx16 D;
    for (int v = 0; v < 16; ++v) D[v] = 0.0f;
    const float a = (h == 0) ? (float)(n + 1) : 0.5f * (n + 1);         // A[i][0] = i + 1, A[i][1] = (i + 1) / 2
    const float b = (h == 0) ? (float)(100 + n) : (float)(7 * n);       // B[0][n] = 100 + n, B[1][n] = 7 n
    D = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, D, 0, 0, 0);
    int nb = 0;
    for (int v = 0; v < 16; ++v) {
        const int row = 8 * (v / 4) + 4 * h + (v % 4);
        const float want = (row + 1) * (100.0f + n) + 0.5f * (row + 1) * 7.0f * n;
        if (fabsf(D[v] - want) > 1e-3f * fabsf(want) + 1e-3f) ++nb;
    }
    // v_permlane32_swap of two different registers
    const unsigned x = 1000u + lane, y = 2000u + lane;
    const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    // expected: [0] = [x.low | y.low], [1] = [x.up | y.up]
    const unsigned w0 = (h == 0) ? 1000u + lane : 2000u + (lane - 32), w1 = (h == 0) ? 1000u + (lane + 32) : 2000u + lane;
    if ((unsigned)sw[0] != w0) nb += 100;
    if ((unsigned)sw[1] != w1) nb += 10000;
    atomicAdd(bad, nb);
}

int main()
{
    {
        int* dbad; hipMalloc(&dbad, 4); hipMemset(dbad, 0, 4);
        k_probe<<<1, 64>>>(dbad);
        int hb = -1; hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost);
        printf("layout probe: %d (0 = the accumulator layout and the swap are what gjm_solve_f32 assumes)\n", hb);
    }
    const int n_sys = 1024, N = 32;
    std::mt19937 rng(5);
    std::normal_distribution<double> nd;
    std::vector<float> M((size_t)n_sys * 1024), rhs(n_sys * 32);
    std::vector<double> Md((size_t)n_sys * 1024), zd(n_sys * 32);
    for (int s = 0; s < n_sys; ++s) {
        // c W c + a I scaled to a diagonal of O(1): a Gram matrix with a decaying spectrum
        double X[32][48];
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 48; ++k) X[i][k] = nd(rng) * std::exp(-0.25 * i);
        double A[32][32];
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double a = 0; for (int k = 0; k < 48; ++k) a += X[i][k] * X[j][k]; A[i][j] = a + (i == j ? 1e-3 * (1 + s % 7) : 0.0); }
        double sc[32];
        for (int i = 0; i < 32; ++i) sc[i] = 1.0 / std::sqrt(A[i][i]);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { Md[(size_t)s * 1024 + i * 32 + j] = A[i][j] * sc[i] * sc[j]; M[(size_t)s * 1024 + i * 32 + j] = (float)Md[(size_t)s * 1024 + i * 32 + j]; }
        for (int i = 0; i < 32; ++i) rhs[s * 32 + i] = (float)nd(rng);
        // binary64 solve (Gaussian elimination, SPD)
        double B[32][33];
        for (int i = 0; i < 32; ++i) { for (int j = 0; j < 32; ++j) B[i][j] = Md[(size_t)s * 1024 + i * 32 + j]; B[i][32] = rhs[s * 32 + i]; }
        for (int k = 0; k < 32; ++k) { for (int i = 0; i < 32; ++i) if (i != k) { const double f = B[i][k] / B[k][k]; for (int j = k; j < 33; ++j) B[i][j] -= f * B[k][j]; } }
        for (int i = 0; i < 32; ++i) zd[s * 32 + i] = B[i][32] / B[i][i];
    }
    float *dM, *dr, *dz; long long* dc;
    hipMalloc(&dM, M.size() * 4); hipMalloc(&dr, rhs.size() * 4); hipMalloc(&dz, rhs.size() * 4); hipMalloc(&dc, n_sys * 8);
    hipMemcpy(dM, M.data(), M.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, rhs.data(), rhs.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> z(n_sys * 32); std::vector<long long> c(n_sys);
    auto report = [&](const char* tag) {
        hipDeviceSynchronize();
        hipMemcpy(z.data(), dz, z.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n_sys * 8, hipMemcpyDeviceToHost);
        double worst = 0, mean = 0; long long cy = 0;
        for (int s = 0; s < n_sys; ++s) {
            double num = 0, den = 0;
            for (int i = 0; i < 32; ++i) { const double d = z[s * 32 + i] - zd[s * 32 + i]; num += d * d; den += zd[s * 32 + i] * zd[s * 32 + i]; }
            const double e = std::sqrt(num / den); worst = std::max(worst, e); mean += e / n_sys; cy += c[s];
        }
        printf("%-44s rel. error of z against binary64: mean %.2e worst %.2e; %lld cycles per solve\n", tag, mean, worst, cy / n_sys);
    };
    for (int wpb : {1, 4, 8}) {          // waves per workgroup (one workgroup per CU when the grid is small): 1 = a lone wave per SIMD ...
        hipMemset(dz, 0, z.size() * 4);
        k_old<N><<<n_sys / wpb, 64 * wpb>>>(dM, dr, dz, dc, 20);
        char tag[96]; snprintf(tag, sizeof tag, "vector pipe (gj2_solve64_f32), %d waves per block", wpb); report(tag);
        hipMemset(dz, 0, z.size() * 4);
        k_new<N><<<n_sys / wpb, 64 * wpb>>>(dM, dr, dz, dc, 20);
        snprintf(tag, sizeof tag, "one MFMA per pivot pair (gjm_solve_f32), %d", wpb); report(tag);
    }
    return 0;
}
