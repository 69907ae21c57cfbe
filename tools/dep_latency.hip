// micro-benchmark: what a DEPENDENT instruction costs a lone wave on gfx950 (one wave per SIMD, the regime of the serial sections:
// the Gauss-Jordan solves of the home wave and of the one-chain kernel).  Cycles (s_memtime) per instruction of a chain in which
// every instruction needs the result of the one before it, next to the same instruction in eight independent streams.
//   hipcc --offload-arch=gfx950 -O3 tools/dep_latency.hip -o tools/dep_latency && tools/dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ void k_lat(double* out, long long* t, int mode, int n) {
    __shared__ double sh[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x, c = 1e-7 * threadIdx.x;
    double f[8];
    for (int q = 0; q < 8; ++q) f[q] = a + q;
    float fa = (float)a, fb = (float)b, fc = (float)c;
    sh[threadIdx.x] = a; sh[threadIdx.x + 256] = b;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        switch (mode) {
        case 0:   // dependent v_fma_f64
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
            break;
        case 1:   // 8 independent streams of v_fma_f64 (16 instructions)
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(f[0]), "+v"(f[1]) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(f[2]), "+v"(f[3]) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(f[4]), "+v"(f[5]) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(f[6]), "+v"(f[7]) : "v"(b), "v"(c));)
            break;
        case 2:   // dependent v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));)
            break;
        case 3:   // dependent v_rcp_f64
            REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
            break;
        case 4: { // v_readlane_b32 x2 -> v_fma_f64 with the SGPR pair as operand -> next readlane reads the result (dependent)
            REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n s_nop 0\n v_fma_f64 %2, %2, s[20:21], %3" : : "v"(__double2loint(b)), "v"(__double2hiint(b)), "v"(a), "v"(c) : "s20", "s21");)
            break; }
        case 5: { // 16 independent readlanes (b32)
            asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s23, %0, 6\n"
                         "v_readlane_b32 s24, %0, 7\n v_readlane_b32 s25, %0, 8\n v_readlane_b32 s26, %0, 9\n v_readlane_b32 s27, %0, 10\n"
                         "v_readlane_b32 s28, %0, 11\n v_readlane_b32 s29, %0, 12\n v_readlane_b32 s30, %0, 13\n v_readlane_b32 s31, %0, 14\n"
                         "v_readlane_b32 s32, %0, 15\n v_readlane_b32 s33, %0, 16\n v_readlane_b32 s34, %0, 17\n v_readlane_b32 s35, %0, 18"
                         : : "v"(lane) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
            break; }
        case 6:   // LDS round trip: ds_write_b64 -> wait -> ds_read_b64 of another lane's slot -> wait (dependent)
            REP16(asm volatile("ds_write_b64 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(8 * threadIdx.x), "v"(8 * (threadIdx.x ^ 1)) : "memory");)
            break;
        case 7:   // s_barrier, every wave arrives at once
            REP16(asm volatile("s_barrier" ::: "memory");)
            break;
        case 8:   // dependent ds_swizzle (crossbar, no memory)
            REP16(asm volatile("ds_swizzle_b32 %0, %0 offset:0x041f\n s_waitcnt lgkmcnt(0)" : "+v"(fa));)
            break;
        case 9:   // dependent v_mov_b32 dpp (row_shr:1)
            REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(fa));)
            break;
        case 10:  // wave 0 writes LDS, barrier, all read: the multiplier hand-over of gj_solve_4w (per step)
            REP16(if (wave == 0) asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:8" : : "v"(16 * lane), "v"(f[0]), "v"(f[1]) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier\n ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(16 * lane) : "memory");)
            break;
        case 11:  // v_cndmask chain (f64 select = 2 x b32)
            REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(fa) : "v"(fb) : );)
            break;
        }
    }
    const long long t1 = clock64();
    double s = a + b + c + fa + fb + fc;
    for (int q = 0; q < 8; ++q) s += f[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + sh[(threadIdx.x * 7) & 1023];
    if (lane == 0) { t[2 * wave] = t0; t[2 * wave + 1] = t1; }
}

int main() {
    double* out; long long* t;
    hipMalloc(&out, 1 << 20); hipMalloc(&t, 4096);
    long long h[64];
    const int n = 500;
    const char* names[12] = {"v_fma_f64, dependent chain", "v_fma_f64, 8 independent streams", "v_fma_f32, dependent chain", "v_rcp_f64, dependent chain",
                             "2 x v_readlane_b32 + v_fma_f64 on the SGPR pair (dependent), per triple", "v_readlane_b32, 16 independent (distinct SGPRs)",
                             "ds_write_b64 + wait + ds_read_b64 + wait (dependent), per round trip", "s_barrier (all waves there)",
                             "ds_swizzle_b32 + wait (dependent)", "v_mov_b32 dpp row_shr (dependent, s_nop 1)",
                             "wave 0: ds_write_b128; all: wait + s_barrier + ds_read_b64 + wait, per hand-over", "v_cndmask_b32, dependent chain"};
    for (int nw = 1; nw <= 4; nw *= 4)
        for (int mode = 0; mode < 12; ++mode) {
            if (nw == 1 && (mode == 7 || mode == 10)) continue;
            hipLaunchKernelGGL(k_lat, dim3(1), dim3(64 * nw), 0, 0, out, t, mode, 10);
            hipLaunchKernelGGL(k_lat, dim3(1), dim3(64 * nw), 0, 0, out, t, mode, n);
            hipDeviceSynchronize();
            hipMemcpy(h, t, 16 * nw, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int w = 0; w < nw; ++w) worst = (double)(h[2 * w + 1] - h[2 * w]) > worst ? (double)(h[2 * w + 1] - h[2 * w]) : worst;
            const double per = worst / (16.0 * n) / (mode == 1 ? 8.0 : 1.0);
            printf("%d wave(s) on a CU  %-90s %8.1f s_memtime ticks\n", nw, names[mode], per);
        }
    // the tick of s_memtime against the shader clock: a known 4-pass instruction stream
    return 0;
}
