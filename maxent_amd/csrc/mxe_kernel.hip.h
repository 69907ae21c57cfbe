// mxe_kernel.hip.h -- the alpha-chain solver kernel (gfx950, wave64, fp64)
//
// One workgroup of NW wavefronts solves one chain = the warm-started alpha
// scan of one matrix element (reference: maxent_loop.py:241-245 around
// levenberg_minimizer.py:123-248).  All state of the chain lives in LDS and
// registers; the shared singular basis V is streamed from L2 (it is read by
// every chain and never leaves the XCD's L2 once warm), the element's D and
// ghat come from HBM once, the per-alpha results go to HBM once.
//
// Mathematics (whitened singular basis, see DESIGN.md and oracle/sform.py):
//   u = V v,  H = D e^u | D(e^u - e^-u),  w = H | D(e^u + e^-u)
//   h = V^T H,  rho = c*h - ghat,  chi2 = |rho|^2 + c_perp
//   S = sum(H - D - H u) | sum(H+ - D - H+ u) + sum(H- - D + H- u)
//   g = c*rho + alpha v,  W = V^T diag(w) V
//   Newton step (Bryan):  (c W c + (alpha+mu) I) z = rho + alpha v / c,
//   delta = c*z, accepted when delta^T W delta <= step_max * sum(D) and the
//   trial point is finite; converged when r = |w * V delta| / |H| < tol_h, or,
//   after a full step, when the estimated next correction expm1(max|du|) r is.
//
// Active subspace: the singular weights c_k decay exponentially.  For the
// directions with c_k^2 max(w) <= theta (alpha+mu) the Newton matrix is
// (alpha+mu) I to relative accuracy theta, so only the leading n_act x n_act
// block of W is assembled and factorised; the remaining components take the
// diagonal step z_k = rhs_k/(alpha+mu).  This changes the Newton matrix (an
// inexact Newton method with contraction ~theta), never the residual g whose
// zero defines the answer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace mxe {

constexpr int WAVE = 64;
constexpr int GRAM_R = 4;         // rows staged per wave per Gram tile
constexpr int GBLK = 6;           // staged doubles per 4-column block (16-B aligned, bank spread)

struct KParams {
    int n_omega;        // number of frequencies
    int n_omega_pad;    // multiple of 64
    int n_s;            // kept singular values (<= NP)
    int NP;             // padded n_s (row stride of V): 64 or 128
    int n_alpha;
    int n_chain;
    // basis arrays, indexed by data-set id
    const double* V;     // [n_ds][n_omega_pad][NP]  zero padded rows and columns
    const double* Vt;    // [n_ds][NP][n_omega_pad]  zero padded
    const double* Vx;    // V with the columns of every row interleaved for 16-byte loads in the lock-step fused pass (NP = 64):
                         //   position 32 t' + 2 m + s holds column 16 (2 t' + s) + m
    const double* c;     // [n_ds][NP]               descending, padded with 1
    const double* cinv;  // [n_ds][NP]
    // elements
    const int* elem_ds;     // [n_elem]
    const int* elem_kind;   // [n_elem]
    const double* ghat;     // [n_elem][NP]
    const double* cperp;    // [n_elem]
    const double* D;        // [n_elem][n_omega_pad]  zero padded
    const double* sumD;     // [n_elem]  sum(D) (x2 for plusminus)
    // chains
    const int* chain_elem;  // [n_chain]
    const int* chain_prob0; // [n_chain] first problem (index into alpha[] and the outputs)
    const int* chain_len;   // [n_chain] number of alphas of the (sub-)chain
    const int* chain_v0;    // [n_chain] row of v0[] to start from
    // (lock-step kernel) the state of a piece after the evaluation of its start vector, tabulated per class of
    // pieces that start from the same (v0, default model, entropy): the Gram tiles in accumulator layout times
    // sc2 [NPAIR * 256], h = V^T H [NP], then S, sum H^2, max w, sc2 -- the first round of a piece is a Newton step
    const double* init_tab;     // [n_class][MC_INIT_STRIDE] or nullptr
    const int* chain_init;      // [n_chain] class of the piece, -1: none
    const int* chain_lead;  // [n_chain] or nullptr: solve alpha[chain_prob0 - chain_lead] BEFORE the piece's first one (no record; 0 = none); lock-step kernel only
    // (lock-step kernels) a led piece on a mesh too coarse to walk on walks a ladder of its own: chain_lead[c] alphas from
    // walk_alpha[chain_walk0[c]] on (the first is its leading alpha); chain_walk0[c] < 0 or nullptr: the scan's mesh
    const int* chain_walk0;     // [n_chain] or nullptr
    const double* walk_alpha;   // or nullptr
    const double* alpha;    // [P]
    const double* v0;       // [n_parent][NP]   whitened basis
    // outputs, problem p = chain_prob0[chain] + i
    double* out_v;      // [P][NP]  whitened basis
    double* out_H;      // [P][n_omega]
    double* out_chi2;   // [P]
    double* out_S;
    double* out_Q;
    int* out_niter;
    int* out_conv;
    int* out_nevals;
    int* out_nact;      // [P] size of the active block at the last iteration
    // options
    int maxiter, miniter;
    double tol_h, tol_d, tol_relq, step_max, mu_first, mu_grow, mu_max, theta;
    int stop_estimate;  // tol_h also applies to the estimated next correction (see mxe_opts)
    long long* prof;    // [n_chain][8] phase cycle counters (diagnostic build only)
    // binary32 copies of V / Vt (same shapes) for the fp32 streaming variant (mxe_opts.precision)
    const float* Vf;
    const float* Vtf;
    // (one-chain kernel, GST build) the omega-space state of every chain -- u, trial u, w, trial w, H: 5 x n_omega_pad
    // of the stream type each -- in device memory, for frequency meshes whose state does not fit the 160 KB of LDS
    void* gstate;
    // (lock-step kernel) an alpha is given up after this many iterations: its Gram matrices are binary16 products
    // (21 bits), which stalls the iteration where the Newton matrix is very ill conditioned; mxe_chains_finish hands
    // those alphas to the one-chain kernel (binary64 Gram matrix)
    int mc_maxiter;
    int mc_abandon;      // lock-step kernel: an alpha it gives up on ends its piece (the rest is left to mxe_chains_finish)
    // (one-chain kernel) iterations problem p may still spend, or nullptr: maxiter for every alpha.  mxe_chains_finish sets it:
    // the caller's maxiter caps the iterations of ONE alpha over both passes (levenberg_minimizer.py:155 caps per alpha)
    const int* prob_maxiter;    // [P]
    // (one-chain kernel, the finishing pass on a coarse mesh) the entries of a chain may be RUNGS between two alphas of the mesh:
    // entry e = chain_prob0 + i writes its record to problem out_index[e], or nowhere when that is < 0 (a rung: a few iterations
    // towards the next alpha of the mesh, whose record then counts the rung's iterations and evaluations as its own); nullptr: e
    const int* out_index;       // [entries] or nullptr
    int mc_maxevals;            // lock-step kernels: evaluations an alpha may cost before it is handed over like one that ran out of damping
    int stuck_skip;             // an alpha held at one damping stops trying the smaller ones at every iteration (the damping loop of chain_kernel)
    double* dbg_hist;           // (diagnostic build -DMXE_DEBUG_HIST) [entries][16]: the stopping quantity of an alpha at fixed iteration counts
};

#if defined(MXE_PROFILE) && defined(MXE_PROFILE_EVAL)
#define MXE_STAMP_E(idx) do { const long long t__ = clock64(); if (tid == 0) prof_acc[idx] += t__ - prof_t; prof_t = t__; } while (0)
#else
#define MXE_STAMP_E(idx) do {} while (0)
#endif
#if defined(MXE_PROFILE) && defined(MXE_PROFILE_GJ)
// (diagnostic build of the four-wave elimination alone: the ordinary stamps only move the clock)
#define MXE_STAMP(idx) do { prof_t = clock64(); } while (0)
#define MXE_STAMP_G(idx) do { const long long t__ = clock64(); if (tid == 0) prof_acc[idx] += t__ - prof_t; prof_t = t__; } while (0)
#elif defined(MXE_PROFILE)
#define MXE_STAMP(idx) do { const long long t__ = clock64(); if (tid == 0) prof_acc[idx] += t__ - prof_t; prof_t = t__; } while (0)
#define MXE_STAMP_G(idx) do {} while (0)
#else
#define MXE_STAMP(idx) do {} while (0)
#define MXE_STAMP_G(idx) do {} while (0)
#endif

__device__ __forceinline__ void wave_sync() {
    // orders LDS traffic between the lanes of one wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exp(-u) from e = exp(u): v_rcp_f64 and two Newton steps (full precision; an
// overflowed or underflowed e gives NaN / inf, which the callers treat like any
// other non-finite trial point)
__device__ __forceinline__ double recip_exp(double e) {
    double y = __builtin_amdgcn_rcp(e);
    y = fma(fma(-e, y, 1.0), y, y);
    y = fma(fma(-e, y, 1.0), y, y);
    return y;
}

// exp(x) in binary64 for the row pass of the lock-step kernel: k = rint(x log2 e), r = x - k ln 2 (two
// steps), Taylor polynomial of degree 12 in Horner form as plain 3-operand FMAs, result ldexp(p, k).
// |r| <= 0.347: truncation 1.7e-16 relative; overflow / underflow / NaN come out of ldexp and the
// arithmetic as inf / 0 / NaN, which is what the callers test for.  (The library exp spends a register
// move per coefficient and five instructions on range checks: 35 instructions against 19.)
__device__ __forceinline__ double fast_exp(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    // k beyond the int range of ldexp only for |x| > 1e9: clamp (the result is inf / 0 either way)
    const int ki = (int)fmin(fmax(k, -4000.0), 4000.0);
    return ldexp(p, ki);
}

__device__ __forceinline__ float recip_exp(float e) {
    float y = __builtin_amdgcn_rcpf(e);
    y = fmaf(fmaf(-e, y, 1.0f), y, y);
    return y;
}

// stream type of the omega-space arithmetic: double (default) or float (mxe_opts.precision = F32:
// V, u, w, H, exp, the two mat-vecs and the Gram matrix in binary32; the n_act x n_act Newton
// system, the residual rho and every scalar of the iteration stay binary64)
template <typename TS> struct Stream;
template <> struct Stream<double> {
    typedef double2 vec2;
    typedef double acc4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ double exp_(double x) { return exp(x); }
    static __device__ __forceinline__ const double* V(const KParams& p) { return p.V; }
    static __device__ __forceinline__ const double* Vt(const KParams& p) { return p.Vt; }
    static __device__ __forceinline__ acc4 mfma(double a, double b, acc4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = l & 15, row = (l >> 4) + 4 r
    static __device__ __forceinline__ int crow(int kq, int r) { return kq + 4 * r; }
};
template <> struct Stream<float> {
    typedef float2 vec2;
    typedef float acc4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float exp_(float x) { return expf(x); }
    static __device__ __forceinline__ const float* V(const KParams& p) { return p.Vf; }
    static __device__ __forceinline__ const float* Vt(const KParams& p) { return p.Vtf; }
    static __device__ __forceinline__ acc4 mfma(float a, float b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f32_16x16x4_f32: col = l & 15, row = 4 (l >> 4) + r
    static __device__ __forceinline__ int crow(int kq, int r) { return 4 * kq + r; }
};

// x[l] + x[l ^ 32] and x[l] + x[l ^ 16] in every lane with the gfx950 lane-swap
// instructions (v_permlane32_swap / v_permlane16_swap exchange half-waves / odd and
// even rows of 16 lanes between two registers): no LDS crossbar round trip
__device__ __forceinline__ double sum_xor32(double x) {
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double sum_xor16(double x) {
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

__device__ __forceinline__ double max_xor32(double x) {
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
}
__device__ __forceinline__ double max_xor16(double x) {
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
}

// cross-lane move of a double inside every row of 16 lanes (two v_mov_b32_dpp)
template <int CTRL> __device__ __forceinline__ double dpp_row(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // i <-> 7 - i   inside 8 lanes
constexpr int DPP_MIRROR = 0x140;       // i <-> 15 - i  inside 16 lanes

constexpr int DPP_ROR4 = 0x124;         // rotate right by 4 inside 16 lanes
constexpr int DPP_ROR8 = 0x128;

// sum / max over the 16 lanes of equal (lane & 3), result in all of them
__device__ __forceinline__ double slot_sum(double x) {
    x += dpp_row<DPP_ROR4>(x);
    x += dpp_row<DPP_ROR8>(x);
    return sum_xor32(sum_xor16(x));
}
__device__ __forceinline__ double slot_max(double x) {
    x = fmax(x, dpp_row<DPP_ROR4>(x));
    x = fmax(x, dpp_row<DPP_ROR8>(x));
    return max_xor32(max_xor16(x));
}

// wave-wide sum / max, result in every lane: four DPP steps inside the rows of 16
// lanes, then the four row results through v_readlane (no LDS crossbar round trips)
__device__ __forceinline__ double wave_sum(double x) {
    x += dpp_row<DPP_XOR1>(x);
    x += dpp_row<DPP_XOR2>(x);
    x += dpp_row<DPP_HALF_MIRROR>(x);
    x += dpp_row<DPP_MIRROR>(x);
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double wave_max(double x) {
    x = fmax(x, dpp_row<DPP_XOR1>(x));
    x = fmax(x, dpp_row<DPP_XOR2>(x));
    x = fmax(x, dpp_row<DPP_HALF_MIRROR>(x));
    x = fmax(x, dpp_row<DPP_MIRROR>(x));
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return fmax(fmax(r0, r1), fmax(r2, r3));
}

// broadcast lane `src` (wave-uniform index) of x to all lanes via v_readlane
__device__ __forceinline__ double wave_bcast(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

template <int NW> __device__ __forceinline__ void block_sync() {
    if (NW == 1) wave_sync(); else __syncthreads();
}

// f(integral_constant<int, J>) for every J of the sequence, in order
template <int... J, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, J...>, F&& f) {
    (f(std::integral_constant<int, J>{}), ...);
}

// lane J of the lane's own group of 32 (ds_swizzle_b32, bit-mask mode: and 0, or J, xor 0 -- the LDS crossbar, no memory)
template <int J> __device__ __forceinline__ double half_bcast(double x) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), J << 5);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), J << 5);
    return __hiloint2double(hi, lo);
}

// Gauss-Jordan elimination of an N x N positive definite system (N <= 32, even) on the 64 lanes of one wavefront,
// two pivots per step (the solve of the lock-step kernel, mxe_kernel_mc.hip.h: gj_home, explains the layout): lane
// (h = lane >> 5, i = lane & 31) holds of row i the columns of parity h -- A[kk] = column 2 kk + h -- and the
// right-hand side b_i (both halves carry it).  Returns false where a pivot block is not positive definite; z = the
// lane's solution component.
template <int N>
__device__ __forceinline__ bool gj2_solve64(double (&A)[N / 2], double b, int i, double& z)
{
    constexpr int NHALF = N / 2;
    bool ok = true;
    double ps = 1.0, pc = 0.0;                 // z_i = ps b_i + pc b_(i ^ 1)
    auto pivot2 = [&](auto KTag) {
        constexpr int kj = decltype(KTag)::value, j = 2 * kj;
        double c0, c1;
        {
            const unsigned xlo = (unsigned)__double2loint(A[kj]), xhi = (unsigned)__double2hiint(A[kj]);
            const auto slo = __builtin_amdgcn_permlane32_swap(xlo, xlo, false, false);
            const auto shi = __builtin_amdgcn_permlane32_swap(xhi, xhi, false, false);
            c0 = __hiloint2double((int)shi[0], (int)slo[0]);      // column j (the lower half's values in both halves)
            c1 = __hiloint2double((int)shi[1], (int)slo[1]);      // column j + 1
        }
        const double pa = wave_bcast(c0, j), pb = wave_bcast(c1, j), pd = wave_bcast(c1, j + 1);
        const double det = fma(pa, pd, -pb * pb);
        if (!(pa > 0.0) || !(det > 0.0)) ok = false;
        double inv = __builtin_amdgcn_rcp(det);
        inv = fma(fma(-det, inv, 1.0), inv, inv);
        const double qa = pa * inv, qb = pb * inv, qd = pd * inv;       // P^-1 = [[qd, -qb], [-qb, qa]]
        const bool prow = (i >> 1) == kj;
        if (prow) { ps = (i & 1) ? qa : qd; pc = -qb; }
        const double f0 = prow ? 0.0 : fma(c0, qd, -c1 * qb);
        const double f1 = prow ? 0.0 : fma(c1, qa, -c0 * qb);
        {
            const double b0 = wave_bcast(b, j), b1 = wave_bcast(b, j + 1);
            b = fma(-f1, b1, fma(-f0, b0, b));
        }
#pragma unroll
        for (int k0 = kj + 1; k0 < NHALF; k0 += 4) {
            double r0[4], r1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) if (k0 + r < NHALF) { r0[r] = half_bcast<j>(A[k0 + r]); r1[r] = half_bcast<j + 1>(A[k0 + r]); }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (k0 + r < NHALF) A[k0 + r] = fma(-f1, r1[r], fma(-f0, r0[r], A[k0 + r]));
        }
    };
    static_for_seq(std::make_integer_sequence<int, NHALF>{}, pivot2);
    const int plo = __builtin_amdgcn_ds_swizzle(__double2loint(b), (1 << 10) | 0x1f);     // lane ^ 1
    const int phi = __builtin_amdgcn_ds_swizzle(__double2hiint(b), (1 << 10) | 0x1f);
    z = fma(ps, b, pc * __hiloint2double(phi, plo));
    return ok;
}

// The same elimination in binary32: half the cross-lane traffic (one dword per value instead of two) and full-rate
// multiply-adds.  For the lock-step kernel, whose Newton matrix is inexact at the 1e-6 level anyway (Gram tiles from split
// binary16 products, 21 bits; decoupling threshold 1e-5): the backward error of this solve, ~N eps_32 |A| = 2e-6 |A|, is of
// the same size.  The caller scales rows and columns by powers of two so that the diagonal is O(1) (the entries span
// twenty decades otherwise and det P would leave the binary32 range).
template <int J> __device__ __forceinline__ float half_bcast_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), J << 5));
}
__device__ __forceinline__ float wave_bcast_f(float x, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src));
}
#ifndef MXE_X_GJ_LOOKAHEAD
#define MXE_X_GJ_LOOKAHEAD 1
#endif
#if MXE_X_GJ_LOOKAHEAD
// Round 5: the same elimination with a SHORTER dependent chain per step and the chain of step j + 1 started before the trailing
// update of step j is through (look-ahead).  A step used to be: swap -> three broadcasts -> determinant (2) -> reciprocal ->
// two Newton refinements -> P^-1 (1) -> multipliers (2) -> update (2): twelve dependent instructions at 20-26 cycles each on a
// lone wave (tools/dep_latency.hip), behind which the 2 (N/2 - j - 1) swizzles and as many multiply-adds of the trailing update
// queued: ~450-520 cycles per step, 8.3 k of the 52 k cycles of a round (profiles/r03_g_serial_section.txt).  Now: the multipliers
// as (numerator) x (1 / det) with the numerators formed beside the reciprocal, no refinement of v_rcp_f32 (1 ulp: the solve only
// preconditions the step) -- eight dependent instructions --, and the order of the code is: update the NEXT pivot columns first,
// start the next step's chain (swap, broadcasts, determinant, reciprocal, multipliers, right-hand side), then the rest of this
// step's trailing update with the multipliers kept from before.
template <int N>
__device__ __forceinline__ bool gj2_solve64_f32(float (&A)[N / 2], float b, int i, float& z, bool& small_pivot)
{
    constexpr int NHALF = N / 2;
#ifndef MXE_X_PIV_TAU
#define MXE_X_PIV_TAU 1e-3f
#endif
    constexpr float PIV_TAU = MXE_X_PIV_TAU;    // pivots below this (the diagonal was scaled to [1, 4)): 24 bits are too few
    bool ok = true;
    small_pivot = false;
    float ps = 1.0f, pc = 0.0f;                 // z_i = ps b_i + pc b_(i ^ 1)
    float f0 = 0.0f, f1 = 0.0f;                 // multipliers of the step whose trailing update is being applied
    float g0 = 0.0f, g1 = 0.0f;                 // ... of the step whose chain has been started (look-ahead)
    // the chain of step kj: needs the columns 2 kj, 2 kj + 1 (register A[kj]) and b as the steps before left them
    auto head = [&](auto KTag, float& m0, float& m1) {
        constexpr int kj = decltype(KTag)::value, j = 2 * kj;
        const unsigned x = __builtin_bit_cast(unsigned, A[kj]);
        const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        const float c0 = __builtin_bit_cast(float, (unsigned)sw[0]);      // column j (the lower half's values in both halves)
        const float c1 = __builtin_bit_cast(float, (unsigned)sw[1]);      // column j + 1
        const float pa = wave_bcast_f(c0, j), pb = wave_bcast_f(c1, j), pd = wave_bcast_f(c1, j + 1);
        const float det = __builtin_fmaf(pa, pd, -pb * pb);
        if (!(pa > 0.0f) || !(det > 0.0f)) ok = false;
        if (pa < PIV_TAU || det < PIV_TAU * pa) small_pivot = true;      // (pivots: pa and det / pa)
        const float inv = __builtin_amdgcn_rcpf(det);
        // (f0, f1) = (c0, c1) P^-1,  P^-1 = [[pd, -pb], [-pb, pa]] / det: the numerators do not wait for the reciprocal
        const float n0 = __builtin_fmaf(c0, pd, -c1 * pb), n1 = __builtin_fmaf(c1, pa, -c0 * pb);
        const bool prow = (i >> 1) == kj;
        if (prow) { ps = ((i & 1) ? pa : pd) * inv; pc = -pb * inv; }
        m0 = prow ? 0.0f : n0 * inv;
        m1 = prow ? 0.0f : n1 * inv;
        const float b0 = wave_bcast_f(b, j), b1 = wave_bcast_f(b, j + 1);
        b = __builtin_fmaf(-m1, b1, __builtin_fmaf(-m0, b0, b));
    };
    head(std::integral_constant<int, 0>{}, f0, f1);
    auto step = [&](auto KTag) {
        constexpr int kj = decltype(KTag)::value, j = 2 * kj;
        if constexpr (kj + 1 < NHALF) {
            // the columns of the next pivot block first ...
            {
                const float r0 = half_bcast_f<j>(A[kj + 1]), r1 = half_bcast_f<j + 1>(A[kj + 1]);
                A[kj + 1] = __builtin_fmaf(-f1, r1, __builtin_fmaf(-f0, r0, A[kj + 1]));
            }
            // ... so that its chain runs beside the rest of this step's update
            head(std::integral_constant<int, kj + 1>{}, g0, g1);
#pragma unroll
            for (int k0 = kj + 2; k0 < NHALF; k0 += 8) {
                float r0[8], r1[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) if (k0 + r < NHALF) { r0[r] = half_bcast_f<j>(A[k0 + r]); r1[r] = half_bcast_f<j + 1>(A[k0 + r]); }
#pragma unroll
                for (int r = 0; r < 8; ++r) if (k0 + r < NHALF) A[k0 + r] = __builtin_fmaf(-f1, r1[r], __builtin_fmaf(-f0, r0[r], A[k0 + r]));
            }
            f0 = g0; f1 = g1;
        }
    };
    static_for_seq(std::make_integer_sequence<int, NHALF>{}, step);
    const float bp = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, b), (1 << 10) | 0x1f));     // lane ^ 1
    z = __builtin_fmaf(ps, b, pc * bp);
    return ok;
}
#else
template <int N>
__device__ __forceinline__ bool gj2_solve64_f32(float (&A)[N / 2], float b, int i, float& z, bool& small_pivot)
{
    constexpr int NHALF = N / 2;
#ifndef MXE_X_PIV_TAU
#define MXE_X_PIV_TAU 1e-3f
#endif
    constexpr float PIV_TAU = MXE_X_PIV_TAU;    // pivots below this (the diagonal was scaled to [1, 4)): 24 bits are too few
    bool ok = true;
    small_pivot = false;
    float ps = 1.0f, pc = 0.0f;                 // z_i = ps b_i + pc b_(i ^ 1)
    auto pivot2 = [&](auto KTag) {
        constexpr int kj = decltype(KTag)::value, j = 2 * kj;
        float c0, c1;
        {
            const unsigned x = __builtin_bit_cast(unsigned, A[kj]);
            const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            c0 = __builtin_bit_cast(float, (unsigned)sw[0]);      // column j (the lower half's values in both halves)
            c1 = __builtin_bit_cast(float, (unsigned)sw[1]);      // column j + 1
        }
        const float pa = wave_bcast_f(c0, j), pb = wave_bcast_f(c1, j), pd = wave_bcast_f(c1, j + 1);
        const float det = __builtin_fmaf(pa, pd, -pb * pb);
        if (!(pa > 0.0f) || !(det > 0.0f)) ok = false;
        if (pa < PIV_TAU || det < PIV_TAU * pa) small_pivot = true;      // (pivots: pa and det / pa)
        float inv = __builtin_amdgcn_rcpf(det);
        inv = __builtin_fmaf(__builtin_fmaf(-det, inv, 1.0f), inv, inv);
        const float qa = pa * inv, qb = pb * inv, qd = pd * inv;       // P^-1 = [[qd, -qb], [-qb, qa]]
        const bool prow = (i >> 1) == kj;
        if (prow) { ps = (i & 1) ? qa : qd; pc = -qb; }
        const float f0 = prow ? 0.0f : __builtin_fmaf(c0, qd, -c1 * qb);
        const float f1 = prow ? 0.0f : __builtin_fmaf(c1, qa, -c0 * qb);
        {
            const float b0 = wave_bcast_f(b, j), b1 = wave_bcast_f(b, j + 1);
            b = __builtin_fmaf(-f1, b1, __builtin_fmaf(-f0, b0, b));
        }
#pragma unroll
        for (int k0 = kj + 1; k0 < NHALF; k0 += 8) {
            float r0[8], r1[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) if (k0 + r < NHALF) { r0[r] = half_bcast_f<j>(A[k0 + r]); r1[r] = half_bcast_f<j + 1>(A[k0 + r]); }
#pragma unroll
            for (int r = 0; r < 8; ++r) if (k0 + r < NHALF) A[k0 + r] = __builtin_fmaf(-f1, r1[r], __builtin_fmaf(-f0, r0[r], A[k0 + r]));
        }
    };
    static_for_seq(std::make_integer_sequence<int, NHALF>{}, pivot2);
    const float bp = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, b), (1 << 10) | 0x1f));     // lane ^ 1
    z = __builtin_fmaf(ps, b, pc * bp);
    return ok;
}

#endif
// Round 5: the same system by a SYMMETRIC block elimination whose rank-2 update is ONE matrix instruction per pivot pair.
// The elimination above is bound by the instructions a lone wave can issue (~76 per step at N = 32 -- 28 ds_swizzle and 28
// multiply-adds of the trailing update among them --, 5.5 cycles each: 7.0-7.4 k cycles of the 51 k of a round,
// profiles/r05_d_home_phases.txt).  Here the matrix lives in the accumulator layout of v_mfma_f32_32x32x2_f32 -- lane
// (h = lane >> 5, n = lane & 31) holds column n of the rows 8 (v / 4) + 4 h + (v % 4), v = 0 .. 15 -- and is kept SYMMETRIC
// where it is still read: sweeping the pivot pair J = {j, j + 1} with P = D[J][J], R = D[J][:], G = P^-1 R replaces
//     D[i][n] <- D[i][n] - sum_k R_k[i] G_k[n]        (i, n not in J: the Schur complement, symmetric)
//     D[i][J] <- G[.][i]                              (the swept columns: what Gauss-Jordan keeps as the multipliers of row i)
// and both are the ONE product  D += A B  with  A (32 x 2): lane (k, i) = -R_k[i],  B (2 x 32): lane (k, n) = G_k[n] - [n in J]
// P^-1[k][n - j]  (for n in J the symmetric entry D[i][n] = R_(n-j)[i] then turns into G_(n-j)[i]).  The operands come from the
// two pivot ROWS alone -- two registers of the accumulator, one v_permlane32_swap each way --; no column is ever gathered.  The
// right-hand side rides along in the lanes (lane n of either half: b_n): b_n <- b_n - sum_k B_k[n] b_(j+k), which is P^-1 b_J in the
// pivot lanes.  Swept rows are not maintained (nothing reads them again); after N / 2 steps b is the solution.
// Per step ~26 vector instructions + one MFMA of 16 passes instead of ~76: see tools/gj_mfma.hip for the cycles and the error.
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int N>
__device__ __forceinline__ bool gjm_solve_f32(floatx16& D, float b, int n_act, float& z, bool& small_pivot)
{
    static_assert(N <= 32 && N % 2 == 0, "one 32 x 32 accumulator tile");
    constexpr int NHALF = N / 2;
#ifndef MXE_X_PIV_TAU
#define MXE_X_PIV_TAU 1e-3f
#endif
    constexpr float PIV_TAU = MXE_X_PIV_TAU;
    bool ok = true;
    small_pivot = false;
    int ln = threadIdx.x & 63;
    asm volatile("" : "+v"(ln));                  // (opaque: the lane masks below must not be hoisted out of the caller's loop)
    const int n = ln & 31;
    const bool upper = ln >= 32;
    auto step = [&](auto KTag) {
        constexpr int kj = decltype(KTag)::value, j = 2 * kj;
        if (j < n_act) {                          // (wave-uniform; rows >= n_act are identity rows)
            constexpr int v = 4 * (j / 8) + (j % 4), hj = (j % 8) / 4;       // rows j, j + 1: registers v, v + 1 of half hj
            // Y0 / Y1: R_0[n] / R_1[n] in both halves (two independent swaps of the two accumulator registers with themselves);
            // X: lane (k, n) = R_k[n]
            const float r0 = D[v], r1 = D[v + 1];     // (as floats first: bit_cast of two vector elements in ONE declaration read element v twice)
            unsigned a0 = __builtin_bit_cast(unsigned, r0), a1 = a0, c0 = __builtin_bit_cast(unsigned, r1), c1 = c0;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1));        // (v_permlane32_swap exchanges its registers in place: copies)
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, a1, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(c0, c1, false, false);
            float Y0 = __builtin_bit_cast(float, (unsigned)s0[hj]), Y1 = __builtin_bit_cast(float, (unsigned)s1[hj]);
            const float X = upper ? Y1 : Y0;
            const float pa = wave_bcast_f(Y0, j), pb = wave_bcast_f(Y0, j + 1), pd = wave_bcast_f(Y1, j + 1);
            const float det = __builtin_fmaf(pa, pd, -pb * pb);
#ifdef GJM_DEBUG
            if (ln == 0) printf("T step %d: v %d hj %d pa %g pb %g pd %g det %g\n", kj, v, hj, pa, pb, pd, det);
#endif
            if (!(pa > 0.0f) || !(det > 0.0f)) ok = false;
            if (pa < PIV_TAU || det < PIV_TAU * pa) small_pivot = true;      // (pivots: pa and det / pa)
            const float ninv = -__builtin_amdgcn_rcpf(det);
            // -(G_k[n] - [n in J] P^-1[k][n - j]) = -(P^-1 (R - [n in J] e_(n-j)))_k: take the unit vectors off R in the pivot lanes.
            // The numerators do not wait for the reciprocal; the operand of the matrix instruction is one multiplication behind it
            Y0 = (n == j) ? Y0 - 1.0f : Y0;
            Y1 = (n == j + 1) ? Y1 - 1.0f : Y1;
            const float n0 = __builtin_fmaf(pd, Y0, -pb * Y1), n1 = __builtin_fmaf(pa, Y1, -pb * Y0);
            const float Bop = (upper ? n1 : n0) * ninv;
            const float bj = wave_bcast_f(b, j), bj1 = wave_bcast_f(b, j + 1);
            b = __builtin_fmaf(__builtin_fmaf(n1, bj1, n0 * bj), ninv, b);
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(X, Bop, D, 0, 0, 0);
        }
    };
    static_for_seq(std::make_integer_sequence<int, NHALF>{}, step);
    z = b;
    return ok;
}

// Gauss-Jordan elimination of an N x N positive definite system with MORE than 32 rows (N <= 64) on one wavefront, in
// binary32: lane i holds row i -- all N columns, static indices -- and its right-hand side; one pivot per step, the pivot
// row broadcast column by column with v_readlane (the two-half layout of gj2_solve64 has 32 rows).  Rows and columns
// are scaled by the caller to a diagonal of O(1), as there.  N (N - 1) / 2 multiply-adds and as many broadcasts per
// lane: 64 rows cost four times what 32 rows cost in the two-half layout -- and a sixth of the Cholesky factorisation in
// LDS that systems of this size go through in the one-chain kernel.  Steps j >= n_live are identity rows and skipped.
template <int N>
__device__ __forceinline__ bool gj1_solve_rows_f32(float (&A)[N], float b, int i, int n_live, float& z, bool& small_pivot)
{
#ifndef MXE_X_PIV_TAU
#define MXE_X_PIV_TAU 1e-3f
#endif
    constexpr float PIV_TAU = MXE_X_PIV_TAU;
    bool ok = true;
    small_pivot = false;
    float d = 1.0f;                              // the lane's own pivot
    auto step = [&](auto JTag) {
        constexpr int j = decltype(JTag)::value;
        if (j < n_live) {                        // (wave-uniform)
            const float pjj = wave_bcast_f(A[j], j);
            if (!(pjj > 0.0f)) ok = false;
            if (pjj < PIV_TAU) small_pivot = true;
            float inv = __builtin_amdgcn_rcpf(pjj);
            inv = __builtin_fmaf(__builtin_fmaf(-pjj, inv, 1.0f), inv, inv);
            const bool own = (i == j);
            if (own) d = pjj;
            const float m = own ? 0.0f : A[j] * inv;             // this row's multiplier (the pivot row stays)
            b = __builtin_fmaf(-m, wave_bcast_f(b, j), b);
#pragma unroll
            for (int k0 = j + 1; k0 < N; k0 += 8) {
                float r[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k0 + e < N) r[e] = wave_bcast_f(A[k0 + e], j);
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k0 + e < N) A[k0 + e] = __builtin_fmaf(-m, r[e], A[k0 + e]);
            }
        }
    };
    static_for_seq(std::make_integer_sequence<int, N>{}, step);
    float inv = __builtin_amdgcn_rcpf(d);
    inv = __builtin_fmaf(__builtin_fmaf(-d, inv, 1.0f), inv, inv);
    z = b * inv;
    return ok;
}

// block-wide reduction of NV sums and one max; results valid in every thread.
template <int NW, int NV>
__device__ __forceinline__ void block_reduce(double (&x)[NV], double& mx, double* red /*[NW*(NV+1)]*/) {
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = wave_sum(x[q]);
    mx = wave_max(mx);
    if (NW == 1) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[wave * (NV + 1) + q] = x[q];
        red[wave * (NV + 1) + NV] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) s += red[wv * (NV + 1) + q];
        x[q] = s;
    }
    double m = red[NV];
#pragma unroll
    for (int wv = 1; wv < NW; ++wv) m = fmax(m, red[wv * (NV + 1) + NV]);
    mx = m;
}

// same with two maxima (red: [NW*(NV+2)])
template <int NW, int NV>
__device__ __forceinline__ void block_reduce2(double (&x)[NV], double& mx, double& mx2, double* red) {
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = wave_sum(x[q]);
    mx = wave_max(mx); mx2 = wave_max(mx2);
    if (NW == 1) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[wave * (NV + 2) + q] = x[q];
        red[wave * (NV + 2) + NV] = mx; red[wave * (NV + 2) + NV + 1] = mx2;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) s += red[wv * (NV + 2) + q];
        x[q] = s;
    }
    double m = red[NV], m2 = red[NV + 1];
#pragma unroll
    for (int wv = 1; wv < NW; ++wv) { m = fmax(m, red[wv * (NV + 2) + NV]); m2 = fmax(m2, red[wv * (NV + 2) + NV + 1]); }
    mx = m; mx2 = m2;
}

// NW  wavefronts per chain
// NAB padded singular dimension in units of 32 (2 -> NP = 64, 4 -> NP = 128)
// TS  stream type (double; float = the fp32 variant, NAB = 2 only)
// GST the omega-space state (u, w, H and their trial copies) lives in device memory (KParams::gstate) instead of
//     LDS: any n_omega; the waves exchange it behind __syncthreads (also with one wave per chain: its fences
//     order global memory inside the workgroup)
template <int NW, int NAB, typename TS = double, bool GST = false>
__global__ __launch_bounds__(64 * NW)
void chain_kernel(const KParams p)
{
    constexpr int SYNCW = GST ? 4 : NW;        // (device-memory state: always the workgroup barrier, whose fences order global memory)
    typedef Stream<TS> ST;
    typedef typename ST::vec2 TS2;
    constexpr bool F64 = std::is_same<TS, double>::value;
    static_assert(F64 || NAB == 2, "the fp32 variant is built for n_s <= 64");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = WAVE * NW;
    constexpr int NP = 32 * NAB;          // padded n_s == capacity of the active block
    constexpr int LD = NP + 1;            // row stride of Wm (odd: conflict-free column walks)
    constexpr int RPL = NP / 64;          // rows per lane in the factorisation
    constexpr int SROW = (NP / 4) * GBLK; // staged doubles per Gram row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.x;
    if (chain >= p.n_chain) return;

    const int ns = p.n_s;
    const int nw = p.n_omega, nwp = p.n_omega_pad;

    // ---- LDS carve (all offsets multiples of 2 doubles) ----
    double* Wm   = lds;                      // [NP][LD] upper+diag: W ; strict lower: L
    double* dinv = Wm + NP * LD;             // [NP] 1/L_jj   (NP*LD is even: 16-B alignment kept)
    double* v    = dinv + NP;
    double* dl   = v + NP;
    double* g    = dl + NP;
    double* rhs  = g + NP;
    double* zz   = rhs + NP;
    double* cc   = zz + NP;
    double* ci   = cc + NP;
    double* gh   = ci + NP;
    double* rho  = gh + NP;
    double* rhot = rho + NP;
    // the alpha-path predictor keeps two more vectors; not in the NP = 128 build, whose 132 KB of W
    // leave no room for them
    constexpr bool PRED = (NAB == 2);
    double* ecor = rhot + NP;                // [NP] defect of the first Newton iterate of the previous alpha (predictor)
    double* eacc = ecor + (PRED ? NP : 0);   // [NP] ... of this alpha, being accumulated
    double* hpart = eacc + (PRED ? NP : 0);  // [NW][NP]
    double* red  = hpart + NW * NP;          // [NW*8]
    TS* u    = GST ? reinterpret_cast<TS*>(p.gstate) + (size_t)chain * 5 * nwp
                   : reinterpret_cast<TS*>(red + NW * 8);   // [nwp]
    TS* ut   = u + nwp;
    TS* w    = ut + nwp;
    TS* wt   = w + nwp;
    TS* Hs   = wt + nwp;
    TS* vecs = GST ? reinterpret_cast<TS*>(red + NW * 8) : Hs + nwp;   // [NP] stream-type copy of the vector of a pass
    double* stage = reinterpret_cast<double*>(vecs + NP);   // [NW][2 (x,y)][GRAM_R][SROW] (fp64 VALU Gram only)

    const int elem = p.chain_elem[chain];
    const int ds = p.elem_ds[elem];
    const int kind = p.elem_kind[elem];
    const TS* __restrict__ V  = ST::V(p)  + (size_t)ds * nwp * NP;
    const TS* __restrict__ Vt = ST::Vt(p) + (size_t)ds * NP * nwp;
    const double* __restrict__ Dg = p.D + (size_t)elem * nwp;
    const double cperp = p.cperp[elem];
    const double step_lim = p.step_max * p.sumD[elem];

#ifdef MXE_PROFILE
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
#endif
    for (int k = tid; k < NP; k += T) {
        cc[k] = p.c[ds * NP + k];
        ci[k] = p.cinv[ds * NP + k];
        gh[k] = p.ghat[(size_t)elem * NP + k];
        v[k]  = p.v0[(size_t)p.chain_v0[chain] * NP + k];
        dl[k] = 0.0;
        if (PRED) { ecor[k] = 0.0; eacc[k] = 0.0; }
    }
    block_sync<SYNCW>();

    // ------------------------------------------------------------------
    // evaluation pass: trial u = u - V*vec (from_scratch: u = V*vec).
    // Fills ut, wt, Hs, rhot; returns chi2, S, |w_old o V vec|^2, |H_t|^2,
    // max(w_t), max|V vec|.  Each thread owns two adjacent omega rows (16-B loads of V^T).
    // ------------------------------------------------------------------
    auto eval_pass = [&](const double* vec, bool from_scratch,
                         double& chi2, double& S, double& dH2, double& Hn2, double& wmax, double& dumax) {
        double pS = 0.0, pdH = 0.0, pHn = 0.0, pwm = 0.0, pdu = 0.0;
        const TS* vs;
        if constexpr (F64) vs = vec;
        else {
            for (int k = tid; k < NP; k += T) vecs[k] = (TS)vec[k];
            block_sync<SYNCW>();
            vs = vecs;
        }
        // (rows k >= n_s of V^T are zero and so are the entries of the vector there: the loop runs over whole blocks of 16 rows with
        //  all their loads in flight -- one wave per SIMD and nothing else on the CU, the pass waits for L2, and a remainder loop
        //  would wait once per row)
        const int ns16 = min(NP, (ns + 15) & ~15);
        for (int i = 2 * tid; i < nwp; i += 2 * T) {
            TS a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            const TS* col = Vt + i;
            for (int k = 0; k < ns16; k += 16) {
                TS2 x[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e] = *reinterpret_cast<const TS2*>(col + (size_t)(k + e) * nwp);
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const TS q0 = vs[k + e], q1 = vs[k + e + 1];
                    a0 = fma(x[e].x, q0, a0); b0 = fma(x[e].y, q0, b0);
                    a1 = fma(x[e + 1].x, q1, a1); b1 = fma(x[e + 1].y, q1, b1);
                }
            }
            const TS vd2[2] = {a0 + a1, b0 + b1};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int ii = i + r;
                const TS vd = vd2[r];
                TS ui;
                if (from_scratch) ui = vd;
                else {
                    ui = u[ii] - vd;
                    const TS t = w[ii] * vd;
                    pdH += (double)(t * t);
                    pdu = fmax(pdu, (double)fabs(vd));
                }
                const TS Di = (TS)Dg[ii];
                TS Hi, wi, Si;
                if (kind == 0) {
                    const TS e = ST::exp_(ui);
                    Hi = Di * e; wi = Hi;
                    Si = Hi - Di - Hi * ui;
                } else {
                    const TS ep = ST::exp_(ui), em = recip_exp(ep);
                    const TS Hp = Di * ep, Hm = Di * em;
                    Hi = Hp - Hm; wi = Hp + Hm;
                    Si = (Hp - Di - Hp * ui) + (Hm - Di + Hm * ui);
                }
                if (ii >= nw) { Hi = 0; wi = 0; Si = 0; }
                ut[ii] = ui; wt[ii] = wi; Hs[ii] = Hi;
                pS += (double)Si;
                pHn += (double)(Hi * Hi);
                pwm = fmax(pwm, (double)wi);     // NaN-ignoring; non-finite states are caught through Q
            }
        }
        block_sync<SYNCW>();                    // Hs complete
        MXE_STAMP_E(3);
        // h = V^T H : rows split over the waves; lanes 0-31 take even rows,
        // lanes 32-63 odd rows, each lane two adjacent singular columns
        // (16-B loads); the two half-waves are summed with one shuffle.
        {
            // (the rows of V beyond n_omega are zero and so is H there: every wave takes n_omega_pad / NW rows -- a multiple of 16 --,
            //  its half-waves alternate rows, eight loads of 16 B per lane in flight, no remainder)
            const int rows_per = nwp / NW;
            const int r0 = wave * rows_per;
            const int r1 = r0 + rows_per;
            const int half = lane >> 5, cl = lane & 31;
#pragma unroll
            for (int cb = 0; cb < NP / 64; ++cb) {
                TS s0 = 0, s1 = 0, t0 = 0, t1 = 0;
                const TS* Vc = V + 64 * cb + 2 * cl;
                int i = r0 + half;
                // (sixteen loads in flight where the wave's rows allow it -- n_omega_pad a multiple of 128 --, else eight; the
                //  NP = 128 build has no registers for them)
                if (NAB == 2 && (rows_per & 31) == 0) {
                    for (; i < r1; i += 32) {
                        TS2 x[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) x[e] = *reinterpret_cast<const TS2*>(Vc + (size_t)(i + 2 * e) * NP);
#pragma unroll
                        for (int e = 0; e < 16; e += 2) {
                            const TS h0 = Hs[i + 2 * e], h1 = Hs[i + 2 * e + 2];
                            s0 = fma(x[e].x, h0, s0); s1 = fma(x[e].y, h0, s1);
                            t0 = fma(x[e + 1].x, h1, t0); t1 = fma(x[e + 1].y, h1, t1);
                        }
                    }
                }
                for (; i < r1; i += 16) {
                    TS2 x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = *reinterpret_cast<const TS2*>(Vc + (size_t)(i + 2 * e) * NP);
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const TS h0 = Hs[i + 2 * e], h1 = Hs[i + 2 * e + 2];
                        s0 = fma(x[e].x, h0, s0); s1 = fma(x[e].y, h0, s1);
                        t0 = fma(x[e + 1].x, h1, t0); t1 = fma(x[e + 1].y, h1, t1);
                    }
                }
                s0 += t0; s1 += t1;
                s0 += __shfl_xor(s0, 32, WAVE);
                s1 += __shfl_xor(s1, 32, WAVE);
                if (half == 0) {
                    hpart[wave * NP + 64 * cb + 2 * cl] = s0;
                    hpart[wave * NP + 64 * cb + 2 * cl + 1] = s1;
                }
            }
        }
        block_sync<SYNCW>();
        MXE_STAMP_E(4);
        double r2 = 0.0;
        for (int k = tid; k < NP; k += T) {
            double h = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) h += hpart[wv * NP + k];
            const double r = (k < ns) ? cc[k] * h - gh[k] : 0.0;
            rhot[k] = r;
            r2 = fma(r, r, r2);
        }
        double x4[4] = {pS, pdH, pHn, r2};
        block_reduce2<NW, 4>(x4, pwm, pdu, red);
        S = x4[0]; dH2 = x4[1]; Hn2 = x4[2]; chi2 = x4[3] + cperp; wmax = pwm; dumax = pdu;
        MXE_STAMP_E(5);
    };

    auto accept_trial = [&]() {
        for (int i = tid; i < nwp; i += T) { u[i] = ut[i]; w[i] = wt[i]; }
        for (int k = tid; k < NP; k += T) { v[k] -= dl[k]; rho[k] = rhot[k]; }
        block_sync<SYNCW>();
    };

    // ------------------------------------------------------------------
    // Gram matrix of the active block: W_aa = V_a^T diag(w) V_a -> Wm
    // (upper triangle + diagonal).  Lane (br, bc) of every wave owns one
    // 4 x 4 register tile in each 32 x 32 super-block; the waves split the
    // omega rows; rows are staged through LDS (double buffered), operands
    // are fetched with 16-byte LDS reads.
    // ------------------------------------------------------------------
    auto gram_sweep = [&](auto PTag, const int* psr, const int* psc) {
        constexpr int P = decltype(PTag)::value;     // super-block pairs per sweep
        const int br = lane >> 3, bc = lane & 7;
        double acc[P][4][4];
#pragma unroll
        for (int q = 0; q < P; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[q][j][k] = 0.0;
        double* xs = stage + (size_t)wave * 2 * GRAM_R * SROW;
        double* ys = xs + GRAM_R * SROW;
        const int n_tiles = (nw + GRAM_R - 1) / GRAM_R;
        double val[RPL][GRAM_R], wv_[GRAM_R];
        auto load_tile = [&](int t) {          // global -> registers (columns lane, lane+64)
            const int i0 = t * GRAM_R;
#pragma unroll
            for (int r = 0; r < GRAM_R; ++r) {
                const int i = i0 + r;            // < n_omega_pad (zero rows beyond n_omega)
                wv_[r] = w[i];
#pragma unroll
                for (int q = 0; q < RPL; ++q) val[q][r] = V[(size_t)i * NP + lane + 64 * q];
            }
        };
        auto store_tile = [&]() {              // registers -> LDS (x = w*V, y = V)
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const int k = lane + 64 * q;
                const int pos = (k >> 2) * GBLK + (k & 3);
#pragma unroll
                for (int r = 0; r < GRAM_R; ++r) {
                    ys[r * SROW + pos] = val[q][r];
                    xs[r * SROW + pos] = val[q][r] * wv_[r];
                }
            }
        };
        int t = wave;
        if (t < n_tiles) load_tile(t);
        for (; t < n_tiles; t += NW) {
            wave_sync();                            // previous tile fully consumed
            store_tile();
            wave_sync();
            if (t + NW < n_tiles) load_tile(t + NW); // global loads in flight during the FMAs
#pragma unroll
            for (int r = 0; r < GRAM_R; ++r) {
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const double2* xp = reinterpret_cast<const double2*>(xs + r * SROW + (8 * psr[q] + br) * GBLK);
                    const double2* yp = reinterpret_cast<const double2*>(ys + r * SROW + (8 * psc[q] + bc) * GBLK);
                    const double2 xa = xp[0], xb = xp[1], ya = yp[0], yb = yp[1];
                    const double x[4] = {xa.x, xa.y, xb.x, xb.y};
                    const double y[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[q][j][k] = fma(x[j], y[k], acc[q][j][k]);
                }
            }
        }
        // reduce over the waves into Wm (upper + diagonal)
        for (int wv = 0; wv < NW; ++wv) {
            if (wave == wv) {
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    if (psr[q] == psc[q] && br > bc) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int row = 32 * psr[q] + 4 * br + j, col = 32 * psc[q] + 4 * bc + k;
                            if (row <= col) {
                                if (wv == 0) Wm[row * LD + col] = acc[q][j][k];
                                else Wm[row * LD + col] += acc[q][j][k];
                            }
                        }
                }
            }
            block_sync<SYNCW>();
        }
    };

    // ------------------------------------------------------------------
    // Gram matrix on the matrix cores: W_aa = (w o V_a)^T V_a as rank-4
    // updates of 16 x 16 tiles with v_mfma_f64_16x16x4_f64.  Operand layout
    // (one f64 per lane): A[m = l&15][k = l>>4], B[k = l>>4][n = l&15], so a
    // lane simply loads V[i0 + (l>>4)][16 t + (l&15)] -- four 128-byte row
    // segments per wave, straight from L2, no LDS staging.  The upper tile
    // pairs (mt <= nt) are accumulated; C/D layout of the f64 form:
    // col = l&15, row = (l>>4) + 4 r.
    // ------------------------------------------------------------------
    auto gram_mfma = [&](auto NTTag) {
        constexpr int NT = decltype(NTTag)::value;       // 16-column tiles covering the active block
        constexpr int NPAIR = NT * (NT + 1) / 2;
        constexpr int DEPTH = 4;                         // row groups in flight
        typedef typename ST::acc4 d4;
        d4 acc[NPAIR];
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) acc[q] = d4{0, 0, 0, 0};
        const int kq = lane >> 4, cn = lane & 15;
        const int n_groups = nwp >> 2;                   // 4 omega rows per MFMA (zero rows beyond n_omega)
        const TS* Vl = V + (size_t)kq * NP + cn;
        TS f[DEPTH][NT], wv_[DEPTH];
        auto load_group = [&](int d, int gidx) {
            const int i0 = 4 * gidx;
            wv_[d] = w[i0 + kq];
#pragma unroll
            for (int t = 0; t < NT; ++t) f[d][t] = Vl[(size_t)i0 * NP + 16 * t];
        };
        int gidx = wave;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (gidx + d * NW < n_groups) load_group(d, gidx + d * NW);
        for (; gidx < n_groups; gidx += DEPTH * NW) {
            TS fc[DEPTH][NT], wc[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                wc[d] = wv_[d];
#pragma unroll
                for (int t = 0; t < NT; ++t) fc[d][t] = f[d][t];
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int gn = gidx + (DEPTH + d) * NW;
                if (gn < n_groups) load_group(d, gn);   // next chunk's loads fly during the MFMAs
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (gidx + d * NW < n_groups) {
                    TS a[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) a[t] = fc[d][t] * wc[d];
                    int q = 0;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int nt = mt; nt < NT; ++nt) {
                            acc[q] = ST::mfma(a[mt], fc[d][nt], acc[q]);
                            ++q;
                        }
                }
            }
        }
        for (int wv = 0; wv < NW; ++wv) {
            if (wave == wv) {
                int q = 0;
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int nt = mt; nt < NT; ++nt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * mt + ST::crow(kq, r), col = 16 * nt + cn;
                            if (wv == 0) Wm[row * LD + col] = (double)acc[q][r];
                            else Wm[row * LD + col] += (double)acc[q][r];
                        }
                        ++q;
                    }
            }
            block_sync<SYNCW>();
        }
    };

    auto gram = [&](int n_act) {
#ifdef MXE_GRAM_VALU
        if constexpr (!F64)
#endif
        {
            const int ntile = (n_act + 15) >> 4;
            if (ntile <= 1) { gram_mfma(std::integral_constant<int, 1>{}); return; }
            if (ntile == 2) { gram_mfma(std::integral_constant<int, 2>{}); return; }
            if (ntile == 3) { gram_mfma(std::integral_constant<int, 3>{}); return; }
            if (ntile == 4) { gram_mfma(std::integral_constant<int, 4>{}); return; }
        }
        if constexpr (F64) {
        const int nsb = (n_act + 31) >> 5;
        if (nsb == 1) {
            const int sr[1] = {0}, sc[1] = {0};
            gram_sweep(std::integral_constant<int, 1>{}, sr, sc);
        } else if (nsb == 2) {
            const int sr[3] = {0, 0, 1}, sc[3] = {0, 1, 1};
            gram_sweep(std::integral_constant<int, 3>{}, sr, sc);
        } else if (NAB > 2) {
            // rare slow path (n_act > 64): sweep the upper super-block pairs three at a time
            int sr[3], sc[3], cnt = 0;
            for (int a = 0; a < nsb; ++a)
                for (int b2 = a; b2 < nsb; ++b2) {
                    sr[cnt] = a; sc[cnt] = b2; ++cnt;
                    if (cnt == 3) { gram_sweep(std::integral_constant<int, 3>{}, sr, sc); cnt = 0; }
                }
            for (int q = 0; q < cnt; ++q) {
                const int s1[1] = {sr[q]}, s2[1] = {sc[q]};
                gram_sweep(std::integral_constant<int, 1>{}, s1, s2);
            }
        }
        }
    };

    // ------------------------------------------------------------------
    // wave 0: Cholesky of A = c W c + a I on the active block (left-looking,
    // lane = row, RPL rows per lane), forward and back substitution.
    // L -> strict lower triangle of Wm, 1/L_jj -> dinv, solution -> zz.
    // ------------------------------------------------------------------
    auto chol_solve = [&](double a, int n_act) -> bool {
        bool ok = true;
        if (wave == 0) {
            // The right-hand side rides along as one more row (index n_act) of
            // the matrix being factorised: its "L" entries are y = L^-1 rhs.
            const bool fused = (n_act < 64 * RPL);
            const int n_rows = fused ? n_act + 1 : n_act;
            double ci_[RPL];
#pragma unroll
            for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; ci_[q] = (i < n_act) ? cc[i] : 0.0; }
            double* Ly = Wm + n_act * LD;               // row n_act: y (fused) -- inside Wm when n_act < NP
            for (int j = 0; j < n_act; ++j) {
                const double cj = cc[j];
                const double rj = rhs[j];
                double s[RPL];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const int i = lane + 64 * q;
                    double sv = 0.0;
                    if (i >= j && i < n_rows) {
                        if (i < n_act) {
                            const double wij = Wm[j * LD + i];
                            sv = (i == j) ? fma(ci_[q] * wij, ci_[q], a) : ci_[q] * wij * cj;
                        } else {
                            sv = rj;
                        }
                        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                        const double* Li = Wm + i * LD;
                        const double* Lj = Wm + j * LD;
                        int k = 0;
                        for (; k + 7 < j; k += 8) {
                            const double l0 = Li[k], l1 = Li[k + 1], l2 = Li[k + 2], l3 = Li[k + 3];
                            const double l4 = Li[k + 4], l5 = Li[k + 5], l6 = Li[k + 6], l7 = Li[k + 7];
                            const double m0 = Lj[k], m1 = Lj[k + 1], m2 = Lj[k + 2], m3 = Lj[k + 3];
                            const double m4 = Lj[k + 4], m5 = Lj[k + 5], m6 = Lj[k + 6], m7 = Lj[k + 7];
                            s0 = fma(l0, m0, s0); s1 = fma(l1, m1, s1); s2 = fma(l2, m2, s2); s3 = fma(l3, m3, s3);
                            s0 = fma(l4, m4, s0); s1 = fma(l5, m5, s1); s2 = fma(l6, m6, s2); s3 = fma(l7, m7, s3);
                        }
                        for (; k < j; ++k) s0 = fma(Li[k], Lj[k], s0);
                        sv -= (s0 + s1) + (s2 + s3);
                    }
                    s[q] = sv;
                }
                const double piv = wave_bcast(s[j >> 6], j & 63);
                if (!(piv > 0.0)) { ok = false; break; }
                // 1/sqrt(piv): hardware estimate + two Newton steps (full binary64 accuracy)
                double inv = __builtin_amdgcn_rsq(piv);
                inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
                inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
                if (lane == 0) dinv[j] = inv;
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const int i = lane + 64 * q;
                    if (i > j && i < n_rows) Wm[i * LD + j] = s[q] * inv;
                }
                wave_sync();
            }
            if (ok) {
                double r[RPL];
                if (fused) {
#pragma unroll
                    for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; r[q] = (i < n_act) ? Ly[i] : 0.0; }
                } else {
                    // forward: L y = rhs (column oriented; lane holds residual rows)
#pragma unroll
                    for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; r[q] = (i < n_act) ? rhs[i] : 0.0; }
                    for (int j = 0; j < n_act; ++j) {
                        const double yj = wave_bcast(r[j >> 6], j & 63) * dinv[j];
#pragma unroll
                        for (int q = 0; q < RPL; ++q) {
                            const int i = lane + 64 * q;
                            if (i == j) r[q] = yj;
                            else if (i > j && i < n_act) r[q] = fma(-Wm[i * LD + j], yj, r[q]);
                        }
                    }
                }
                // backward: L^T z = y ; 1/L_jj in a register of lane j, the row of L
                // needed by step j is fetched one step ahead
                double di[RPL];
#pragma unroll
                for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; di[q] = (i < n_act) ? dinv[i] : 0.0; }
                double lnext[RPL];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const int i = lane + 64 * q;
                    lnext[q] = (i < n_act - 1) ? Wm[(n_act - 1) * LD + i] : 0.0;
                }
                for (int j = n_act - 1; j >= 0; --j) {
                    double lcur[RPL];
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        lcur[q] = lnext[q];
                        const int i = lane + 64 * q;
                        lnext[q] = (j > 0 && i < j - 1) ? Wm[(j - 1) * LD + i] : 0.0;
                    }
                    const double zj = wave_bcast(r[j >> 6] * di[j >> 6], j & 63);
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        const int i = lane + 64 * q;
                        if (i == j) r[q] = zj;
                        else if (i < j) r[q] = fma(-lcur[q], zj, r[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < RPL; ++q) { const int i = lane + 64 * q; if (i < n_act) zz[i] = r[q]; }
            }
        }
        if (NW > 1) {
            if (tid == 0) red[0] = ok ? 1.0 : 0.0;
            __syncthreads();
            ok = red[0] != 0.0;
            __syncthreads();
        } else {
            wave_sync();
        }
        return ok;
    };

    // ------------------------------------------------------------------
    // wave 0, active block of at most N <= 32 rows: the Newton system solved
    // entirely in registers.  Lane i holds the full row i of A = c W c + a I
    // (N doubles, static indices: the j and k loops are fully unrolled) and its
    // right-hand side; Gauss-Jordan elimination with the pivot row broadcast
    // through v_readlane; z falls out of the right-hand-side column.  Rows >=
    // n_act are padded with the identity.  (n_act > 32: chol_solve, in LDS.)
    // ------------------------------------------------------------------
    auto gj_solve_reg = [&](auto NTag, double a, int n_act) -> bool {
        constexpr int N = decltype(NTag)::value;
        bool ok = true;
        if (wave == 0) {
            // Gauss-Jordan elimination on all 64 lanes, two pivots per step (gj2_solve64): the solution falls
            // out of the right-hand-side column, no transposed solve
            const int i = lane & 31, h = lane >> 5;
            const bool live = i < n_act;
            const double ci_ = live ? cc[i] : 0.0;
            double A[N / 2];
            {
                const int ic = min(i, N - 1);
#pragma unroll
                for (int kk = 0; kk < N / 2; ++kk) {
                    // W is kept as upper triangle + diagonal: entry (i, k) sits at [min][max]
                    const int k = 2 * kk + h;
                    double x = 0.0;
                    if (live && k < n_act) x = ci_ * Wm[min(k, ic) * LD + max(k, ic)] * cc[k];
                    if (k == i) x = live ? x + a : 1.0;
                    A[kk] = x;
                }
            }
            double z;
            ok = gj2_solve64<N>(A, live ? rhs[i] : 0.0, i, z);
            if (ok && live && h == 0) zz[i] = z;
        }
        if (NW > 1) {
            if (tid == 0) red[0] = ok ? 1.0 : 0.0;
            __syncthreads();
            ok = red[0] != 0.0;
            __syncthreads();
        } else {
            wave_sync();
        }
        return ok;
    };

    // ------------------------------------------------------------------
    // The same solve for FOUR dampings at once, one per wave (binary64 build with four waves): the damped Newton step tries
    // mu = 0, then mu_1, mu_1 g, mu_1 g^2, ... until a step is accepted -- the sequence is known before the first trial, the
    // matrix is the same, and three of the four waves idle during a solve.  Wave w solves (c W c + (alpha + mu_w) I) z = rhs
    // into zzs[w]; the trial loop takes the solutions in turn.  The arithmetic of every solve is that of gj_solve_reg: the
    // iterates do not change by a bit.  A chain that needs damping at every iteration (the chains that run into maxiter:
    // 2.8 evaluations per iteration) pays one solve per iteration instead of one per evaluation.
    // ------------------------------------------------------------------
    constexpr bool SPEC = F64 && NW == 4 && NAB == 2;
    double* zzs = stage + 272;                              // [4][NP]   (stage: free between the Gram sweeps)
    int* spec_ok = reinterpret_cast<int*>(stage + 258);     // [4]
    auto gj_solve_spec = [&](auto NTag, const double (&aw)[4], int n_act) {
        constexpr int N = decltype(NTag)::value;
        if constexpr (SPEC) {
            const double a = wave == 0 ? aw[0] : wave == 1 ? aw[1] : wave == 2 ? aw[2] : aw[3];
            const int i = lane & 31, h = lane >> 5;
            const bool live = i < n_act;
            const double ci_ = live ? cc[i] : 0.0;
            double A[N / 2];
            {
                const int ic = min(i, N - 1);
#pragma unroll
                for (int kk = 0; kk < N / 2; ++kk) {
                    const int k = 2 * kk + h;
                    double x = 0.0;
                    if (live && k < n_act) x = ci_ * Wm[min(k, ic) * LD + max(k, ic)] * cc[k];
                    if (k == i) x = live ? x + a : 1.0;
                    A[kk] = x;
                }
            }
            double z;
            const bool ok = gj2_solve64<N>(A, live ? rhs[i] : 0.0, i, z);
            if (ok && live && h == 0) zzs[wave * NP + i] = z;
            if (lane == 0) spec_ok[wave] = ok ? 1 : 0;
            __syncthreads();
        }
    };

    // ------------------------------------------------------------------
    // all four waves, active block of 33 .. 64 rows (binary64 build, NP = 64): Gauss-Jordan elimination on pivot PAIRS
    // with the matrix in registers -- lane i of wave w holds the columns 16 w .. 16 w + 15 of row i of A = c W c + a I,
    // every wave a copy of the right-hand side.  A step for the pivots (p, q = p + 1): the wave that owns their columns
    // gives every row its two multipliers [A_ip A_iq] P^-1 (P: the 2 x 2 pivot block) through LDS (two buffers in turn,
    // ONE barrier per step); every wave broadcasts the rows p and q of its own columns with v_readlane; then 34 FMAs per
    // lane.  Rows p and q are left as they are: at the end the matrix is block diagonal and every pair solves its 2 x 2
    // system.  No pivoting (the matrix is positive definite; a pivot block that is not ends the solve like a failed
    // Cholesky).  47.5 k cycles for 56 rows against the 139 k of chol_solve on one wave (profiles/r04_f_onechain_phases.txt)
    // -- 1 700 cycles per step for ~100 instructions: on a lone wave a dependent instruction costs 20-26 cycles and a
    // v_readlane 20-24 (profiles/r04_f_dep_latency.txt); three other arrangements of the step came out within 10 %.
    // ------------------------------------------------------------------
    auto gj_solve_4w = [&](double a, int n_act) -> bool {
        bool ok = true;
        if constexpr (F64 && NW == 4 && NAB == 2) {
            const int i = lane;
            const bool live = i < n_act;
            const double ci_ = live ? cc[i] : 0.0;
            double A[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 16 * wave + kk;
                double x = 0.0;
                if (live && k < n_act) x = ci_ * Wm[min(k, i) * LD + max(k, i)] * cc[k];
                if (k == i) x = live ? x + a : 1.0;
                A[kk] = x;
            }
            double b = live ? rhs[i] : 0.0;
            double* mult = stage;                       // [2][64][2]: (m_p, m_q) of row i
            int* flag = reinterpret_cast<int*>(stage + 256);            // [2] (+ padding up to rowb)
            // (unrolled over the eight pairs of a column block: the owner's two columns are then static registers; ONE loop body
            //  with the columns picked by conditional moves was measured and is 10 % slower, 52.9 k against 47.5 k cycles)
            int step = 0;
            for (int blk = 0; blk < 4; ++blk) {
                if (16 * blk >= n_act) break;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int p = 16 * blk + 2 * kk;
                    if (ok && p < n_act) {
                        const int buf = step & 1;
                        double* mb = mult + buf * 128;
                        if (wave == blk) {
                            const double pp = wave_bcast(A[2 * kk], p), pq = wave_bcast(A[2 * kk + 1], p);
                            const double qp = wave_bcast(A[2 * kk], p + 1), qq = wave_bcast(A[2 * kk + 1], p + 1);
                            const double det = fma(pp, qq, -pq * qp);
                            const bool good = pp > 0.0 && det > 0.0;
                            // 1 / det: hardware estimate + two Newton steps (full binary64 accuracy, a shorter chain than the division)
                            double inv = __builtin_amdgcn_rcp(det);
                            inv = fma(fma(-det, inv, 1.0), inv, inv);
                            inv = fma(fma(-det, inv, 1.0), inv, inv);
                            double mp = fma(A[2 * kk], qq, -A[2 * kk + 1] * qp) * inv;
                            double mq = fma(A[2 * kk + 1], pp, -A[2 * kk] * pq) * inv;
                            if (lane == p || lane == p + 1) { mp = 0.0; mq = 0.0; }
                            *reinterpret_cast<double2*>(mb + 2 * lane) = make_double2(mp, mq);
                            if (lane == 0) flag[buf] = good ? 1 : 0;
                        }
                        // the rows p and q of the wave's own columns, lane to all lanes (v_readlane); the owner does this behind its
                        // multipliers, the other waves while they wait for them.  (Through LDS instead -- two lanes write, a barrier,
                        // all read 17 x 16 B, the pivot block taken from those reads, a second barrier for the multipliers --: 50.9 k
                        // cycles; one barrier with rows and multipliers together: 52.7 k.)
                        double rp[17], rq[17];
#pragma unroll
                        for (int k = 0; k < 16; ++k) { rp[k] = wave_bcast(A[k], p); rq[k] = wave_bcast(A[k], p + 1); }
                        rp[16] = wave_bcast(b, p); rq[16] = wave_bcast(b, p + 1);
                        __syncthreads();
                        const double2 m = *reinterpret_cast<const double2*>(mb + 2 * lane);
                        const int fl = flag[buf];
#pragma unroll
                        for (int k = 0; k < 16; ++k) A[k] = fma(-m.y, rq[k], fma(-m.x, rp[k], A[k]));
                        b = fma(-m.y, rq[16], fma(-m.x, rp[16], b));
                        if (__builtin_amdgcn_readfirstlane(fl) == 0) ok = false;
                        ++step;
                    }
                }
            }
            if (ok) {
                // the 2 x 2 systems of the pairs, each in the wave that holds its columns
                double d = 0.0, off = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if ((lane & 15) == k) d = A[k];
                    if (((lane & 15) ^ 1) == k) off = A[k];
                }
                const double d2 = __shfl_xor(d, 1, WAVE), off2 = __shfl_xor(off, 1, WAVE), b2 = __shfl_xor(b, 1, WAVE);
                const double z = fma(d2, b, -off * b2) / fma(d, d2, -off * off2);
                if ((lane >> 4) == wave && live) zz[i] = z;
            }
            __syncthreads();
        }
        return ok;
    };

    // symmetric mat-vec on the active block: out = W_aa x_a  (threads < NP)
    auto symv = [&](const double* x, double* out, int n_act) {
        for (int i = tid; i < NP; i += T) {
            double s0 = 0.0, s1 = 0.0;
            if (i < n_act) {
                for (int j = 0; j < i; ++j) s0 = fma(Wm[j * LD + i], x[j], s0);
                for (int j = i; j < n_act; ++j) s1 = fma(Wm[i * LD + j], x[j], s1);
            }
            out[i] = s0 + s1;
        }
    };

    // ------------------------------------------------------------------
    // initial state: u = V v0
    // ------------------------------------------------------------------
    double chi2, S, dH2, Hn2, wmax, dumax;
    eval_pass(v, true, chi2, S, dH2, Hn2, wmax, dumax);
    accept_trial();                 // dl == 0: v unchanged
    int nevals_pending = 1, niter_pending = 0;

    const int prob0 = p.chain_prob0[chain], clen = p.chain_len[chain];
    for (int ia = 0; ia < clen; ++ia) {
        const double alpha = p.alpha[(size_t)prob0 + ia];
        int n_iter = 0, conv = 0, nevals = nevals_pending, n_act_last = 0, capp = 0;
        nevals_pending = 0;
        double Qprev = __builtin_nan("");
        double Q = 0.5 * chi2 - alpha * S;
        bool failed = false;
        double relH_prev = 1e300;
        double mu_hint = 0.0;       // damping the last damped step of this alpha needed
        int stuck = 0;              // consecutive iterations that ended at that same damping

        const int maxit = p.prob_maxiter ? p.prob_maxiter[(size_t)prob0 + ia] : p.maxiter;
        for (int it = 0; it < maxit && !failed; ++it) {
            for (int k = tid; k < NP; k += T) {
                const double vv = v[k], r = rho[k];
                g[k]   = (k < ns) ? fma(cc[k], r, alpha * vv) : 0.0;
                rhs[k] = (k < ns) ? fma(alpha * vv, ci[k], r) : 0.0;
            }
            // active block for mu = 0 (the largest it can be in this iteration)
            int n_act0;
            {
                const double thr = p.theta * alpha / fmax(wmax, 1e-300);
                int cnt = 0;
                for (int k = lane; k < ns; k += 64) cnt += (cc[k] * cc[k] > thr) ? 1 : 0;
                n_act0 = (int)wave_sum((double)cnt);
                if (p.theta <= 0.0) n_act0 = ns;
                if (n_act0 < 1) n_act0 = 1;
                if (n_act0 > NP) n_act0 = NP;
            }
            block_sync<SYNCW>();
            MXE_STAMP(0);
            gram(n_act0);               // ends with a block sync
            MXE_STAMP(1);
            n_act_last = n_act0;
            // reference-style criteria (convergence_methods.py:81-122)
            bool stop = false;
            if (p.tol_d > 0.0) {
                symv(g, zz, n_act0);
                block_sync<SYNCW>();
                double mx = 0.0;
                for (int k = tid; k < n_act0; k += T) mx = fmax(mx, fabs(zz[k]));
                double none[1] = {0.0};
                block_reduce<NW, 1>(none, mx, red);
                if (mx < p.tol_d) stop = true;
                block_sync<SYNCW>();
            }
            if (p.tol_relq > 0.0 && it > 0) {
                if (fabs(fabs(Qprev - Q) / Q) < p.tol_relq) stop = true;
            }
            if (stop && it >= p.miniter) { conv = 1; break; }

            // ---- damped Newton step with Bryan's step bound ----
            // An alpha that crawls -- hundreds of iterations at one heavy damping, each of which first tries the undamped
            // step and the damping below and has both refused (profiles/r05_experiments.txt section 6) -- pays three or four
            // evaluations for every step it takes.  After four iterations at the same damping those two tries are made every
            // eighth iteration only; the others start at the damping that was accepted.  (Trying the damping below at every
            // iteration, or every second, keeps 2 of the 4 alphas in 367 359 of the stress set that this costs -- they converge
            // within ten iterations of maxiter -- for 20 % and 7 % more time.)
            double mu = (p.stuck_skip && stuck >= 4 && (it & 7) != 0) ? mu_hint : 0.0;
            double chi2t = 0.0, St = 0.0, dH2t = 0.0, Hn2t = 0.0, wmaxt = 0.0, dumaxt = 0.0;
            bool accepted = false, scaled = false, predicted = false;
#ifdef MXE_DEBUG_HIST
            double sc_cur = 1.0, dnorm = 0.0;   // the factor the step in dl was shortened by, and its length before
#endif
            int spec_cnt = 0;                   // dampings of this iteration whose solutions are in zzs (gj_solve_spec)
            double spec_mu[4] = {0.0, 0.0, 0.0, 0.0};
            while (true) {
                const double a = alpha + mu;
                bool okc;
                const double* zsol = zz;
                if (SPEC && n_act0 <= 32) {
                    int hit = -1;
#pragma unroll
                    for (int w = 0; w < 4; ++w) if (w < spec_cnt && spec_mu[w] == mu) hit = w;
                    if (hit < 0) {
                        // this damping and the three the loop below would try next
                        double aw[4];
                        spec_mu[0] = mu;
#pragma unroll
                        for (int w = 1; w < 4; ++w)
                            spec_mu[w] = (spec_mu[w - 1] == 0.0) ? fmax(p.mu_first * alpha, mu_hint / p.mu_grow) : spec_mu[w - 1] * p.mu_grow;
#pragma unroll
                        for (int w = 0; w < 4; ++w) aw[w] = alpha + spec_mu[w];
                        if (n_act0 <= 16) gj_solve_spec(std::integral_constant<int, 16>{}, aw, n_act0);
                        else if (n_act0 <= 24) gj_solve_spec(std::integral_constant<int, 24>{}, aw, n_act0);
                        else gj_solve_spec(std::integral_constant<int, 32>{}, aw, n_act0);
                        spec_cnt = 4; hit = 0;
                    }
                    okc = __builtin_amdgcn_readfirstlane(spec_ok[hit]) != 0;
                    zsol = zzs + hit * NP;
                }
                else if (n_act0 <= 16) okc = gj_solve_reg(std::integral_constant<int, 16>{}, a, n_act0);
                else if (n_act0 <= 24) okc = gj_solve_reg(std::integral_constant<int, 24>{}, a, n_act0);
                else if (n_act0 <= 32) okc = gj_solve_reg(std::integral_constant<int, 32>{}, a, n_act0);
                else if (F64 && NW == 4 && NAB == 2) okc = gj_solve_4w(a, n_act0);
                else okc = chol_solve(a, n_act0);
                MXE_STAMP(2);
                bool good = okc;
                if (okc) {
                    // delta = c z ; inactive directions: z = rhs / a
                    double nrm = 0.0;
                    for (int k = tid; k < NP; k += T) {
                        double z = 0.0;
                        if (k < n_act0) {
                            z = zsol[k];
                            nrm += z * (rhs[k] - a * z);     // z^T (c W c) z = delta^T W delta
                        } else if (k < ns) {
                            z = rhs[k] / a;
                        }
                        dl[k] = (k < ns) ? cc[k] * z : 0.0;
                    }
                    double x1[1] = {nrm};
                    double dummy = 0.0;
                    block_reduce<NW, 1>(x1, dummy, red);
                    block_sync<SYNCW>();
                    scaled = false;
#ifdef MXE_DEBUG_HIST
                    sc_cur = 1.0; dnorm = x1[0];
#endif
                    if (!(x1[0] <= step_lim)) {
                        // an undamped Newton step beyond Bryan's bound is shortened onto it (same
                        // direction) instead of being recomputed with damping; it is then accepted
                        // like a damped step (Q must not increase), else the damped path takes over
                        if (mu == 0.0 && x1[0] < 1e300) {
                            const double sc = sqrt(step_lim / x1[0]);
                            for (int k = tid; k < NP; k += T) dl[k] *= sc;
#ifdef MXE_DEBUG_HIST
                            sc_cur = sc;
#endif
                            block_sync<SYNCW>();
                            scaled = true;
                        } else good = false;
                    }
                    MXE_STAMP(3);
                    predicted = false;
                    if (PRED && good && !scaled && mu == 0.0 && n_iter == 0 && ia > 0) {
                        // predictor along the alpha path: the defect of the previous alpha's first Newton
                        // iterate (what its later iterations added) is added to this alpha's first step
                        // (see the lock-step kernel, step 1)
                        // safeguards: only a correction smaller than half the Newton step is used, and the
                        // corrected step must not increase Q (else the damped path takes over)
                        double n2[2] = {0.0, 0.0};
                        for (int k = tid; k < NP; k += T) { n2[0] = fma(ecor[k], ecor[k], n2[0]); n2[1] = fma(dl[k], dl[k], n2[1]); }
                        double dummy2 = 0.0;
                        block_reduce<NW, 2>(n2, dummy2, red);
                        block_sync<SYNCW>();
                        if (n2[0] > 0.0 && n2[0] <= 0.25 * n2[1]) {
                            for (int k = tid; k < NP; k += T) dl[k] -= ecor[k];
                            block_sync<SYNCW>();
                            predicted = true;
                        }
                    }
                    if (good) {
                        eval_pass(dl, false, chi2t, St, dH2t, Hn2t, wmaxt, dumaxt);
                        MXE_STAMP(4);
                        ++nevals;
                        const double Qt = 0.5 * chi2t - alpha * St;
                        if (!(fabs(Qt) <= 1.7e308)) good = false;   // NaN / inf
                        // an undamped Newton step may overshoot (it recovers
                        // quadratically); a step that needed damping must not
                        // make Q worse, or a cold start can land far out
                        else if ((mu > 0.0 || scaled) && Qt > Q + 1e-12 * fabs(Q)) good = false;   // (margin: rounding of Q)
                        // a predicted step may overshoot like any undamped Newton step; only a gross
                        // increase of Q (an extrapolation gone wrong on a coarse alpha mesh) rejects it
                        else if (predicted && Qt > 4.0 * fabs(Q) + 1.0) good = false;
                        // a full Newton step may overshoot (a cold start does, and recovers quadratically); one
                        // that multiplies Q by a million (alpha meshes with steps of a decade) does not come back
                        else if (Qt > 1e6 * (fabs(Q) + 1.0)) good = false;
                    }
                }
                if (good) { accepted = true; break; }
                mu = (mu == 0.0) ? fmax(p.mu_first * alpha, mu_hint / p.mu_grow) : mu * p.mu_grow;
                if (!(mu <= p.mu_max * alpha)) break;
            }
            if (!accepted) { failed = true; break; }
            const double relH = sqrt(dH2t / Hn2);
            // estimate of the next Newton correction after a full step (see mxe_opts.stop_estimate)
            const double relH_next = (p.stop_estimate && mu == 0.0 && !scaled && !predicted) ? (expm1(dumaxt) + p.theta) * relH : relH;
            if (n_iter == 0) capp = predicted ? 2 : (mu == 0.0 && !scaled) ? 1 : 0;
            else if (PRED) for (int k = tid; k < NP; k += T) eacc[k] -= dl[k];
            accept_trial();
            MXE_STAMP(5);
            chi2 = chi2t; S = St; Hn2 = Hn2t; wmax = wmaxt;
            Qprev = Q;
            stuck = (mu > 0.0 && mu == mu_hint) ? stuck + 1 : 0;
            mu_hint = mu;
            Q = 0.5 * chi2 - alpha * S;
            ++n_iter;
            // A DAMPED step is short because of its damping, not because the minimum is near: (B + mu) delta_mu = g against
            // B delta = g gives |delta| <= (1 + mu / alpha) |delta_mu| (B >= alpha in this metric), and the test is made on
            // that bound.  Without it alphas that need heavy damping at every iteration (error bars a hundred times below
            // the noise of the data, few data points) were reported converged with exact Newton corrections up to 9e-4
            // (profiles/r03_i_small_sigma.txt).
            const double undamped = (mu > 0.0) ? 1.0 + mu / alpha : 1.0;
#ifdef MXE_DEBUG_HIST
            if (p.dbg_hist && tid == 0) {
                // slots: iteration 10, 20, 40, 60, 80, 100, 150, 200, 300, 400, 500, 600, 700, 800, 900, last
                constexpr int marks[15] = {10, 20, 40, 60, 80, 100, 150, 200, 300, 400, 500, 600, 700, 800, 900};
                double* hrow = p.dbg_hist + ((size_t)prob0 + ia) * 48;
                for (int m = 0; m < 15; ++m) if (n_iter == marks[m]) { hrow[m] = fmin(relH, relH_next) * undamped; hrow[16 + m] = mu / alpha; hrow[32 + m] = dnorm * sc_cur * sc_cur / step_lim; }
                hrow[15] = fmin(relH, relH_next) * undamped; hrow[31] = mu / alpha; hrow[47] = dnorm * sc_cur * sc_cur / step_lim;
            }
#endif
            if (p.tol_h > 0.0 && fmin(relH, relH_next) * undamped < p.tol_h && n_iter > p.miniter) { conv = 1; break; }
            if constexpr (!F64) {
                // binary32 noise floor: an undamped Newton correction that is already small and no
                // longer shrinks is rounding noise of h = V^T H; the point cannot be improved
                if (mu == 0.0 && relH < 1e-3 && relH > 0.5 * relH_prev && n_iter > p.miniter) { conv = 1; break; }
                relH_prev = (mu == 0.0) ? relH : 1e300;
            }
        }

        if (PRED) {
            // defect of this alpha's first Newton iterate -> predictor of the next alpha, scaled with
            // the square of the ratio of the steps in log alpha
            double rr2 = 0.0;
            if (conv && capp > 0 && ia > 0 && ia + 1 < clen) {
                const double a0 = p.alpha[(size_t)prob0 + ia - 1], a2 = p.alpha[(size_t)prob0 + ia + 1];
                const double q0 = alpha / a0, q1 = a2 / alpha;
                const double rr = log(q1) / log(q0);
                // the extrapolation is an expansion in the step of log alpha: fine meshes only
                if (q0 > 0.74 && q0 < 1.35 && q1 > 0.74 && q1 < 1.35) rr2 = rr * rr;
            }
            for (int k = tid; k < NP; k += T) {
                double e = ((capp == 2 ? ecor[k] : 0.0) + eacc[k]) * rr2;
                if (!(fabs(e) < 1e300)) e = 0.0;
                ecor[k] = e; eacc[k] = 0.0;
            }
            block_sync<SYNCW>();
        }
        // ---- results of this alpha (MaxEntResult fields, maxent_result.py:835-967)
        const long long po = p.out_index ? (long long)p.out_index[(size_t)prob0 + ia] : (long long)prob0 + ia;
        if (po < 0) {
            // a rung of a ladder: no record; what it cost is the next alpha's
            nevals_pending += nevals;
            niter_pending += n_iter;
            MXE_STAMP(6);
            continue;
        }
        const size_t prob = (size_t)po;
        if (p.out_H) {
            double* Ho = p.out_H + prob * nw;
            for (int i = tid; i < nw; i += T) {
                const TS Di = (TS)Dg[i], ui = u[i];
                const TS ep = ST::exp_(ui);
                Ho[i] = (double)((kind == 0) ? Di * ep : Di * ep - Di * recip_exp(ep));
            }
        }
        if (p.out_v) for (int k = tid; k < NP; k += T) p.out_v[prob * NP + k] = v[k];
        if (tid == 0) {
            p.out_chi2[prob] = chi2;
            p.out_S[prob] = S;
            p.out_Q[prob] = Q;
            p.out_niter[prob] = n_iter + niter_pending;
            p.out_conv[prob] = conv;
            p.out_nevals[prob] = nevals;
            p.out_nact[prob] = n_act_last;
        }
        niter_pending = 0;
        MXE_STAMP(6);
    }
#ifdef MXE_PROFILE
    if (tid == 0 && p.prof) for (int q = 0; q < 8; ++q) p.prof[(size_t)chain * 8 + q] = prof_acc[q];
#endif
}

} // namespace mxe
