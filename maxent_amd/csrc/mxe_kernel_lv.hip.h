// mxe_kernel_lv.hip.h -- four alpha chains per workgroup in lock-step, the singular basis RESIDENT IN LDS as binary32
//
// Launches that do not fill the GPU (one scan, a 4 x 4 matrix, one rank's shard of a multi-GPU job) are as long as
// their deepest chain of Newton rounds, and a round of chain_kernel_mc (mxe_kernel_mc.hip.h) spends more than half of
// its time streaming the 448 KB of V and V^T through what one CU gets out of its L2 (32-64 B per cycle,
// profiles/r03_a_l2_stream_rate.txt).  Most of those rounds -- the cold start of a piece, the walk of a led piece --
// run to tolerances of 1e-1 ... 1e-5 and do not need binary64 operands.  This kernel keeps V^T as binary32 in the
// LDS of the workgroup's CU (n_s x (n_omega_pad + 4) floats: 113 KB on the BASELINE grids) and runs the omega-space
// arithmetic -- u = V v, exp, H, w, h = V^T H, the Gram operands -- in binary32 from there; the Newton system, the
// residual rho, chi2, S, Q, v and every decision stay binary64 (the "home" section is the one of chain_kernel_mc).
//
// It serves two callers (mxe_chains_launch):
//   * mxe_opts.precision = MXE_PRECISION_F32 (BASELINE config 5's binary32 leg): the records of this kernel are the
//     caller's results;
//   * binary64 launches that do not fill the GPU: this kernel is the FIRST PASS -- every alpha to 1e-5 --, and
//     chain_kernel_mc then takes every alpha as a piece of its own from that v (one binary64 Newton step and the
//     evaluation that confirms it: two rounds per alpha, all alphas side by side).
//
// Layout of the work.  A workgroup has eight waves and four chain slots that share one data set; wave w serves slot
// q = w & 3 on the half w >> 2 of the omega rows in BOTH passes, so a slot's round involves two waves and the only
// traffic between waves is the step (home wave -> partner, before the passes) and the partial Gram tiles / h
// (partner -> home wave, after them): two workgroup barriers per round, none between the passes.
//   row pass    lane l owns four consecutive rows: u -= V delta as 4 x n_s multiply-adds from one ds_read_b128 of
//               V^T per direction, exp, H, w, S, the sums of the stopping rule; u and w stay in the lane's registers,
//               H and sw = sqrt(w sc2) go to a staging array in LDS for the other lane layout of
//   fused pass  lane (g, m) holds column 16 t + m of V on the rows 8 s + e of its quarter g of the half (the operand
//               layout of v_mfma_f32_16x16x32_f16 with the omega rows as the K index): h = V^T H on the vector unit
//               from the registers as loaded, the Gram tiles X^T X, X = diag(sw) V_a split into two binary16 numbers
//               per element, on the binary16 matrix pipe (three products per tile pair, as in chain_kernel_mc).
// V^T rows are S32 = n_omega_pad + 4 floats apart: the 16-byte quad of element (k, i) is (k * S32 + i) / 4 and
// S32 / 4 is odd, so the sixteen lanes of a ds_read_b128 group (one k each, the same rows) hit sixteen different
// quads, and a row pass read (one k, consecutive rows) is contiguous.
#pragma once
#include "mxe_kernel_mc.hip.h"

namespace mxe {

constexpr int LV_NWV = 8;
#ifndef MXE_X_LV_PARTNER_PRIO
#define MXE_X_LV_PARTNER_PRIO 1
#endif
#ifndef MXE_X_LV_HIONLY
#define MXE_X_LV_HIONLY 0        // 1: Gram tiles from the leading binary16 parts only (11 bits) -- experiment
#endif
constexpr int LV_ACAP = 16;             // alphas of a slot's piece kept in LDS (longer pieces read the mesh from memory)
constexpr double LV_FLOOR = 2e-5;       // corrections below this that no longer halve: the binary32 rounding floor
constexpr int LV_STATIC_LDS = 2304;     // bound on the __shared__ arrays of the kernel (host side: fits-the-LDS test)

// rows of V^T the kernel keeps: the 32 columns of the active block at least, a multiple of eight (row-pass unroll)
inline int lv_rows(int ns) { const int r = (ns + 7) & ~7; return r < 32 ? 32 : r; }
// dynamic LDS in bytes (the carve at the top of the kernel)
inline size_t lv_lds_bytes(int ns, int nwp)
{
    const size_t S32 = (size_t)nwp + 4;
    const size_t stage = 4 * S32 * 4;                                   // H or sw of the four slots
    const size_t solve = 4 * 4 * 64 * 8;                                // rhs, z, two scale vectors (alias sw)
    return (size_t)lv_rows(ns) * S32 * 4 + (4 * 4 * 64 + 2 * 64 + 4 * 64 + 64) * 8 + 3 * 4 * 64 * 4 +
           stage + (stage > solve ? stage : solve) + 4 * 3 * 256 * 4;
}

// exp(x) in binary32: k = rint(x log2 e), the remainder in two pieces (log2 e split), v_exp_f32, ldexp.
// Overflow / underflow come out as inf / 0 like fast_exp.
__device__ __forceinline__ float fast_expf(float x)
{
    const float t = x * 1.44269502162933349609375f;
    const float k = __builtin_rintf(t);
    float r = __builtin_fmaf(x, 1.44269502162933349609375f, -k);
    r = __builtin_fmaf(x, 1.925963033500011e-8f, r);
    const float e = __builtin_amdgcn_exp2f(r);
    const int ki = (int)fminf(fmaxf(k, -300.0f), 300.0f);
    return ldexpf(e, ki);
}

// wave-wide sum / max of a binary32 value, result in every lane (four DPP steps inside the rows of 16 lanes, the four rows by v_readlane)
template <int CTRL> __device__ __forceinline__ float dpp_row_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_f(float x) {
    x += dpp_row_f<DPP_XOR1>(x); x += dpp_row_f<DPP_XOR2>(x); x += dpp_row_f<DPP_HALF_MIRROR>(x); x += dpp_row_f<DPP_MIRROR>(x);
    return (wave_bcast_f(x, 0) + wave_bcast_f(x, 16)) + (wave_bcast_f(x, 32) + wave_bcast_f(x, 48));
}
__device__ __forceinline__ float wave_max_f(float x) {
    x = fmaxf(x, dpp_row_f<DPP_XOR1>(x)); x = fmaxf(x, dpp_row_f<DPP_XOR2>(x));
    x = fmaxf(x, dpp_row_f<DPP_HALF_MIRROR>(x)); x = fmaxf(x, dpp_row_f<DPP_MIRROR>(x));
    return fmaxf(fmaxf(wave_bcast_f(x, 0), wave_bcast_f(x, 16)), fmaxf(wave_bcast_f(x, 32), wave_bcast_f(x, 48)));
}

__global__ __launch_bounds__(64 * LV_NWV, 1)
void chain_kernel_lv(const KParams p, const MCExtra x)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NP = 64, NA = 32, NT = 2, NPAIR = 3, T = 64 * LV_NWV, ACAP = LV_ACAP;
    constexpr bool LEAD = true;
    typedef float g4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = wave & 3, half = wave >> 2;
    if ((int)blockIdx.x >= x.n_wg) return;
    const int ns = p.n_s, nw = p.n_omega, nwp = p.n_omega_pad;
    const int S32 = nwp + 4;
    const int NSL = (((ns + 7) & ~7) < 32) ? 32 : ((ns + 7) & ~7);
    const bool dynamic = x.n_queue > 0;

    // ---- LDS carve (lv_lds_bytes) ----
    float*  VtL  = reinterpret_cast<float*>(lds);                        // [NSL][S32]
    double* vv   = reinterpret_cast<double*>(VtL + (size_t)NSL * S32);   // [MCC][NP]   v
    double* gh   = vv + MCC * NP;                // [MCC][NP]   ghat
    double* rho  = gh + MCC * NP;                // [MCC][NP]
    double* dlc  = rho + MCC * NP;               // [MCC][NP]   step per chain
    double* cc   = dlc + MCC * NP;               // [NP]
    double* ci   = cc + NP;                      // [NP]
    double* hpart = ci + NP;                     // [MCC][NP]   h = V^T H, summed by the two waves of a slot with LDS atomics
    double* red  = hpart + MCC * NP;             // [NWV][8]    the five sums of the row pass per wave
    float*  ecor = reinterpret_cast<float*>(red + LV_NWV * 8);          // [MCC][NP] predictor: defect of the previous alpha's first iterate
    float*  eacc = ecor + MCC * NP;              // [MCC][NP]   ... of this alpha, being accumulated
    float*  xop  = eacc + MCC * NP;              // [MCC][NP]   operand of the row pass: the step (or v), binary32
    float*  Hf   = xop + MCC * NP;               // [MCC][S32]  H of the evaluated point
    float*  swf  = Hf + (size_t)MCC * S32;       // [MCC][S32]  sqrt(w sc2): Gram operand scale, alive inside the passes only
    // the vectors of the solve live where sw does: the passes and the home section never overlap
    double* rhs  = reinterpret_cast<double*>(swf);   // [MCC][NP]
    double* zz   = rhs + MCC * NP;
    double* ssc  = zz + MCC * NP;
    double* csc  = ssc + MCC * NP;
    const size_t stage_f = (size_t)MCC * S32 > (size_t)4 * MCC * NP * 2 ? (size_t)MCC * S32 : (size_t)4 * MCC * NP * 2;
    float*  Wt   = swf + stage_f;                // [MCC][NPAIR][4][64] Gram tiles, accumulator layout of the MFMA, binary32
    __shared__ int s_elem[MCC], s_kind[MCC], s_act[MCC], s_scr[MCC];
    __shared__ double s_alpha[MCC][ACAP];
    __shared__ double s_sd[MCC][14];
    __shared__ float s_zp[MCC][2][16];
    __shared__ double s_scw[MCC][2];
    __shared__ int s_si[MCC][15];

    struct Slot {
        double alpha, mu, chi2, S, Hn2, wmax, Q, Qprev, cperp, steplim, muh;
        double sc2, pred;
        int elem, prob0, clen, ia, niter, nevals, nact, active, scratch, okprev, bt, capp;
        int lead, wide;
        double dHp;
        int slow;
    };
    auto load_slot = [&](Slot& t) {
        const double* d = s_sd[wave]; const int* n = s_si[wave];
        t.alpha = d[0]; t.mu = d[1]; t.chi2 = d[2]; t.S = d[3]; t.Hn2 = d[4]; t.wmax = d[5];
        t.Q = d[6]; t.Qprev = d[7]; t.cperp = d[8]; t.steplim = d[9]; t.muh = d[10]; t.sc2 = d[11]; t.pred = d[12];
        t.elem = n[0]; t.prob0 = n[1]; t.clen = n[2]; t.ia = n[3]; t.niter = n[4]; t.nevals = n[5];
        t.nact = n[6]; t.active = n[7]; t.scratch = n[8]; t.okprev = n[9]; t.bt = n[10]; t.capp = n[11];
        t.lead = n[12]; t.wide = n[13]; t.slow = n[14]; t.dHp = d[13];
    };
    auto store_slot = [&](const Slot& t) {
        if (lane == 0) {
            double* d = s_sd[wave]; int* n = s_si[wave];
            d[0] = t.alpha; d[1] = t.mu; d[2] = t.chi2; d[3] = t.S; d[4] = t.Hn2; d[5] = t.wmax;
            d[6] = t.Q; d[7] = t.Qprev; d[8] = t.cperp; d[9] = t.steplim; d[10] = t.muh; d[11] = t.sc2; d[12] = t.pred;
            n[0] = t.elem; n[1] = t.prob0; n[2] = t.clen; n[3] = t.ia; n[4] = t.niter; n[5] = t.nevals;
            n[6] = t.nact; n[7] = t.active; n[8] = t.scratch; n[9] = t.okprev; n[10] = t.bt; n[11] = t.capp;
            n[12] = t.lead; n[13] = t.wide; n[14] = t.slow; d[13] = t.dHp;
            s_act[wave] = t.active; s_scr[wave] = t.scratch;
        }
        wave_sync();
    };
    auto start_piece = [&](Slot& t, int c) {     // home wave: take chain (piece) c into this slot (as chain_kernel_mc)
        t.elem = p.chain_elem[c];
        t.cperp = p.cperp[t.elem];
        t.steplim = p.step_max * p.sumD[t.elem];
        t.prob0 = p.chain_prob0[c]; t.clen = p.chain_len[c];
        const int lead = p.chain_lead ? p.chain_lead[c] : 0;
        t.lead = lead;
        t.ia = -lead; t.niter = 0; t.nevals = 0; t.nact = 0; t.okprev = 0; t.bt = 0; t.capp = 0; t.wide = 0; t.slow = 0; t.dHp = 0.0;
        // (the alphas of the walk come from the scan's own mesh -- the entries before the piece's first -- or, on a mesh too coarse to
        //  walk on, from a ladder the library laid for this piece: KParams::walk_alpha from chain_walk0[c] on; both loads with indices
        //  inside their arrays)
        const int walk0 = (p.chain_walk0 && lead > 0) ? p.chain_walk0[c] : -1;
        for (int i = lane; i < min(t.clen + lead, ACAP); i += 64) {
            const double a_mesh = p.alpha[(size_t)max(t.prob0 - lead + i, 0)];
            const double a_walk = p.walk_alpha ? p.walk_alpha[max(walk0, 0) + min(i, max(lead - 1, 0))] : a_mesh;
            s_alpha[wave][i] = (walk0 >= 0 && i < lead) ? a_walk : a_mesh;
        }
        t.alpha = (walk0 >= 0) ? p.walk_alpha[walk0] : p.alpha[(size_t)(t.prob0 - lead)];
        t.mu = 0.0; t.muh = 0.0; t.Qprev = __builtin_nan("");
        t.chi2 = 0.0; t.S = 0.0; t.Hn2 = 1.0; t.wmax = 1.0; t.Q = 0.0; t.sc2 = 1.0; t.pred = 0.0;
        t.active = 1; t.scratch = 1;
        gh[wave * NP + lane] = p.ghat[(size_t)t.elem * NP + lane];
        vv[wave * NP + lane] = p.v0[(size_t)p.chain_v0[c] * NP + lane];
        ecor[wave * NP + lane] = 0.0f; eacc[wave * NP + lane] = 0.0f;
        if (lane == 0) { s_elem[wave] = t.elem; s_kind[wave] = p.elem_kind[t.elem]; }
        if (p.init_tab) {
            const int ic = p.chain_init[c];
            if (ic >= 0) {
                const double* Tb = p.init_tab + (size_t)ic * MC_INIT_STRIDE;
                float* Wq = Wt + (size_t)wave * NPAIR * 256;
                for (int i = lane; i < NPAIR * 256; i += 64) Wq[i] = (float)Tb[i];
                const double r = (lane < ns) ? p.c[p.elem_ds[t.elem] * NP + lane] * Tb[NPAIR * 256 + lane] - gh[wave * NP + lane] : 0.0;
                rho[wave * NP + lane] = r;
                const double r2 = wave_sum(r * r);
                const double* sc = Tb + NPAIR * 256 + NP;
                t.S = sc[0]; t.Hn2 = sc[1]; t.wmax = sc[2]; t.sc2 = sc[3];
                t.chi2 = r2 + t.cperp;
                t.Q = 0.5 * t.chi2 - t.alpha * t.S;
                t.scratch = 2;
            }
        }
    };
    auto alpha_at = [&](const Slot& t, int i) -> double {
        return (t.clen + t.lead <= ACAP) ? s_alpha[wave][i + t.lead] : p.alpha[(size_t)t.prob0 + i];
    };

    // ---- first pieces ----
    if (wave < MCC) {
        Slot t;
        int c;
        if (dynamic) {
            int idx = 0;
            if (lane == 0) idx = atomicAdd(x.counter, 1);
            idx = __builtin_amdgcn_readfirstlane(idx);
            c = (idx < x.n_queue) ? x.queue[idx] : -1;
        } else {
            c = x.wg_chains[blockIdx.x * MCC + wave];
        }
        if (c >= 0) start_piece(t, c);
        else {
            t = Slot{1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0.0, 0};
            gh[wave * NP + lane] = 0.0; vv[wave * NP + lane] = 0.0;
            if (lane == 0) { s_elem[wave] = -1; s_kind[wave] = 0; }
        }
        store_slot(t);
        dlc[wave * NP + lane] = 0.0;
        hpart[wave * NP + lane] = 0.0;
    }
    __syncthreads();
    int any_elem = -1;
#pragma unroll
    for (int qq = 0; qq < MCC; ++qq) if (s_elem[qq] >= 0 && any_elem < 0) any_elem = s_elem[qq];
    if (any_elem < 0) return;                    // nothing for this workgroup
    const int ds = __builtin_amdgcn_readfirstlane(p.elem_ds[__builtin_amdgcn_readfirstlane(any_elem)]);
    if (wave == 0) { cc[lane] = p.c[ds * NP + lane]; ci[lane] = p.cinv[ds * NP + lane]; }
    // ---- V^T of the data set into LDS (binary32 copy of the context: KParams::Vtf, [n_ds][NP][n_omega_pad]) ----
    {
        const float* __restrict__ src = p.Vtf + (size_t)ds * NP * nwp;
        const int nq = nwp >> 2;                                         // 16-byte quads per row
        for (int idx = tid; idx < NSL * nq; idx += T) {
            const int k = idx / nq, i4 = idx - k * nq;
            float4 v4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (k < ns) v4 = *reinterpret_cast<const float4*>(src + (size_t)k * nwp + 4 * i4);
            *reinterpret_cast<float4*>(VtL + (size_t)k * S32 + 4 * i4) = v4;
        }
    }
    __syncthreads();

#ifdef MXE_PROFILE
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
    long long prof_rounds = 0;
#define LV_STAMP(idx) do { const long long t__ = clock64(); prof_acc[idx] += t__ - prof_t; prof_t = t__; } while (0)
#else
#define LV_STAMP(idx) do {} while (0)
#endif

    // home wave: the slot's Newton system on the active block (gj_home of chain_kernel_mc; the tiles are binary32 here)
    auto gj_home = [&](auto NTag, double a, int n_act, double isc2, double& nrm_out, bool& small_pivot) -> bool {
        constexpr int N = decltype(NTag)::value;
        static_assert(N <= 32 && N % 2 == 0, "two half-waves of 32 rows");
        constexpr int NHALF = N / 2;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int i = ln & 31, h = ln >> 5;
        const float* Wq = Wt + (size_t)q * NPAIR * 256;
        const double* rq = rhs + q * NP;
        bool ok = true;
        const bool live = i < n_act;
        const double* cq = csc + q * NP;
        const double si = ssc[q * NP + min(i, NP - 1)];
        const double ci_ = live ? cq[i] : 0.0;
        const double cis = ci_ * isc2;
        float A[NHALF], A0[NHALF];
        {
            const int ic = min(i, N - 1);
            const int imt = ic >> 4, iri = ic & 15;
            const int up_l = (imt * NT - imt * (imt - 1) / 2 - imt) * 256 + (iri & 3) * 64 + (iri >> 2) * 16 + h;
            const int lo_l = imt * 256 + iri + h * 64;
            float wr[NHALF], ck[NHALF];
#pragma unroll
            for (int kk = 0; kk < NHALF; ++kk) {
                const int k0 = 2 * kk;
                const int kmt = k0 >> 4, kri = k0 & 15;
                const int up_s = kmt * 256 + kri;
                const int lo_s = (kmt * NT - kmt * (kmt - 1) / 2 - kmt) * 256 + (kri & 3) * 64 + (kri >> 2) * 16;
                wr[kk] = Wq[(k0 + h >= ic) ? up_l + up_s : lo_l + lo_s];
                ck[kk] = (float)cq[k0 + h];
            }
            // (binary32 throughout: the entries are O(1) after the scaling and carry the 21 bits of the Gram products)
            const float cisf = (float)cis, dgf = (float)(a * si * si);
#pragma unroll
            for (int kk = 0; kk < NHALF; ++kk) {
                const int k = 2 * kk + h;
                float xv = (k < n_act) ? cisf * wr[kk] * ck[kk] : 0.0f;
                A0[kk] = xv;
                if (k == i) xv = live ? xv + dgf : 1.0f;
                A[kk] = xv;
            }
        }
        double z;
        small_pivot = false;
        {
            float zf;
            ok = gj2_solve64_f32<N>(A, live ? (float)(rq[i] * si) : 0.0f, i, zf, small_pivot);
            z = (double)zf * si;
            if (h == 0) s_zp[q][i & 1][i >> 1] = live ? zf : 0.0f;
            wave_sync();
            float y = 0.0f;
#pragma unroll
            for (int kk = 0; kk < NHALF; ++kk) y = __builtin_fmaf(A0[kk], s_zp[q][h][kk], y);
            const unsigned yu = __builtin_bit_cast(unsigned, y);
            const auto ys = __builtin_amdgcn_permlane32_swap(yu, yu, false, false);
            const float yt = __builtin_bit_cast(float, (unsigned)ys[0]) + __builtin_bit_cast(float, (unsigned)ys[1]);
            nrm_out = wave_sum((h == 0 && live) ? (double)(zf * yt) : 0.0);
        }
        if (ok && live && h == 0) zz[q * NP + i] = z;
        return ok;
    };

    // per-lane state of the passes: u and w of the lane's four rows of slot q (row pass), the wave's partial Gram tiles
    float ureg[4] = {0.0f, 0.0f, 0.0f, 0.0f}, wreg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    g4 acc[NPAIR];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) acc[pr] = g4{0, 0, 0, 0};
    const int RH = nwp >> 1;                      // omega rows of a half
    long long guard = 0;
    const long long guard_max = (long long)(dynamic ? x.n_queue : 1) * p.n_alpha * (p.maxiter + 64) + 64;

    bool first_round = true;
    while (guard++ < guard_max) {
        // ---- 1. home wave: accept the round before, then right-hand side, active block, solve, step ----
        if (wave < MCC) {
            const int k = lane;
            Slot t;
            if (!first_round) {
                // the slot's Gram tiles: the partner's partial sums are in LDS, this wave's in its registers
                {
                    float* Wq = Wt + (size_t)q * NPAIR * 256;
#pragma unroll
                    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Wq[(pr * 4 + r) * 64 + lane] += acc[pr][r];
                }
                const double h = hpart[q * NP + k];
                hpart[q * NP + k] = 0.0;                               // (the passes of this round add to it again)
                const double r = (k < ns) ? cc[k] * h - gh[q * NP + k] : 0.0;
                const double r2 = wave_sum(r * r);
                const double* ra = red + q * 8, * rb = red + (q + 4) * 8;
                const double sS = ra[0] + rb[0], sdH = ra[1] + rb[1], sHn = ra[2] + rb[2];
                const double swm = fmax(ra[3], rb[3]), sdu = fmax(ra[4], rb[4]);
                load_slot(t);
                if (t.active) {
                    rho[q * NP + k] = r;
                    const double chi2t = r2 + t.cperp, St = sS;
                    const double Qt = 0.5 * chi2t - t.alpha * St;
                    const bool finite = fabs(Qt) <= 1.7e308;
                    bool finish_alpha = false, failed = false; int conv = 0;
                    bool on_path = false;      // (a landing of the walk that ended well inside its tolerance: the next step may be long)
                    const bool fresh = t.scratch == 2;
                    if (t.scratch == 1 || (fresh && !t.okprev)) {
                        ++t.nevals;
                        if (finite) { t.scratch = 0; t.chi2 = chi2t; t.S = St; t.Hn2 = sHn; t.wmax = swm; t.Q = Qt; }
                        else { finish_alpha = true; failed = true; }
                    } else if (!t.okprev) {
                        finish_alpha = true; failed = true;
                    } else if (!finite || ((t.mu > 0.0 || (t.okprev >= 2 && t.okprev <= 4)) && Qt > t.Q + 3e-5 * fabs(t.Q)) ||   // (margin: binary32 rounding of h -- chi2 carries ~1e-4 of it near the minimum)
                               (t.okprev == 5 && Qt > 4.0 * fabs(t.Q) + 1.0) ||
                               (t.okprev == 1 && Qt > 1e6 * (fabs(t.Q) + 1.0))) {
                        ++t.nevals;
                        if (finite && t.bt < 3 && !(t.mu == 0.0 && t.muh > 0.0)) {
                            ++t.bt;
                            t.okprev = 3;
                        } else {
                            t.mu = (t.mu == 0.0) ? fmax(p.mu_first * t.alpha, t.muh / p.mu_grow) : t.mu * p.mu_grow;
                            t.scratch = 1; t.bt = 0;
                            if (!(t.mu <= p.mu_max * t.alpha) || t.nevals >= p.mc_maxevals) { finish_alpha = true; failed = true; }    // (see chain_kernel_mc)
                        }
                    } else {
                        // accepted (the stopping rule of chain_kernel_mc, see there)
                        ++t.nevals;
                        double fac2 = 1.0;
                        const bool estimated = p.stop_estimate && t.mu == 0.0 && t.okprev == 1 && sdu <= 1.0;
                        if (estimated) {
                            const double em1 = sdu * fma(sdu, fma(sdu, fma(sdu, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0);
                            const double fac = em1 + p.theta + MC_GRAM_ERR;
                            fac2 = fmin(1.0, fac * fac);
                        }
                        double relH2_min = fac2 * sdH;
                        if (t.mu > 0.0) {
                            const double ud = 1.0 + t.mu / t.alpha;
                            relH2_min *= ud * ud;
                        }
                        const double pred_here = estimated ? relH2_min : 0.0;
                        if (estimated && t.pred > 0.0 && sdH > t.pred) relH2_min = fmin(sdH, relH2_min * (sdH / t.pred));
                        const double tol_here = (t.ia < 0) ? fmax(p.tol_h, (t.ia == -t.lead) ? MXE_X_LEAD_TOL : MXE_X_WALK_TOL) : p.tol_h;
                        const double tol2Hn = tol_here * tol_here * t.Hn2;
                        vv[q * NP + k] -= dlc[q * NP + k];
                        if (t.niter == 0) t.capp = (t.okprev == 5) ? 2 : (t.okprev == 1 && t.mu == 0.0) ? 1 : 0;
                        else eacc[q * NP + k] -= (float)dlc[q * NP + k];
                        t.chi2 = chi2t; t.S = St; t.Hn2 = sHn; t.wmax = swm;
                        t.Qprev = t.Q; t.Q = Qt; t.muh = t.mu; t.mu = 0.0;
                        t.pred = pred_here;
                        ++t.niter;
                        const bool newton_step = t.okprev != 4 && !fresh;
                        const bool full = newton_step && t.okprev == 1 && t.muh == 0.0;
                        // binary32 rounding floor: a full Newton correction that is already small (below LV_FLOOR: the noise of
                        // h = V^T H in binary32 is ~2e-7 ... 1e-6 of |H|) and no longer halves cannot be improved here.  (With the
                        // 1e-3 of chain_kernel<.., float> single alphas of the BASELINE 4 x 4 batch stopped 1.5e-4 from their
                        // fixed points: a correction of that size that shrinks slowly is slow convergence, not rounding)
                        const bool at_floor = full && t.dHp > 0.0 && sdH < LV_FLOOR * LV_FLOOR * t.Hn2 && sdH > 0.25 * t.dHp;
                        t.slow = 0;
                        t.dHp = full ? sdH : 0.0;
                        t.bt = 0;
                        on_path = newton_step && t.niter == 1 && relH2_min < (MXE_X_WALK_SKIP_TOL * MXE_X_WALK_SKIP_TOL) * t.Hn2;
                        if (newton_step && p.tol_h > 0.0 && relH2_min < tol2Hn && t.niter > p.miniter) { conv = 1; finish_alpha = true; }
                        else if (at_floor && t.niter > p.miniter) { conv = 1; finish_alpha = true; }
                        else if (p.tol_relq > 0.0 && fabs(fabs(t.Qprev - t.Q) / t.Q) < p.tol_relq && t.niter > p.miniter) { conv = 1; finish_alpha = true; }
                        else if (t.niter >= ((t.ia < 0 && t.ia > -t.lead) ? MXE_X_WALK_ITERS : p.mc_maxiter)) finish_alpha = true;
                        else if (t.wide > 0 && t.niter - t.wide >= MXE_X_ILL_ITERS) finish_alpha = true;
                    }
                    if (fresh && t.scratch == 2) t.scratch = 0;
                    if (finish_alpha) {
                        const size_t prob = (size_t)t.prob0 + max(t.ia, 0);
                        const bool own = t.ia >= 0;
                        if (p.out_H && own) {
                            double* Ho = p.out_H + prob * nw;
                            const float* Hq = Hf + (size_t)q * S32;
                            for (int i = lane; i < nw; i += 64) Ho[i] = failed ? __builtin_nan("") : (double)Hq[i];
                        }
                        if (p.out_v && own) p.out_v[prob * NP + lane] = vv[q * NP + lane];
                        if (lane == 0 && own) {
                            p.out_chi2[prob] = t.chi2; p.out_S[prob] = t.S; p.out_Q[prob] = t.Q;
                            p.out_niter[prob] = t.niter; p.out_conv[prob] = conv;
                            p.out_nevals[prob] = t.nevals; p.out_nact[prob] = t.nact;
                        }
                        {
                            float e = 0.0f;
                            if (conv && t.capp > 0 && t.ia > 0 && t.ia + 1 < t.clen) {
                                const double a0 = alpha_at(t, t.ia - 1), a1 = t.alpha, a2 = alpha_at(t, t.ia + 1);
                                const double q0 = a1 / a0, q1 = a2 / a1;
                                const double rr = (fabs(q1 - q0) < 1e-9 * q0) ? 1.0 : log(q1) / log(q0);
                                const bool fine = q0 > 0.74 && q0 < 1.35 && q1 > 0.74 && q1 < 1.35;
                                if (fine)
                                e = ((t.capp == 2 ? ecor[q * NP + k] : 0.0f) + eacc[q * NP + k]) * (float)(rr * rr);
                                if (!(fabsf(e) < 1e30f)) e = 0.0f;
                            }
                            ecor[q * NP + k] = e; eacc[q * NP + k] = 0.0f;
                        }
                        ++t.ia;
                        if (t.ia < -1 && on_path) {         // (the walk: landings a factor MXE_X_WALK_RATIO apart, as in chain_kernel_mc)
                            const double lo = t.alpha * (1.0 / MXE_X_WALK_RATIO), hi = t.alpha * MXE_X_WALK_RATIO;
                            while (t.ia < -1) {
                                const double an = alpha_at(t, t.ia + 1);
                                if (!(an >= lo && an <= hi)) break;
                                ++t.ia;
                            }
                        }
                        if (!conv && p.mc_abandon) {
                            for (int i = max(t.ia, 0) + lane; i < t.clen; i += 64) {
                                const size_t pr = (size_t)t.prob0 + i;
                                p.out_conv[pr] = 0; p.out_niter[pr] = 0; p.out_nevals[pr] = 0; p.out_nact[pr] = 0;
                            }
                            // (the binary64 pass starts every alpha from the v of this pass: an alpha that never ran starts
                            //  where the piece stopped)
                            if (p.out_v)
                                for (int i = max(t.ia, 0); i < t.clen; ++i) p.out_v[((size_t)t.prob0 + i) * NP + lane] = vv[q * NP + lane];
                            t.ia = t.clen;
                        }
                        t.niter = 0; t.nevals = 0; t.mu = 0.0; t.bt = 0; t.capp = 0; t.pred = 0.0; t.wide = 0; t.slow = 0; t.dHp = 0.0;
                        t.Qprev = __builtin_nan("");
                        if (t.ia >= t.clen) {
                            t.active = 0;
                            if (dynamic) {
                                int idx = 0;
                                if (lane == 0) idx = atomicAdd(x.counter, 1);
                                idx = __builtin_amdgcn_readfirstlane(idx);
                                if (idx < x.n_queue) start_piece(t, x.queue[idx]);
                            }
                        } else {
                            t.alpha = alpha_at(t, t.ia);
                            t.Q = 0.5 * t.chi2 - t.alpha * t.S;
                        }
                    }
                }
            } else {
                load_slot(t);
            }
            LV_STAMP(0);
            int okflag = 0;
            double dk = 0.0;
            double dtot = 0.0;
            const double isc2 = ldexp(1.0, -ilogb(t.sc2));
            {
                // (binary32 operands: the scale stays inside the binary32 exponent range)
                const double wm = (t.wmax > 1e-30 && t.wmax < 1e30) ? t.wmax : 1.0;
                const double sc2n = ldexp(1.0, 8 - ilogb(wm));
                if (lane == 0) { s_scw[q][0] = isc2; s_scw[q][1] = sc2n; }
                t.sc2 = sc2n;
            }
            if (t.active && t.scratch == 1) {
                dk = vv[q * NP + k];
            } else if (t.active && t.okprev == 3) {
                dtot = 0.5 * dlc[q * NP + k];
                dk = -dtot;
                okflag = 4;
            } else if (t.active) {
                rhs[q * NP + k] = (k < ns) ? fma(t.alpha * vv[q * NP + k], ci[k], rho[q * NP + k]) : 0.0;
                const double thr = p.theta * t.alpha, wmx = fmax(t.wmax, 1e-300);
                const unsigned long long m = __ballot(k < ns && cc[k] * cc[k] * wmx > thr);
                int na = (p.theta > 0.0) ? __popcll(m) : ns;
                na = max(1, min(na, NA));
                t.nact = na;
                {
                    const double sk = ldexp(1.0, -(ilogb(fma(cc[k] * cc[k], wmx, t.alpha)) >> 1));
                    ssc[q * NP + k] = sk; csc[q * NP + k] = cc[k] * sk;
                }
                wave_sync();
                while (true) {
                    const double a = t.alpha + t.mu;
                    double ia = __builtin_amdgcn_rcp(a);
                    ia = fma(fma(-a, ia, 1.0), ia, ia);
                    ia = fma(fma(-a, ia, 1.0), ia, ia);
                    bool ok, small = false;
                    double nrm_gj = -1.0;
                    if (na <= 16) ok = gj_home(std::integral_constant<int, 16>{}, a, na, isc2, nrm_gj, small);
                    else if (na <= 20) ok = gj_home(std::integral_constant<int, 20>{}, a, na, isc2, nrm_gj, small);
                    else if (na <= 24) ok = gj_home(std::integral_constant<int, 24>{}, a, na, isc2, nrm_gj, small);
                    else if (na <= 28) ok = gj_home(std::integral_constant<int, 28>{}, a, na, isc2, nrm_gj, small);
                    else ok = gj_home(std::integral_constant<int, 32>{}, a, na, isc2, nrm_gj, small);
                    if (t.wide == 0 && __builtin_amdgcn_readfirstlane(__any(small ? 1 : 0))) t.wide = t.niter + 1;
                    if (ok) {
                        double z = 0.0;
                        if (k < na) z = zz[q * NP + k];
                        else if (k < ns) z = rhs[q * NP + k] * ia;
                        const double nrm = nrm_gj;
                        if (nrm <= t.steplim) {
                            okflag = 1; dk = (k < ns) ? cc[k] * z : 0.0;
                            if (t.niter == 0 && t.ia > 0 && t.mu == 0.0) {
                                const double ek = (double)ecor[q * NP + k];
                                const double ne = wave_sum(ek * ek), nd = wave_sum(dk * dk);
                                if (ne > 0.0 && ne <= 0.25 * nd) { dk -= ek; okflag = 5; }
                            }
                            break;
                        }
                        if (t.mu == 0.0 && nrm < 1e300) {
                            const double sc = sqrt(t.steplim / nrm);
                            okflag = 2; dk = (k < ns) ? cc[k] * z * sc : 0.0; break;
                        }
                    }
                    t.mu = (t.mu == 0.0) ? p.mu_first * t.alpha : t.mu * p.mu_grow;
                    if (!(t.mu <= p.mu_max * t.alpha)) break;
                }
            }
            if (t.active && t.scratch == 2) {
                dtot = okflag ? dk : 0.0;
                dk = vv[q * NP + k] - dtot;
            } else
            if (okflag != 4) dtot = okflag ? dk : 0.0;
            dlc[q * NP + k] = dtot;
            xop[q * NP + k] = (float)dk;
            t.okprev = okflag;
            store_slot(t);
            LV_STAMP(1);
        }
        __syncthreads();
        LV_STAMP(2);
        if (!(s_act[0] | s_act[1] | s_act[2] | s_act[3])) break;

#if MXE_X_LV_PARTNER_PRIO > 0
        // The two waves of a slot sit on one SIMD, and the arbiter serves the older one (the home wave) first: left alone it
        // leaves the passes thousands of cycles before its partner, which then runs the tail at one instruction per four
        // cycles instead of two
        if (half) __builtin_amdgcn_s_setprio(MXE_X_LV_PARTNER_PRIO);
#endif
        // ---- 2. row pass: u, w, H of slot q on the rows of this half ----
        {
            const bool scr = s_scr[q] != 0;
            const bool pm = s_kind[q] != 0;
            const float sc_new = (float)s_scw[q][1];
            const int el = (s_elem[q] >= 0) ? s_elem[q] : any_elem;
            const bool rows_live = 4 * lane < RH;
            const int row0 = half * RH + (rows_live ? 4 * lane : 0);
            const double* Dp = p.D + (size_t)el * nwp + row0;
            const double2 D01 = *reinterpret_cast<const double2*>(Dp), D23 = *reinterpret_cast<const double2*>(Dp + 2);
            // du = V x on the lane's four rows: one ds_read_b128 of V^T per direction, two packed multiply-adds; the reads of
            // the next eight directions are in flight while the eight before them are used
            typedef float f2v __attribute__((ext_vector_type(2)));
            f2v du01 = {0.0f, 0.0f}, du23 = {0.0f, 0.0f};
            const float* vp = VtL + row0;
            const float* xq = xop + q * NP;
            float4 vb[2][8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) vb[0][kk] = *reinterpret_cast<const float4*>(vp + (size_t)kk * S32);
            for (int k0 = 0; k0 < NSL; k0 += 16) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const int kb = k0 + 8 * hb;                      // (NSL is a multiple of 8: the second half may lie behind the end)
                    if (kb < NSL) {
                        const int kn = (kb + 8 < NSL) ? kb + 8 : kb;
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) vb[hb ^ 1][kk] = *reinterpret_cast<const float4*>(vp + (size_t)(kn + kk) * S32);
                        const float4 xa = *reinterpret_cast<const float4*>(xq + kb), xb = *reinterpret_cast<const float4*>(xq + kb + 4);
                        const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            const float4 v4 = vb[hb][kk];
                            const f2v xx = {xs[kk], xs[kk]};
                            du01 = f2v{v4.x, v4.y} * xx + du01;
                            du23 = f2v{v4.z, v4.w} * xx + du23;
                        }
                    }
                }
            }
            const float du[4] = {du01.x, du01.y, du23.x, du23.y};
            const float Dv[4] = {(float)D01.x, (float)D01.y, (float)D23.x, (float)D23.y};
            double pS = 0.0;
            float pdH = 0.0f, pHn = 0.0f, pwm = 0.0f, pdu = 0.0f;
            float Hn[4], sn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float vd = du[e];
                const float uq = scr ? vd : ureg[e] - vd;
                const float tq = scr ? 0.0f : wreg[e] * vd;
                const float Di = Dv[e];
                const float ep = fast_expf(uq);
                const float Hp = Di * ep;
                float Hq = Hp, wq = Hp, Sq = Hp - Di - Hp * uq;
                if (pm) {
                    const float Hm = Di * recip_exp(ep);
                    Hq = Hp - Hm; wq = Hp + Hm;
                    Sq += Hm - Di + Hm * uq;
                }
                if (row0 + e >= nw || !rows_live) { Hq = 0.0f; wq = 0.0f; Sq = 0.0f; }
                ureg[e] = uq; wreg[e] = wq;
                Hn[e] = Hq;
                sn[e] = __builtin_sqrtf(fminf(wq * sc_new, 3.0e38f));
                if (rows_live) {
                    pdH = __builtin_fmaf(tq, tq, pdH);
                    pdu = fmaxf(pdu, scr ? 0.0f : fabsf(vd));
                    pS += (double)Sq;
                    pHn = __builtin_fmaf(Hq, Hq, pHn);
                    pwm = fmaxf(pwm, wq);
                }
            }
            if (rows_live) {
                *reinterpret_cast<float4*>(Hf + (size_t)q * S32 + row0) = float4{Hn[0], Hn[1], Hn[2], Hn[3]};
                *reinterpret_cast<float4*>(swf + (size_t)q * S32 + row0) = float4{sn[0], sn[1], sn[2], sn[3]};
            }
            // (S in binary64 -- alpha S is compared with chi2 / 2 --, the quantities of the stopping rule in binary32)
            pS = wave_sum(pS);
            pdH = wave_sum_f(pdH); pHn = wave_sum_f(pHn);
            pwm = wave_max_f(pwm); pdu = wave_max_f(pdu);
            if (lane == 0) {
                double* rw = red + wave * 8;
                rw[0] = pS; rw[1] = (double)pdH; rw[2] = (double)pHn; rw[3] = (double)pwm; rw[4] = (double)pdu;
            }
        }
        wave_sync();                             // H, sw of this wave's rows: written and read by the wave itself
        LV_STAMP(3);

        // ---- 3. fused pass: h = V^T H (vector unit, binary32) and the Gram tiles (split binary16) of slot q on this half ----
        {
            const int g = lane >> 4, m = lane & 15;
            const int RG = RH >> 2;              // rows of a quarter of the half: lane group g
            const int rbase = half * RH + g * RG;
            const int nstep = RG >> 3;           // K = 32 steps: eight rows of every quarter
            const float* swq = swf + (size_t)q * S32 + rbase;
            const float* hq  = Hf + (size_t)q * S32 + rbase;
            const float* vcol[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) vcol[t] = VtL + (size_t)min(16 * t + m, NSL - 1) * S32 + rbase;
            typedef float f2v __attribute__((ext_vector_type(2)));
            f2v hacc[4] = {f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}};
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) acc[pr] = g4{0, 0, 0, 0};
            const int nth = (ns + 15) >> 4;      // 16-column tiles of V that h needs (2 .. 4)
            // one K = 32 step: eight rows of the quarter -- sw and H of the slot (two 16-byte reads each, the same for the
            // sixteen lanes of the group), the lane's column of every tile of V (two reads per tile)
            struct Step { float4 s[2], h[2], v[4][2]; };
            auto load_step = [&](Step& L, int st) {
                L.s[0] = *reinterpret_cast<const float4*>(swq + 8 * st); L.s[1] = *reinterpret_cast<const float4*>(swq + 8 * st + 4);
                L.h[0] = *reinterpret_cast<const float4*>(hq + 8 * st);  L.h[1] = *reinterpret_cast<const float4*>(hq + 8 * st + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= 2 && t >= nth) continue;                    // (uniform)
                    L.v[t][0] = *reinterpret_cast<const float4*>(vcol[t] + 8 * st);
                    L.v[t][1] = *reinterpret_cast<const float4*>(vcol[t] + 8 * st + 4);
                }
            };
            auto use_step = [&](const Step& L) {
                u4v xh[NT], xl[NT];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= 2 && t >= nth) continue;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float4 v4 = L.v[t][hf], h4 = L.h[hf];
                        hacc[t] = f2v{v4.x, v4.y} * f2v{h4.x, h4.y} + hacc[t];
                        hacc[t] = f2v{v4.z, v4.w} * f2v{h4.z, h4.w} + hacc[t];
                        if (t < NT) {
                            const float4 s4 = L.s[hf];
                            const f2v xa = f2v{v4.x, v4.y} * f2v{s4.x, s4.y}, xb = f2v{v4.z, v4.w} * f2v{s4.z, s4.w};
                            const float xs[4] = {xa.x, xa.y, xb.x, xb.y};
#pragma unroll
                            for (int e2 = 0; e2 < 2; ++e2) {
                                const float x0 = xs[2 * e2], x1 = xs[2 * e2 + 1];
                                const auto hh = __builtin_amdgcn_cvt_pkrtz(x0, x1);
                                const unsigned hu = __builtin_bit_cast(unsigned, hh);
                                float l0, l1;
                                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hu), "v"(x0));
                                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hu), "v"(x1));
                                const auto ll = __builtin_amdgcn_cvt_pkrtz(l0, l1);
                                xh[t][2 * hf + e2] = hu;
                                xl[t][2 * hf + e2] = __builtin_bit_cast(unsigned, ll);
                            }
                        }
                    }
                }
#pragma unroll
                for (int prod = 0; prod < (MXE_X_LV_HIONLY ? 1 : 3); ++prod) {
                    int pr = 0;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int nt = mt; nt < NT; ++nt) {
                            const h8 a = __builtin_bit_cast(h8, prod == 2 ? xl[mt] : xh[mt]);
                            const h8 b = __builtin_bit_cast(h8, prod == 1 ? xl[nt] : xh[nt]);
                            acc[pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[pr], 0, 0, 0);
                            ++pr;
                        }
                }
            };
            // (nstep is even: n_omega_pad is a multiple of 128.  The reads of a step are in flight while the one before is used)
            Step La, Lb;
            load_step(La, 0);
            for (int st = 0; st < nstep; st += 2) {
                load_step(Lb, st + 1);
                use_step(La);
                load_step(La, (st + 2 < nstep) ? st + 2 : st);
                use_step(Lb);
            }
            // h: the four quarters of the half are the four lane groups; the two halves meet in LDS (ds_add_f64)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double hs = sum_xor32(sum_xor16((double)hacc[t].x + (double)hacc[t].y));
                if (g == t && 16 * t + m < ns)
                    __hip_atomic_fetch_add(hpart + q * NP + 16 * t + m, hs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // Gram tiles: the partner wave hands its partial tiles over in LDS, the home wave keeps its own in
            // registers and adds them at the top of its section (binary32: ds_add_f32 costs 770 cycles on gfx950)
            if (half == 1) {
                float* Wq = Wt + (size_t)q * NPAIR * 256;
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Wq[(pr * 4 + r) * 64 + lane] = acc[pr][r];
            }
        }
#if MXE_X_LV_PARTNER_PRIO > 0
        if (half) __builtin_amdgcn_s_setprio(0);
#endif
        LV_STAMP(4);
        __syncthreads();
        LV_STAMP(5);
        first_round = false;
#ifdef MXE_PROFILE
        ++prof_rounds;
#endif
    }
    if (dynamic && x.wg_chains && tid == 0) const_cast<int*>(x.wg_chains)[blockIdx.x] = (int)guard - 1;      // (MCExtra: the rounds of this workgroup)
#ifdef MXE_PROFILE
    if (lane == 0 && p.prof && blockIdx.x < 1024) {
        long long* pr = p.prof + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int r = 0; r < 7; ++r) pr[r] = prof_acc[r];
        pr[7] = prof_rounds;
    }
#endif
#undef LV_STAMP
}

} // namespace mxe
