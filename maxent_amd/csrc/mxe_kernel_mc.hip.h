// mxe_kernel_mc.hip.h -- four alpha chains per workgroup, in lock-step
//
// Same mathematics as mxe_kernel.hip.h (see there and DESIGN.md).  What is
// different is how the work is laid on the CU:
//
//  * a workgroup of 4 wavefronts owns 4 chain slots that share one data set
//    (one V).  Every load of V feeds all four chains: the row pass streams
//    V^T once (u = u - V delta for the four steps), the fused pass streams V
//    once and produces both h = V^T H (VALU) and the Gram matrices
//    W = V_a^T diag(w) V_a (f64 MFMA) of the four trial points from the same
//    registers;
//  * wave q is the "home" of slot q: it keeps the slot's scalars in its own
//    registers, factorises the slot's Newton matrix (register Cholesky), takes
//    the step, decides acceptance / convergence and writes the results --
//    four factorisations run side by side on the four SIMDs;
//  * per-chain state in LDS is interleaved [row][chain] so that one 16-byte
//    LDS read serves two chains;
//  * the grid is persistent: a slot that has finished its piece of an alpha
//    scan takes the next piece from a queue (most expensive first).
//
// A round = one Newton iteration for every busy slot.  The Gram matrix of a
// trial point is computed speculatively in the same pass that evaluates it;
// if the trial is rejected (not finite, or a damped step that made Q worse)
// the slot is restored from its v (evaluation from scratch) in the next round
// and retries with a larger damping.
#pragma once
#include "mxe_kernel.hip.h"

namespace mxe {

constexpr int MCC = 4;            // chain slots per workgroup == wavefronts per workgroup

struct MCExtra {
    const int* wg_chains;         // static layout: [n_wg][MCC] chain ids, -1 = empty slot
    int n_wg;
    const int* queue;             // dynamic layout: chain ids in the order they are handed out
    int n_queue;                  //   (0 = static layout)
    int* counter;                 //   next queue position (zeroed before every launch)
};

// NA  capacity of the active block: 32 or 48
// NWV wavefronts per workgroup: 4 (the home waves) or 8 (4 home + 4 helper waves that
//     take half of the rows of the two streaming passes: two waves per SIMD there)
template <int NA, int NWV>
__global__ __launch_bounds__(64 * NWV)
void chain_kernel_mc(const KParams p, const MCExtra x)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = 64 * NWV;
    constexpr int RPT = (NWV == 4) ? 2 : 1;       // omega rows per thread in the row pass
    constexpr int NP = 64;
    constexpr int LD = NA + 1;
    constexpr int NT = NA / 16;                   // 16-column tiles of the Gram block
    constexpr int NPAIR = NT * (NT + 1) / 2;
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= x.n_wg) return;
    const int ns = p.n_s, nw = p.n_omega, nwp = p.n_omega_pad;
    const bool dynamic = x.n_queue > 0;

    // ---- LDS carve ----
    double* Wm   = lds;                          // [MCC][NA][LD]
    double* vv   = Wm + MCC * NA * LD;           // [MCC][NP]   v
    double* rhs  = vv + MCC * NP;                // [MCC][NP]
    double* zz   = rhs + MCC * NP;               // [MCC][NP]
    double* gh   = zz + MCC * NP;                // [MCC][NP]
    double* rho  = gh + MCC * NP;                // [MCC][NP]
    double* dlc  = rho + MCC * NP;               // [MCC][NP]   step per chain (chain major)
    double* cc   = dlc + MCC * NP;               // [NP]
    double* ci   = cc + NP;                      // [NP]
    double* vecI = ci + NP;                      // [NP][MCC]   step / v, chain minor (row-pass operand)
    double* hpart = vecI + NP * MCC;             // [NWV waves][MCC chains][NP]
    double* red  = hpart + NWV * MCC * NP;       // [NWV waves][32]
    double* ui   = red + NWV * 32;               // [nwp][MCC]
    double* wi   = ui + (size_t)nwp * MCC;       // [nwp][MCC]
    double* Hi   = wi + (size_t)nwp * MCC;       // [nwp][MCC]
    __shared__ int s_elem[MCC], s_act[MCC], s_scr[MCC], s_exh, s_new;

    // ---- slot state: lives in the registers of the home wave only ----
    int my_elem = 0, my_prob0 = 0, my_clen = 0, my_ia = 0, my_niter = 0, my_nevals = 0, my_nact = 0;
    double my_alpha = 1.0, my_mu = 0.0, my_chi2 = 0.0, my_S = 0.0, my_Hn2 = 1.0, my_wmax = 1.0, my_Q = 0.0;
    double my_Qprev = __builtin_nan(""), my_cperp = 0.0, my_steplim = 0.0;
    bool my_active = false, my_scratch = false;

    auto start_piece = [&](int c) {              // home wave: take chain (piece) c into this slot
        my_elem = p.chain_elem[c];
        my_cperp = p.cperp[my_elem];
        my_steplim = p.step_max * p.sumD[my_elem];
        my_prob0 = p.chain_prob0[c]; my_clen = p.chain_len[c];
        my_ia = 0; my_niter = 0; my_nevals = 0; my_nact = 0;
        my_alpha = p.alpha[(size_t)my_prob0];
        my_mu = 0.0; my_Qprev = __builtin_nan("");
        my_active = true; my_scratch = true;
        gh[wave * NP + lane] = p.ghat[(size_t)my_elem * NP + lane];
        vv[wave * NP + lane] = p.v0[(size_t)p.chain_v0[c] * NP + lane];
        if (lane == 0) { s_elem[wave] = my_elem; s_act[wave] = 1; s_scr[wave] = 1; s_new = 1; }
    };

    // ---- first pieces ----
    if (tid == 0) { s_exh = dynamic ? 0 : 1; s_new = 0; }
    __syncthreads();
    if (wave < MCC) {
        int c;
        if (dynamic) {
            int idx = 0;
            if (lane == 0) idx = atomicAdd(x.counter, 1);
            idx = __builtin_amdgcn_readfirstlane(idx);
            c = (idx < x.n_queue) ? x.queue[idx] : -1;
        } else {
            c = x.wg_chains[blockIdx.x * MCC + wave];
        }
        if (c >= 0) start_piece(c);
        else {
            // empty slot: evaluates v = 0 of a neighbour's element every round (finite, never used)
            gh[wave * NP + lane] = 0.0; vv[wave * NP + lane] = 0.0;
            if (lane == 0) { s_elem[wave] = -1; s_act[wave] = 0; s_scr[wave] = 1; }
        }
        dlc[wave * NP + lane] = 0.0;
    }
    __syncthreads();
    int any_elem = -1;
#pragma unroll
    for (int q = 0; q < MCC; ++q) if (s_elem[q] >= 0 && any_elem < 0) any_elem = s_elem[q];
    if (any_elem < 0) return;                    // nothing for this workgroup
    const int ds = p.elem_ds[any_elem];
    const double* __restrict__ V  = p.V  + (size_t)ds * nwp * NP;
    const double* __restrict__ Vt = p.Vt + (size_t)ds * NP * nwp;
    if (wave == 0) { cc[lane] = p.c[ds * NP + lane]; ci[lane] = p.cinv[ds * NP + lane]; }
    int kind[MCC];
    const double* Dg[MCC];
    auto refresh_slots = [&]() {
#pragma unroll
        for (int q = 0; q < MCC; ++q) {
            const int e = (s_elem[q] >= 0) ? s_elem[q] : any_elem;
            kind[q] = p.elem_kind[e];
            Dg[q] = p.D + (size_t)e * nwp;
        }
    };
    refresh_slots();
    __syncthreads();

#ifdef MXE_PROFILE
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
    long long prof_rounds = 0;
#endif

    // ------------------------------------------------------------------
    // home wave: register Cholesky of the slot's active block + solve
    // (see chol_solve_reg in mxe_kernel.hip.h)
    // ------------------------------------------------------------------
    auto chol_home = [&](auto NTag, double a, int n_act) -> bool {
        constexpr int N = decltype(NTag)::value;
        const int q = wave, i = lane;
        double* Wq = Wm + (size_t)q * NA * LD;
        const double* rq = rhs + q * NP;
        bool ok = true;
        const bool live = i < n_act;
        const double ci_ = live ? cc[i] : 0.0;
        double A[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double xv = 0.0;
            if (live && j <= i && j < n_act) xv = ci_ * Wq[j * LD + i] * cc[j];
            if (j == i) xv = live ? xv + a : 1.0;
            A[j] = xv;
        }
        double b = live ? rq[i] : 0.0;
        double dinv_i = 1.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double piv = wave_bcast(A[j], j);
            if (!(piv > 0.0)) ok = false;
            double inv = __builtin_amdgcn_rsq(piv);
            inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
            if (i == j) dinv_i = inv;
            const double lij = (i > j) ? A[j] * inv : 0.0;
            A[j] = lij;
            const double yj = wave_bcast(b, j) * inv;
            if (i == j) b = yj;
            b = fma(-lij, yj, b);
#pragma unroll
            for (int k0 = j + 1; k0 < N; k0 += 8) {
                double lk[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) if (k0 + r < N) lk[r] = wave_bcast(lij, k0 + r);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 8; ++r) if (k0 + r < N) A[k0 + r] = fma(-lij, lk[r], A[k0 + r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ok) {
#pragma unroll
            for (int j = 0; j < N; ++j) if (j < i && i < n_act) Wq[i * LD + j] = A[j];
            wave_sync();
            double r = b;
            double lnext = (i < n_act - 1) ? Wq[(n_act - 1) * LD + i] : 0.0;
            for (int j = n_act - 1; j >= 0; --j) {
                const double lcur = lnext;
                lnext = (j > 0 && i < j - 1) ? Wq[(j - 1) * LD + i] : 0.0;
                const double zj = wave_bcast(r * dinv_i, j);
                if (i == j) r = zj;
                else if (i < j) r = fma(-lcur, zj, r);
            }
            if (live) zz[q * NP + i] = r;
        }
        return ok;
    };

    bool okflag_prev = false;                    // home wave: did this round carry a real trial step
    long long guard = 0;
    const long long guard_max = (long long)(dynamic ? x.n_queue : 1) * p.n_alpha * (p.maxiter + 64) + 64;

    while (guard++ < guard_max) {
        // ---- 0. idle slots take the next piece from the queue ----
        if (dynamic && s_exh == 0) {
            __syncthreads();                     // everybody has read s_exh / s_new
            if (tid == 0) s_new = 0;
            __syncthreads();
            if (wave < MCC && !my_active) {
                int idx = 0;
                if (lane == 0) idx = atomicAdd(x.counter, 1);
                idx = __builtin_amdgcn_readfirstlane(idx);
                if (idx < x.n_queue) start_piece(x.queue[idx]);
                else if (lane == 0) s_exh = 1;
            }
            __syncthreads();
            if (s_new) refresh_slots();          // uniform: written before the barrier
        }
        if (!(s_act[0] | s_act[1] | s_act[2] | s_act[3])) break;      // uniform: LDS flags behind a barrier

        // ---- 1. home wave: right-hand side, active block, factorise, solve, step ----
        if (wave < MCC) {
            const int q = wave, k = lane;
            int okflag = 0;
            double dk = 0.0;
            if (my_active && my_scratch) {
                dk = vv[q * NP + k];                 // evaluation from scratch: the operand is v
            } else if (my_active) {
                rhs[q * NP + k] = (k < ns) ? fma(my_alpha * vv[q * NP + k], ci[k], rho[q * NP + k]) : 0.0;
                const double thr = p.theta * my_alpha / fmax(my_wmax, 1e-300);
                const unsigned long long m = __ballot(k < ns && cc[k] * cc[k] > thr);
                int na = (p.theta > 0.0) ? __popcll(m) : ns;
                na = max(1, min(na, NA));
                my_nact = na;
                wave_sync();
                // damping loop: raise mu until the factorisation succeeds and Bryan's bound holds
                while (true) {
                    const double a = my_alpha + my_mu;
                    bool ok;
                    if (na <= 16) ok = chol_home(std::integral_constant<int, 16>{}, a, na);
                    else if (na <= 24) ok = chol_home(std::integral_constant<int, 24>{}, a, na);
                    else if (NA <= 32 || na <= 32) ok = chol_home(std::integral_constant<int, 32>{}, a, na);
                    else ok = chol_home(std::integral_constant<int, (NA > 32 ? NA : 32)>{}, a, na);
                    if (ok) {
                        double z = 0.0, nrm = 0.0;
                        if (k < na) { z = zz[q * NP + k]; nrm = z * (rhs[q * NP + k] - a * z); }
                        else if (k < ns) z = rhs[q * NP + k] / a;
                        nrm = wave_sum(nrm);
                        if (nrm <= my_steplim) { okflag = 1; dk = (k < ns) ? cc[k] * z : 0.0; break; }
                    }
                    my_mu = (my_mu == 0.0) ? p.mu_first * my_alpha : my_mu * p.mu_grow;
                    if (!(my_mu <= p.mu_max * my_alpha)) break;
                }
            }
            dlc[q * NP + k] = (okflag ? dk : 0.0);
            vecI[k * MCC + q] = dk;
            okflag_prev = okflag != 0;
        }
        __syncthreads();
        MXE_STAMP(2);

        // ---- 2. row pass (V^T once): u, w, H of the four trial points, in place ----
        {
            bool scr[MCC];
#pragma unroll
            for (int q = 0; q < MCC; ++q) scr[q] = s_scr[q] != 0;
            double pS[MCC], pdH[MCC], pHn[MCC], pwm[MCC];
#pragma unroll
            for (int q = 0; q < MCC; ++q) { pS[q] = 0.0; pdH[q] = 0.0; pHn[q] = 0.0; pwm[q] = 0.0; }
            for (int i = RPT * tid; i < nwp; i += RPT * T) {
                double a[MCC][RPT];
#pragma unroll
                for (int q = 0; q < MCC; ++q)
#pragma unroll
                    for (int r = 0; r < RPT; ++r) a[q][r] = 0.0;
                const double* col = Vt + i;
#pragma unroll 4
                for (int k = 0; k < ns; ++k) {
                    double xr[RPT];
                    if (RPT == 2) {
                        const double2 xv = *reinterpret_cast<const double2*>(col + (size_t)k * nwp);
                        xr[0] = xv.x; xr[RPT - 1] = xv.y;
                    } else {
                        xr[0] = col[(size_t)k * nwp];
                    }
                    const double2 d01 = *reinterpret_cast<const double2*>(vecI + k * MCC);
                    const double2 d23 = *reinterpret_cast<const double2*>(vecI + k * MCC + 2);
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        a[0][r] = fma(xr[r], d01.x, a[0][r]); a[1][r] = fma(xr[r], d01.y, a[1][r]);
                        a[2][r] = fma(xr[r], d23.x, a[2][r]); a[3][r] = fma(xr[r], d23.y, a[3][r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const int ii = i + r;
                    double un[MCC], wn[MCC], Hn_[MCC];
#pragma unroll
                    for (int q = 0; q < MCC; ++q) {
                        const double vd = a[q][r];
                        double uq;
                        if (scr[q]) uq = vd;
                        else {
                            uq = ui[ii * MCC + q] - vd;
                            const double t = wi[ii * MCC + q] * vd;
                            pdH[q] = fma(t, t, pdH[q]);
                        }
                        const double Di = Dg[q][ii];
                        double Hq, wq, Sq;
                        if (kind[q] == 0) {
                            const double e = exp(uq);
                            Hq = Di * e; wq = Hq;
                            Sq = Hq - Di - Hq * uq;
                        } else {
                            const double ep = exp(uq), em = exp(-uq);
                            const double Hp = Di * ep, Hm = Di * em;
                            Hq = Hp - Hm; wq = Hp + Hm;
                            Sq = (Hp - Di - Hp * uq) + (Hm - Di + Hm * uq);
                        }
                        if (ii >= nw) { Hq = 0.0; wq = 0.0; Sq = 0.0; }
                        un[q] = uq; wn[q] = wq; Hn_[q] = Hq;
                        pS[q] += Sq;
                        pHn[q] = fma(Hq, Hq, pHn[q]);
                        pwm[q] = fmax(pwm[q], wq);
                    }
#pragma unroll
                    for (int q = 0; q < MCC; ++q) {
                        ui[ii * MCC + q] = un[q]; wi[ii * MCC + q] = wn[q]; Hi[ii * MCC + q] = Hn_[q];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < MCC; ++q) {
                pS[q] = wave_sum(pS[q]); pdH[q] = wave_sum(pdH[q]); pHn[q] = wave_sum(pHn[q]);
                pwm[q] = wave_max(pwm[q]);
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < MCC; ++q) {
                    red[wave * 32 + q * 4 + 0] = pS[q]; red[wave * 32 + q * 4 + 1] = pdH[q];
                    red[wave * 32 + q * 4 + 2] = pHn[q]; red[wave * 32 + q * 4 + 3] = pwm[q];
                }
            }
        }
        __syncthreads();                         // Hi, wi, ui and the partial sums complete
        MXE_STAMP(3);

        // ---- 3. fused pass (V once): h = V^T H (VALU) and W = V_a^T diag(w) V_a (MFMA) ----
        {
            d4 acc[MCC][NPAIR];
            double hp[MCC][4];
#pragma unroll
            for (int c = 0; c < MCC; ++c) {
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) acc[c][pr] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) hp[c][t] = 0.0;
            }
            const int kq = lane >> 4, cn = lane & 15;
            const int n_groups = nwp >> 2;           // 4 omega rows per MFMA; multiple of 16
            const double* Vl = V + (size_t)kq * NP + cn;
            // four register sets, software pipelined without copies: the loads of
            // the next three row groups are in flight while one is consumed
            double fA[4], fB[4], fC[4], fD[4];
            double2 hA[4], hB[4], hC[4], hD[4];      // [0,1] = H of chains 01 / 23, [2,3] = w of chains 01 / 23
            auto load_group = [&](double (&f)[4], double2 (&hw)[4], int gidx) {
                const int i0 = 4 * gidx;
                const double2* hptr = reinterpret_cast<const double2*>(Hi + (size_t)(i0 + kq) * MCC);
                const double2* wptr = reinterpret_cast<const double2*>(wi + (size_t)(i0 + kq) * MCC);
                hw[0] = hptr[0]; hw[1] = hptr[1]; hw[2] = wptr[0]; hw[3] = wptr[1];
#pragma unroll
                for (int t = 0; t < 4; ++t) f[t] = Vl[(size_t)i0 * NP + 16 * t];
            };
            auto consume = [&](const double (&f)[4], const double2 (&hw)[4]) {
                const double Hq[MCC] = {hw[0].x, hw[0].y, hw[1].x, hw[1].y};
                const double wq[MCC] = {hw[2].x, hw[2].y, hw[3].x, hw[3].y};
                double a[MCC][NT];
#pragma unroll
                for (int c = 0; c < MCC; ++c)
#pragma unroll
                    for (int t = 0; t < NT; ++t) a[c][t] = f[t] * wq[c];
#pragma unroll
                for (int c = 0; c < MCC; ++c) {
                    int pr = 0;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int nt = mt; nt < NT; ++nt) {
                            acc[c][pr] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][mt], f[nt], acc[c][pr], 0, 0, 0);
                            ++pr;
                        }
#pragma unroll
                    for (int t = 0; t < 4; ++t) hp[c][t] = fma(f[t], Hq[c], hp[c][t]);
                }
                // issue order: the VALU work of a group rides in the shadow of its
                // MFMAs (one matrix instruction, then up to two vector ones)
#pragma unroll
                for (int r = 0; r < MCC * NPAIR; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            };
            // the waves take groups wave, wave + NWV, ...; the register sets rotate
            // (four with one wave per SIMD; two with two waves per SIMD, where the
            // other wave covers the latency and registers are halved)
            constexpr int ST = NWV;
            int g = wave;
            if (NWV == 4) {
                if (g < n_groups) load_group(fA, hA, g);
                if (g + ST < n_groups) load_group(fB, hB, g + ST);
                if (g + 2 * ST < n_groups) load_group(fC, hC, g + 2 * ST);
                for (; g < n_groups; g += 4 * ST) {
                    if (g + 3 * ST < n_groups) load_group(fD, hD, g + 3 * ST);
                    consume(fA, hA);
                    if (g + 4 * ST < n_groups) load_group(fA, hA, g + 4 * ST);
                    if (g + ST < n_groups) consume(fB, hB);
                    if (g + 5 * ST < n_groups) load_group(fB, hB, g + 5 * ST);
                    if (g + 2 * ST < n_groups) consume(fC, hC);
                    if (g + 6 * ST < n_groups) load_group(fC, hC, g + 6 * ST);
                    if (g + 3 * ST < n_groups) consume(fD, hD);
                }
            } else {
                if (g < n_groups) load_group(fA, hA, g);
                for (; g < n_groups; g += 2 * ST) {
                    if (g + ST < n_groups) load_group(fB, hB, g + ST);
                    consume(fA, hA);
                    if (g + 2 * ST < n_groups) load_group(fA, hA, g + 2 * ST);
                    if (g + ST < n_groups) consume(fB, hB);
                }
            }
            // h: sum the four row-residue lane groups, then (in step 4) the waves
#pragma unroll
            for (int c = 0; c < MCC; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double v_ = hp[c][t];
                    v_ += __shfl_xor(v_, 16, WAVE);
                    v_ += __shfl_xor(v_, 32, WAVE);
                    if (kq == 0) hpart[(wave * MCC + c) * NP + 16 * t + cn] = v_;
                }
            // Gram tiles: rotating phases (in phase ph wave w adds into chain (w + ph) mod NWV if < 4)
            for (int ph = 0; ph < NWV; ++ph) {
#pragma unroll
                for (int c = 0; c < MCC; ++c) {
                    if (((c - wave) & (NWV - 1)) == ph) {
                        double* Wq = Wm + (size_t)c * NA * LD;
                        int pr = 0;
#pragma unroll
                        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                            for (int nt = mt; nt < NT; ++nt) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = 16 * mt + kq + 4 * r, col = 16 * nt + cn;
                                    if (ph == 0) Wq[row * LD + col] = acc[c][pr][r];
                                    else Wq[row * LD + col] += acc[c][pr][r];
                                }
                                ++pr;
                            }
                    }
                }
                __syncthreads();
            }
        }
        MXE_STAMP(4);

        // ---- 4. home wave: rho, sums, accept / converge / advance, results ----
        if (wave < MCC) {
            const int q = wave, k = lane;
            double h = 0.0;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) h += hpart[(wv * MCC + q) * NP + k];
            const double r = (k < ns) ? cc[k] * h - gh[q * NP + k] : 0.0;
            const double r2 = wave_sum(r * r);
            double sS = 0.0, sdH = 0.0, sHn = 0.0, swm = 0.0;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) {
                sS += red[wv * 32 + q * 4 + 0]; sdH += red[wv * 32 + q * 4 + 1];
                sHn += red[wv * 32 + q * 4 + 2]; swm = fmax(swm, red[wv * 32 + q * 4 + 3]);
            }
            if (my_active) {
                rho[q * NP + k] = r;
                const double chi2t = r2 + my_cperp, St = sS;
                const double Qt = 0.5 * chi2t - my_alpha * St;
                const bool finite = fabs(Qt) <= 1.7e308;
                bool finish_alpha = false; int conv = 0;
                if (my_scratch) {
                    // state restored from v (or first evaluation of the piece); damping kept
                    ++my_nevals;
                    if (finite) { my_scratch = false; my_chi2 = chi2t; my_S = St; my_Hn2 = sHn; my_wmax = swm; my_Q = Qt; }
                    else finish_alpha = true;                   // cannot even evaluate: give up on this alpha
                } else if (!okflag_prev) {
                    finish_alpha = true;                        // the damping loop ran out of range
                } else if (!finite || (my_mu > 0.0 && Qt > my_Q)) {
                    // not finite, or a damped step that made Q worse: more damping, restore from v
                    ++my_nevals;
                    my_mu = (my_mu == 0.0) ? p.mu_first * my_alpha : my_mu * p.mu_grow;
                    my_scratch = true;
                    if (!(my_mu <= p.mu_max * my_alpha)) finish_alpha = true;
                } else {
                    // accepted
                    ++my_nevals;
                    const double relH = sqrt(sdH / my_Hn2);
                    vv[q * NP + k] -= dlc[q * NP + k];
                    my_chi2 = chi2t; my_S = St; my_Hn2 = sHn; my_wmax = swm;
                    my_Qprev = my_Q; my_Q = Qt; my_mu = 0.0;
                    ++my_niter;
                    if (p.tol_h > 0.0 && relH < p.tol_h && my_niter > p.miniter) { conv = 1; finish_alpha = true; }
                    else if (p.tol_relq > 0.0 && fabs(fabs(my_Qprev - my_Q) / my_Q) < p.tol_relq && my_niter > p.miniter) { conv = 1; finish_alpha = true; }
                    else if (my_niter >= p.maxiter) finish_alpha = true;
                }
                if (finish_alpha) {
                    const size_t prob = (size_t)my_prob0 + my_ia;
                    if (p.out_H) {
                        double* Ho = p.out_H + prob * nw;
                        const int kd = p.elem_kind[my_elem];
                        const double* Dq = p.D + (size_t)my_elem * nwp;
                        for (int i = lane; i < nw; i += 64) {
                            const double Di = Dq[i], uq = ui[i * MCC + q];
                            Ho[i] = (kd == 0) ? Di * exp(uq) : Di * exp(uq) - Di * exp(-uq);
                        }
                    }
                    if (p.out_v) p.out_v[prob * NP + lane] = vv[q * NP + lane];
                    if (lane == 0) {
                        p.out_chi2[prob] = my_chi2; p.out_S[prob] = my_S; p.out_Q[prob] = my_Q;
                        p.out_niter[prob] = my_niter; p.out_conv[prob] = conv;
                        p.out_nevals[prob] = my_nevals; p.out_nact[prob] = my_nact;
                    }
                    ++my_ia;
                    my_niter = 0; my_nevals = 0; my_mu = 0.0;
                    my_Qprev = __builtin_nan("");
                    if (my_ia >= my_clen) my_active = false;
                    else {
                        my_alpha = p.alpha[(size_t)my_prob0 + my_ia];
                        my_Q = 0.5 * my_chi2 - my_alpha * my_S;
                    }
                }
                if (lane == 0) { s_act[q] = my_active ? 1 : 0; s_scr[q] = my_scratch ? 1 : 0; }
            }
        }
        __syncthreads();                         // slot flags, v, rho visible to the next round
        MXE_STAMP(5);
#ifdef MXE_PROFILE
        ++prof_rounds;
#endif
    }
#ifdef MXE_PROFILE
    if (tid == 0 && p.prof) { for (int r = 0; r < 7; ++r) p.prof[(size_t)blockIdx.x * 8 + r] = prof_acc[r]; p.prof[(size_t)blockIdx.x * 8 + 7] = prof_rounds; }
#endif
}

} // namespace mxe
