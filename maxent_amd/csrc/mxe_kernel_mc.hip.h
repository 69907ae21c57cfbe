// mxe_kernel_mc.hip.h -- four alpha chains per workgroup, in lock-step
//
// Same mathematics as mxe_kernel.hip.h (see there and DESIGN.md).  What is
// different is how the work is laid on the CU:
//
//  * a workgroup of 4 wavefronts owns 4 chains that share one data set
//    (one V).  Every load of V in the evaluation pass and in the Gram update
//    feeds all four chains (the passes are L2->CU bandwidth bound, 448 KB per
//    chain and iteration otherwise);
//  * wave q is the "home" of chain q: it factorises chain q's Newton matrix
//    (register Cholesky), takes the step, decides acceptance / convergence and
//    writes chain q's results -- four factorisations run side by side on the
//    four SIMDs instead of one wave working and three waiting;
//  * per-chain state in LDS is interleaved [row][chain] so that one 16-byte
//    LDS read serves two chains.
//
// A round = one Newton iteration for every chain that is still active.  A
// chain whose trial point is not finite is restored from its v (evaluation
// from scratch) in the next round with a larger damping; a chain whose step
// violates Bryan's bound re-solves with a larger damping and sits the round
// out (delta = 0).
#pragma once
#include "mxe_kernel.hip.h"

namespace mxe {

constexpr int MCC = 4;            // chains per workgroup == wavefronts per workgroup

struct MCExtra {
    const int* wg_chains;         // static layout: [n_wg][MCC] chain ids, -1 = empty slot
    int n_wg;
    const int* queue;             // dynamic layout: chain ids in the order they are handed out
    int n_queue;                  //   (0 = static layout)
    int* counter;                 //   next queue position (zeroed before every launch)
};

template <int NA>                 // capacity of the active block: 32 or 48
__global__ __launch_bounds__(64 * MCC)
void chain_kernel_mc(const KParams p, const MCExtra x)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = 64 * MCC;
    constexpr int NP = 64;
    constexpr int LD = NA + 1;
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= x.n_wg) return;
    const int ns = p.n_s, nw = p.n_omega, nwp = p.n_omega_pad;

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
    double* vecI = ci + NP;                      // [NP][MCC]   step / v, chain minor (u-pass operand)
    double* hpart = vecI + NP * MCC;             // [MCC waves][MCC chains][NP]
    double* red  = hpart + MCC * MCC * NP;       // [MCC waves][32]
    double* ui   = red + MCC * 32;               // [nwp][MCC]
    double* wi   = ui + (size_t)nwp * MCC;       // [nwp][MCC]
    double* Hi   = wi + (size_t)nwp * MCC;       // [nwp][MCC]
    __shared__ int s_flag[4 * MCC];              // nact[4], ok[4], scratch[4], active[4]
    __shared__ double s_mu[MCC];                 // damping the home wave ended up with
    int* s_nact = s_flag; int* s_ok = s_flag + MCC; int* s_scr = s_flag + 2 * MCC; int* s_act = s_flag + 3 * MCC;

    // ---- chains of this workgroup (slots are refilled from the queue in the dynamic layout) ----
    __shared__ int s_claim[MCC];
    int chain[MCC], elem[MCC], kind[MCC];
    double cperp[MCC], step_lim[MCC];
    const double* Dg[MCC];
    const bool dynamic = x.n_queue > 0;
    if (dynamic) {
        if (lane == 0) {
            const int idx = atomicAdd(x.counter, 1);
            s_claim[wave] = (idx < x.n_queue) ? x.queue[idx] : -1;
        }
        __syncthreads();
    }
    int first_valid = -1;
#pragma unroll
    for (int q = 0; q < MCC; ++q) {
        chain[q] = dynamic ? s_claim[q] : x.wg_chains[blockIdx.x * MCC + q];
        if (chain[q] >= 0 && first_valid < 0) first_valid = chain[q];
    }
    if (first_valid < 0) return;                 // nothing left for this workgroup
#pragma unroll
    for (int q = 0; q < MCC; ++q) {
        const int e = p.chain_elem[(chain[q] >= 0) ? chain[q] : first_valid];
        elem[q] = e;
        kind[q] = p.elem_kind[e];
        cperp[q] = p.cperp[e];
        step_lim[q] = p.step_max * p.sumD[e];
        Dg[q] = p.D + (size_t)e * nwp;
    }
    const int ds = p.elem_ds[elem[0]];
    const double* __restrict__ V  = p.V  + (size_t)ds * nwp * NP;
    const double* __restrict__ Vt = p.Vt + (size_t)ds * NP * nwp;
    // thread (wave q, lane k) owns component k of chain q in the small vectors
    {
        const int q = wave, k = lane;
        if (q == 0) { cc[k] = p.c[ds * NP + k]; ci[k] = p.cinv[ds * NP + k]; }
        gh[q * NP + k] = p.ghat[(size_t)elem[q] * NP + k];
        const double v0 = (chain[q] >= 0) ? p.v0[(size_t)p.chain_v0[chain[q]] * NP + k] : 0.0;
        vv[q * NP + k] = v0;
        vecI[k * MCC + q] = v0;
        dlc[q * NP + k] = 0.0;
        if (lane == 0) { s_act[q] = (chain[q] >= 0) ? 1 : 0; s_scr[q] = 1; s_ok[q] = 0; s_nact[q] = 1; }
    }
    __syncthreads();

#ifdef MXE_PROFILE
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
    long long prof_rounds = 0;
#define prof_rounds_inc() (++prof_rounds)
#else
#define prof_rounds_inc() ((void)0)
#endif
    // per-chain scalars, identical in every thread
    double alpha[MCC], mu[MCC], chi2[MCC], S[MCC], Hn2[MCC], wmax[MCC], Q[MCC], Qprev[MCC], relH[MCC];
    int ia[MCC], n_iter[MCC], nevals[MCC], it_alpha[MCC], nact_last[MCC], prob0[MCC], clen[MCC];
    bool active[MCC], scratch[MCC];
#pragma unroll
    for (int q = 0; q < MCC; ++q) {
        active[q] = chain[q] >= 0; scratch[q] = true;
        ia[q] = 0; n_iter[q] = 0; nevals[q] = 0; it_alpha[q] = 0; nact_last[q] = 0;
        prob0[q] = active[q] ? p.chain_prob0[chain[q]] : 0;
        clen[q] = active[q] ? p.chain_len[chain[q]] : 0;
        alpha[q] = active[q] ? p.alpha[(size_t)prob0[q]] : 1.0;
        mu[q] = 0.0; chi2[q] = 0.0; S[q] = 0.0; Hn2[q] = 1.0; wmax[q] = 1.0; Q[q] = 0.0;
        Qprev[q] = __builtin_nan(""); relH[q] = 0.0;
    }

    // ------------------------------------------------------------------
    // shared evaluation pass for the four chains.
    //   chain q scratch : u_q = V vec_q           (vec = v)
    //   else            : u_q = u_q - V vec_q     (vec = delta)
    // updates ui, wi, Hi in place; rho; returns per-chain sums.
    // ------------------------------------------------------------------
    auto eval_pass = [&](double (&oS)[MCC], double (&odH)[MCC], double (&oHn)[MCC],
                         double (&owm)[MCC]) {
        double pS[MCC], pdH[MCC], pHn[MCC], pwm[MCC];
#pragma unroll
        for (int q = 0; q < MCC; ++q) { pS[q] = 0.0; pdH[q] = 0.0; pHn[q] = 0.0; pwm[q] = 0.0; }
        for (int i = 2 * tid; i < nwp; i += 2 * T) {
            double a[MCC][2];
#pragma unroll
            for (int q = 0; q < MCC; ++q) { a[q][0] = 0.0; a[q][1] = 0.0; }
            const double* col = Vt + i;
#pragma unroll 4
            for (int k = 0; k < ns; ++k) {
                const double2 xv = *reinterpret_cast<const double2*>(col + (size_t)k * nwp);
                const double2 d01 = *reinterpret_cast<const double2*>(vecI + k * MCC);
                const double2 d23 = *reinterpret_cast<const double2*>(vecI + k * MCC + 2);
                a[0][0] = fma(xv.x, d01.x, a[0][0]); a[0][1] = fma(xv.y, d01.x, a[0][1]);
                a[1][0] = fma(xv.x, d01.y, a[1][0]); a[1][1] = fma(xv.y, d01.y, a[1][1]);
                a[2][0] = fma(xv.x, d23.x, a[2][0]); a[2][1] = fma(xv.y, d23.x, a[2][1]);
                a[3][0] = fma(xv.x, d23.y, a[3][0]); a[3][1] = fma(xv.y, d23.y, a[3][1]);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int ii = i + r;
                double un[MCC], wn[MCC], Hn_[MCC];
#pragma unroll
                for (int q = 0; q < MCC; ++q) {
                    const double vd = a[q][r];
                    double uq;
                    if (scratch[q]) uq = vd;
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
        // partial sums of the row pass: one wave reduction per value
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
        __syncthreads();                         // Hi, wi, ui and the partial sums complete
#pragma unroll
        for (int q = 0; q < MCC; ++q) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int wv = 0; wv < MCC; ++wv) {
                a0 += red[wv * 32 + q * 4 + 0]; a1 += red[wv * 32 + q * 4 + 1];
                a2 += red[wv * 32 + q * 4 + 2]; a3 = fmax(a3, red[wv * 32 + q * 4 + 3]);
            }
            oS[q] = a0; odH[q] = a1; oHn[q] = a2; owm[q] = a3;
        }
    };

    // ------------------------------------------------------------------
    // fused pass over V (row major), once per round, for the state the row
    // pass just produced:  h_q = V^T H_q  (all columns, VALU) and the Gram
    // matrices W_q = V_a^T diag(w_q) V_a of the NT leading 16-column tiles
    // (matrix cores, see mxe_kernel.hip.h).  A lane holds
    // V[i0 + (l>>4)][16 t + (l&15)] for the four tiles t of a 4-row group --
    // the MFMA operand layout -- and uses the same registers for both.
    // Partial Gram tiles go to Wm[chain] in four rotating phases; h is
    // reduced over the lane groups and the waves; rho and |rho|^2 follow.
    // ------------------------------------------------------------------
    auto fused_pass = [&](auto NTTag, double (&or2)[MCC]) {
        constexpr int NT = decltype(NTTag)::value;
        constexpr int NPAIR = NT * (NT + 1) / 2;
        constexpr int DEPTH = 2;
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
        const int n_groups = nwp >> 2;
        const double* Vl = V + (size_t)kq * NP + cn;
        double f[DEPTH][4];
        double2 hw[DEPTH][4];                    // [0,1] = H of chains 01 / 23, [2,3] = w of chains 01 / 23
        auto load_group = [&](int d, int gidx) {
            const int i0 = 4 * gidx;
            const double2* hptr = reinterpret_cast<const double2*>(Hi + (size_t)(i0 + kq) * MCC);
            const double2* wptr = reinterpret_cast<const double2*>(wi + (size_t)(i0 + kq) * MCC);
            hw[d][0] = hptr[0]; hw[d][1] = hptr[1]; hw[d][2] = wptr[0]; hw[d][3] = wptr[1];
#pragma unroll
            for (int t = 0; t < 4; ++t) f[d][t] = Vl[(size_t)i0 * NP + 16 * t];
        };
        int gidx = wave;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if (gidx + d * MCC < n_groups) load_group(d, gidx + d * MCC);
        for (; gidx < n_groups; gidx += DEPTH * MCC) {
            double fc[DEPTH][4];
            double2 hc[DEPTH][4];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { fc[d][t] = f[d][t]; hc[d][t] = hw[d][t]; }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int gn = gidx + (DEPTH + d) * MCC;
                if (gn < n_groups) load_group(d, gn);
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (gidx + d * MCC < n_groups) {
                    const double Hq[MCC] = {hc[d][0].x, hc[d][0].y, hc[d][1].x, hc[d][1].y};
                    const double wq[MCC] = {hc[d][2].x, hc[d][2].y, hc[d][3].x, hc[d][3].y};
#pragma unroll
                    for (int c = 0; c < MCC; ++c) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) hp[c][t] = fma(fc[d][t], Hq[c], hp[c][t]);
                        double a[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) a[t] = fc[d][t] * wq[c];
                        int pr = 0;
#pragma unroll
                        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                            for (int nt = mt; nt < NT; ++nt) {
                                acc[c][pr] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], fc[d][nt], acc[c][pr], 0, 0, 0);
                                ++pr;
                            }
                    }
                }
            }
        }
        // h: sum the four row-residue lane groups, then the waves
#pragma unroll
        for (int c = 0; c < MCC; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double v_ = hp[c][t];
                v_ += __shfl_xor(v_, 16, WAVE);
                v_ += __shfl_xor(v_, 32, WAVE);
                if (kq == 0) hpart[(wave * MCC + c) * NP + 16 * t + cn] = v_;
            }
        // Gram tiles: four rotating phases (wave wv adds into chain (wv + phase) mod 4)
        for (int ph = 0; ph < MCC; ++ph) {
#pragma unroll
            for (int c = 0; c < MCC; ++c) {
                if (((c - wave) & (MCC - 1)) == ph) {
                    double* Wq = Wm + (size_t)c * NA * LD;
                    int pr = 0;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int nt = mt; nt < NT; ++nt) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * mt + kq + 4 * r, col = 16 * nt + cn;
                                if (row < NA && col < NA) {
                                    if (ph == 0) Wq[row * LD + col] = acc[c][pr][r];
                                    else Wq[row * LD + col] += acc[c][pr][r];
                                }
                            }
                            ++pr;
                        }
                }
            }
            __syncthreads();
        }
        // rho of chain q on its home wave (hpart complete after the first phase barrier)
        {
            const int q = wave, k = lane;
            double h = 0.0;
#pragma unroll
            for (int wv = 0; wv < MCC; ++wv) h += hpart[(wv * MCC + q) * NP + k];
            const double r = (k < ns) ? cc[k] * h - gh[q * NP + k] : 0.0;
            rho[q * NP + k] = r;
            const double r2 = wave_sum(r * r);
            if (lane == 0) red[wave * 32 + 16] = r2;       // slot 16 of wave q = |rho_q|^2
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MCC; ++q) or2[q] = red[q * 32 + 16];
        __syncthreads();                         // red / hpart free again
    };

    auto fused = [&](int n_cover, double (&or2)[MCC]) {
        const int ntile = (n_cover + 15) >> 4;
        if (ntile <= 1) fused_pass(std::integral_constant<int, 1>{}, or2);
        else if (ntile == 2 || NA <= 32) fused_pass(std::integral_constant<int, 2>{}, or2);
        else fused_pass(std::integral_constant<int, (NA > 32 ? 3 : 2)>{}, or2);
    };

    // ------------------------------------------------------------------
    // home wave: register Cholesky of chain q's active block + solve
    // (identical to chol_solve_reg of mxe_kernel.hip.h, wave local)
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

    // columns the next Newton block of chain q can need (alpha may advance to the next one)
    auto cover_of = [&](int q, double wm) -> int {
        const double an = (ia[q] + 1 < clen[q]) ? p.alpha[(size_t)prob0[q] + ia[q] + 1] : alpha[q];
        const double thr = p.theta * fmin(an, alpha[q]) / fmax(wm, 1e-300);
        int cnt = 0;
        for (int k = 0; k < ns; ++k) cnt += (cc[k] * cc[k] > thr) ? 1 : 0;
        if (p.theta <= 0.0) cnt = ns;
        return min(cnt + 1, NA);
    };

    // ------------------------------------------------------------------
    // round 0: evaluation from scratch for every chain
    // ------------------------------------------------------------------
    {
        double oS[MCC], odH[MCC], oHn[MCC], or2[MCC], owm[MCC];
        eval_pass(oS, odH, oHn, owm);
        int cover = 1;
#pragma unroll
        for (int q = 0; q < MCC; ++q) if (active[q]) cover = max(cover, cover_of(q, owm[q]));
        fused(cover, or2);
#pragma unroll
        for (int q = 0; q < MCC; ++q) {
            chi2[q] = or2[q] + cperp[q]; S[q] = oS[q]; Hn2[q] = oHn[q]; wmax[q] = owm[q];
            scratch[q] = false;
            if (active[q]) { nevals[q] = 1; Q[q] = 0.5 * chi2[q] - alpha[q] * S[q]; }
        }
    }

    long long guard = 0;
    bool exhausted = !dynamic;
    const long long guard_max = (long long)(dynamic ? x.n_queue : 1) * p.n_alpha * (p.maxiter + 64) + 64;
    while ((active[0] || active[1] || active[2] || active[3] || !exhausted) && guard++ < guard_max) {
        // ---- 0. dynamic layout: idle slots take the next piece from the queue ----
        if (dynamic && !exhausted && !(active[0] && active[1] && active[2] && active[3])) {
            bool idle_q = false;
#pragma unroll
            for (int q = 0; q < MCC; ++q) if (q == wave) idle_q = !active[q];
            if (lane == 0) {
                int c = -2;                               // -2: slot busy
                if (idle_q) { const int idx = atomicAdd(x.counter, 1); c = (idx < x.n_queue) ? x.queue[idx] : -1; }
                s_claim[wave] = c;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MCC; ++q) {
                const int c = s_claim[q];
                if (c == -1) exhausted = true;
                if (c >= 0) {
                    chain[q] = c;
                    const int e = p.chain_elem[c];
                    elem[q] = e; kind[q] = p.elem_kind[e]; cperp[q] = p.cperp[e];
                    step_lim[q] = p.step_max * p.sumD[e]; Dg[q] = p.D + (size_t)e * nwp;
                    active[q] = true; scratch[q] = true;
                    ia[q] = 0; n_iter[q] = 0; nevals[q] = 0; it_alpha[q] = 0; nact_last[q] = 0;
                    prob0[q] = p.chain_prob0[c]; clen[q] = p.chain_len[c];
                    alpha[q] = p.alpha[(size_t)prob0[q]];
                    mu[q] = 0.0; Qprev[q] = __builtin_nan("");
                    if (wave == q) {
                        gh[q * NP + lane] = p.ghat[(size_t)e * NP + lane];
                        vv[q * NP + lane] = p.v0[(size_t)p.chain_v0[c] * NP + lane];
                    }
                }
            }
            __syncthreads();
        }
        // ---- 1. right-hand sides and active blocks (home waves) ----
        {
            const int q = wave, k = lane;
            const double a0 = alpha[0], a1 = alpha[1], a2 = alpha[2], a3 = alpha[3];
            const double aq = (q == 0) ? a0 : (q == 1) ? a1 : (q == 2) ? a2 : a3;
            const double w0 = wmax[0], w1 = wmax[1], w2 = wmax[2], w3 = wmax[3];
            const double wq = (q == 0) ? w0 : (q == 1) ? w1 : (q == 2) ? w2 : w3;
            rhs[q * NP + k] = (k < ns) ? fma(aq * vv[q * NP + k], ci[k], rho[q * NP + k]) : 0.0;
            const double thr = p.theta * aq / fmax(wq, 1e-300);
            int cnt = (k < ns && cc[k] * cc[k] > thr) ? 1 : 0;
            int na = (int)wave_sum((double)cnt);
            if (p.theta <= 0.0) na = ns;
            na = max(1, min(na, NA));
            if (lane == 0) s_nact[q] = na;
        }
        __syncthreads();
        int nact[MCC], namax = 1;
#pragma unroll
        for (int q = 0; q < MCC; ++q) {
            nact[q] = s_nact[q];
            if (active[q] && !scratch[q]) namax = max(namax, nact[q]);
        }

        MXE_STAMP(0);
        MXE_STAMP(1);
        // ---- 3. home wave: factorise, solve, step, Bryan's bound ----
        {
            const int q = wave, k = lane;
            bool act_q = false, scr_q = false; double aq = 1.0, muq = 0.0, lim = 0.0; int na = 1;
#pragma unroll
            for (int c = 0; c < MCC; ++c) if (c == q) {
                act_q = active[c]; scr_q = scratch[c]; aq = alpha[c]; muq = mu[c]; lim = step_lim[c]; na = nact[c];
            }
            int okflag = 0;
            double dk = 0.0;
            if (act_q && scr_q) {
                dk = vv[q * NP + k];                 // evaluation from scratch: operand is v
            } else if (act_q) {
                // damping loop of the home wave: raise mu until the factorisation
                // succeeds and Bryan's bound holds (no evaluation needed for that)
                while (true) {
                    const double a = aq + muq;
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
                        if (nrm <= lim) { okflag = 1; dk = (k < ns) ? cc[k] * z : 0.0; break; }
                    }
                    muq = (muq == 0.0) ? p.mu_first * aq : muq * p.mu_grow;
                    if (!(muq <= p.mu_max * aq)) break;
                }
                if (lane == 0) s_mu[q] = muq;
            }
            dlc[q * NP + k] = (okflag ? dk : 0.0);
            vecI[k * MCC + q] = dk;
            if (lane == 0) s_ok[q] = okflag;
        }
        __syncthreads();
        bool okq[MCC];
#pragma unroll
        for (int q = 0; q < MCC; ++q) { okq[q] = s_ok[q] != 0; if (active[q] && !scratch[q]) mu[q] = s_mu[q]; }

        MXE_STAMP(2);
        // ---- 4. shared evaluation pass ----
        double oS[MCC], odH[MCC], oHn[MCC], or2[MCC], owm[MCC];
        eval_pass(oS, odH, oHn, owm);
        MXE_STAMP(3);
        {
            int cover = 1;
#pragma unroll
            for (int q = 0; q < MCC; ++q) if (active[q]) cover = max(cover, cover_of(q, owm[q]));
            fused(cover, or2);
        }

        MXE_STAMP(4);
        // ---- 5. accept / converge / advance (scalars in every thread, LDS by home wave) ----
#pragma unroll
        for (int q = 0; q < MCC; ++q) {
            if (!active[q]) continue;
            const double chi2t = or2[q] + cperp[q], St = oS[q];
            const double Qt = 0.5 * chi2t - alpha[q] * St;
            const bool finite = fabs(Qt) <= 1.7e308;
            bool finish_alpha = false; int conv = 0;
            if (scratch[q]) {
                // state restored from v (or first evaluation); keep the damping
                if (finite) { scratch[q] = false; chi2[q] = chi2t; S[q] = St; Hn2[q] = oHn[q]; wmax[q] = owm[q]; Q[q] = Qt; }
                else { finish_alpha = true; }                   // cannot even evaluate: give up on this alpha
                ++nevals[q];
            } else if (!okq[q]) {
                // the damping loop ran out of range: give up on this alpha
                finish_alpha = true;
            } else if (!finite || (mu[q] > 0.0 && Qt > Q[q])) {
                // not finite, or a damped step that made Q worse: more damping
                ++nevals[q];
                mu[q] = (mu[q] == 0.0) ? p.mu_first * alpha[q] : mu[q] * p.mu_grow;
                scratch[q] = true;                              // u, w were overwritten: restore from v
                if (!(mu[q] <= p.mu_max * alpha[q])) finish_alpha = true;
            } else {
                // accepted
                ++nevals[q];
                relH[q] = sqrt(odH[q] / Hn2[q]);
                if (wave == q) vv[q * NP + lane] -= dlc[q * NP + lane];
                chi2[q] = chi2t; S[q] = St; Hn2[q] = oHn[q]; wmax[q] = owm[q];
                Qprev[q] = Q[q]; Q[q] = Qt; mu[q] = 0.0;
                nact_last[q] = nact[q];
                ++n_iter[q]; ++it_alpha[q];
                if (p.tol_h > 0.0 && relH[q] < p.tol_h && n_iter[q] > p.miniter) { conv = 1; finish_alpha = true; }
                else if (p.tol_relq > 0.0 && fabs(fabs(Qprev[q] - Q[q]) / Q[q]) < p.tol_relq && n_iter[q] > p.miniter) { conv = 1; finish_alpha = true; }
                else if (n_iter[q] >= p.maxiter) finish_alpha = true;
            }
            if (finish_alpha) {
                const size_t prob = (size_t)prob0[q] + ia[q];
                if (wave == q) {
                    if (p.out_H) {
                        double* Ho = p.out_H + prob * nw;
                        for (int i = lane; i < nw; i += 64) {
                            const double Di = Dg[q][i], uq = ui[i * MCC + q];
                            Ho[i] = (kind[q] == 0) ? Di * exp(uq) : Di * exp(uq) - Di * exp(-uq);
                        }
                    }
                    if (p.out_v) p.out_v[prob * NP + lane] = vv[q * NP + lane];
                    if (lane == 0) {
                        p.out_chi2[prob] = chi2[q]; p.out_S[prob] = S[q]; p.out_Q[prob] = Q[q];
                        p.out_niter[prob] = n_iter[q]; p.out_conv[prob] = conv;
                        p.out_nevals[prob] = nevals[q]; p.out_nact[prob] = nact_last[q];
                    }
                }
                ++ia[q];
                n_iter[q] = 0; nevals[q] = 0; it_alpha[q] = 0; mu[q] = 0.0;
                Qprev[q] = __builtin_nan("");
                if (ia[q] >= clen[q]) active[q] = false;
                else {
                    alpha[q] = p.alpha[(size_t)prob0[q] + ia[q]];
                    Q[q] = 0.5 * chi2[q] - alpha[q] * S[q];
                }
            }
        }
        __syncthreads();                         // vv updates visible before the next round
        MXE_STAMP(5);
        if (tid == 0) prof_rounds_inc();
    }
#ifdef MXE_PROFILE
    if (tid == 0 && p.prof) { for (int r = 0; r < 7; ++r) p.prof[(size_t)blockIdx.x * 8 + r] = prof_acc[r]; p.prof[(size_t)blockIdx.x * 8 + 7] = prof_rounds; }
#endif
}

} // namespace mxe
