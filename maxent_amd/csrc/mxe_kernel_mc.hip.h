// mxe_kernel_mc.hip.h -- four alpha chains per workgroup, in lock-step
//
// Same mathematics as mxe_kernel.hip.h (see there and DESIGN.md).  What is
// different is how the work is laid on the CU:
//
//  * a workgroup of 4 wavefronts owns 4 chain slots that share one data set
//    (one V).  Every load of V feeds all four chains: the row pass streams
//    V^T once (du = V delta of the four steps, v_mfma_f64_4x4x4 with the four
//    slots as the columns of every block), the fused pass streams V once and
//    produces both h = V^T H (v_mfma_f64_4x4x4) and the Gram matrices
//    W = V_a^T diag(w) V_a (v_mfma_f32_16x16x4: they only precondition the step)
//    of the four trial points from the same registers;
//  * wave q is the "home" of slot q: it solves the slot's Newton system
//    (Gauss-Jordan elimination in registers), takes the step, decides acceptance
//    / convergence and writes the results -- four solves run side by side on the
//    four SIMDs; the slot's scalars live in LDS between the home wave's sections;
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
// rows of V behind the last omega row (and LDS doubles behind H) that the look-ahead of the
// fused pass may read without using them: 2 * DEPTH * 4 waves * 4 rows, rounded up
// allowance for the binary32 Gram tiles in the stopping estimate.  Measured on the cfg4 batch: with 0 and
// with 2e-5 the worst sampled alpha-solves end 1.3e-9 from the fixed point (2.3e-10 with binary64 tiles;
// tol_h = 1e-9) -- the same solves either way -- and 2e-5 costs 2 % more iterations, so none is made.
constexpr double MC_GRAM_ERR = 0.0;
constexpr int MC_LOOKAHEAD_ROWS = 512;
constexpr int MC_LOOKAHEAD_LDS = (8 * 4 + 4) * 4 * 4 + 64;       // doubles
template <int NA, int NWV>
#ifndef MXE_X_WGPC
#define MXE_X_WGPC 1        // experiment: workgroups per CU the register budget is sized for
#endif
__global__ __launch_bounds__(64 * NWV, MXE_X_WGPC)
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
    double* ecor = dlc + MCC * NP;               // [MCC][NP]   defect of the first Newton iterate of the previous alpha (predictor)
    double* eacc = ecor + MCC * NP;              // [MCC][NP]   ... of this alpha, being accumulated
    double* cc   = eacc + MCC * NP;              // [NP]
    double* ci   = cc + NP;                      // [NP]
    double* vecI = ci + NP;                      // [NP][MCC]   step / v, chain minor (row-pass operand)
    double* hpart = vecI + NP * MCC;             // [NWV waves][MCC chains][NP]
    double* red  = hpart + NWV * MCC * NP;       // [NWV waves][32]
    double* ui   = red + NWV * 32;               // [nwp][MCC]
    double* wi   = ui + (size_t)nwp * MCC;       // [nwp][MCC]
    float*  wiF  = reinterpret_cast<float*>(wi + (size_t)nwp * MCC);   // [nwp][MCC] binary32 copy of w (Gram operand)
    double* Hi   = wi + (size_t)nwp * MCC + (size_t)nwp * MCC / 2;    // [nwp][MCC]
    __shared__ int s_elem[MCC], s_kind[MCC], s_act[MCC], s_scr[MCC];

    // ---- slot state.  It is owned by the home wave, which loads it from LDS at the
    //      start of its two sections of a round and stores it back at their end, so
    //      that no register is pinned by it during the two streaming passes. ----
    struct Slot {
        double alpha, mu, chi2, S, Hn2, wmax, Q, Qprev, cperp, steplim, muh;    // muh: damping the last damped step of this piece needed
        int elem, prob0, clen, ia, niter, nevals, nact, active, scratch, okprev, bt, capp;
    };
    // the alphas of a slot's piece (a dependent global load in the accept step costs its full latency)
    constexpr int ACAP = 128;
    __shared__ double s_alpha[MCC][ACAP];
    __shared__ double s_sd[MCC][12];
    __shared__ int s_si[MCC][12];
    auto load_slot = [&](Slot& t) {
        const double* d = s_sd[wave]; const int* n = s_si[wave];
        t.alpha = d[0]; t.mu = d[1]; t.chi2 = d[2]; t.S = d[3]; t.Hn2 = d[4]; t.wmax = d[5];
        t.Q = d[6]; t.Qprev = d[7]; t.cperp = d[8]; t.steplim = d[9]; t.muh = d[10];
        t.elem = n[0]; t.prob0 = n[1]; t.clen = n[2]; t.ia = n[3]; t.niter = n[4]; t.nevals = n[5];
        t.nact = n[6]; t.active = n[7]; t.scratch = n[8]; t.okprev = n[9]; t.bt = n[10]; t.capp = n[11];
    };
    auto store_slot = [&](const Slot& t) {
        if (lane == 0) {
            double* d = s_sd[wave]; int* n = s_si[wave];
            d[0] = t.alpha; d[1] = t.mu; d[2] = t.chi2; d[3] = t.S; d[4] = t.Hn2; d[5] = t.wmax;
            d[6] = t.Q; d[7] = t.Qprev; d[8] = t.cperp; d[9] = t.steplim; d[10] = t.muh;
            n[0] = t.elem; n[1] = t.prob0; n[2] = t.clen; n[3] = t.ia; n[4] = t.niter; n[5] = t.nevals;
            n[6] = t.nact; n[7] = t.active; n[8] = t.scratch; n[9] = t.okprev; n[10] = t.bt; n[11] = t.capp;
            s_act[wave] = t.active; s_scr[wave] = t.scratch;
        }
        wave_sync();
    };
    auto start_piece = [&](Slot& t, int c) {     // home wave: take chain (piece) c into this slot
        t.elem = p.chain_elem[c];
        t.cperp = p.cperp[t.elem];
        t.steplim = p.step_max * p.sumD[t.elem];
        t.prob0 = p.chain_prob0[c]; t.clen = p.chain_len[c];
        t.ia = 0; t.niter = 0; t.nevals = 0; t.nact = 0; t.okprev = 0; t.bt = 0; t.capp = 0;
        for (int i = lane; i < min(t.clen, ACAP); i += 64) s_alpha[wave][i] = p.alpha[(size_t)t.prob0 + i];
        t.alpha = p.alpha[(size_t)t.prob0];
        t.mu = 0.0; t.muh = 0.0; t.Qprev = __builtin_nan("");
        t.chi2 = 0.0; t.S = 0.0; t.Hn2 = 1.0; t.wmax = 1.0; t.Q = 0.0;
        t.active = 1; t.scratch = 1;
        gh[wave * NP + lane] = p.ghat[(size_t)t.elem * NP + lane];
        vv[wave * NP + lane] = p.v0[(size_t)p.chain_v0[c] * NP + lane];
        ecor[wave * NP + lane] = 0.0; eacc[wave * NP + lane] = 0.0;
        if (lane == 0) { s_elem[wave] = t.elem; s_kind[wave] = p.elem_kind[t.elem]; }
    };

    auto alpha_at = [&](const Slot& t, int i) -> double {
        return (t.clen <= ACAP) ? s_alpha[wave][i] : p.alpha[(size_t)t.prob0 + i];
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
            // empty slot: evaluates v = 0 of a neighbour's element every round (finite, never used)
            t = Slot{1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0};
            gh[wave * NP + lane] = 0.0; vv[wave * NP + lane] = 0.0;
            if (lane == 0) { s_elem[wave] = -1; s_kind[wave] = 0; }
        }
        store_slot(t);
        dlc[wave * NP + lane] = 0.0;
    }
    __syncthreads();
    int any_elem = -1;
#pragma unroll
    for (int q = 0; q < MCC; ++q) if (s_elem[q] >= 0 && any_elem < 0) any_elem = s_elem[q];
    if (any_elem < 0) return;                    // nothing for this workgroup
    const int ds = __builtin_amdgcn_readfirstlane(p.elem_ds[__builtin_amdgcn_readfirstlane(any_elem)]);     // wave-uniform: V, Vt become scalar base pointers
    const double* __restrict__ V  = p.V  + (size_t)ds * nwp * NP;
    const double* __restrict__ Vt = p.Vt + (size_t)ds * NP * nwp;
    if (wave == 0) { cc[lane] = p.c[ds * NP + lane]; ci[lane] = p.cinv[ds * NP + lane]; }
    __syncthreads();

#ifdef MXE_PROFILE
    // per-wave stamps (diagnostic build only): row = workgroup * 8 + wave
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
    long long prof_rounds = 0;
#ifdef MXE_PROFILE_HOME          // split the home phase instead: everything else goes to slot 5
#define MXE_STAMPW(idx) do { const long long t__ = clock64(); prof_acc[(idx) == 2 ? 2 : 5] += t__ - prof_t; prof_t = t__; } while (0)
#define MXE_STAMPH(idx) do { const long long t__ = clock64(); prof_acc[idx] += t__ - prof_t; prof_t = t__; } while (0)
#else
#define MXE_STAMPW(idx) do { const long long t__ = clock64(); prof_acc[idx] += t__ - prof_t; prof_t = t__; } while (0)
#define MXE_STAMPH(idx) do {} while (0)
#endif
#else
#define MXE_STAMPW(idx) do {} while (0)
#define MXE_STAMPH(idx) do {} while (0)
#endif

    // ------------------------------------------------------------------
    // home wave: the slot's Newton system  (c W c + a I) z = rhs  on the active block, in registers.
    // Lane i holds the FULL row i of the symmetric matrix (N doubles, static indices: the j and k
    // loops are fully unrolled) and its right-hand side.  Gauss-Jordan elimination without pivoting
    // (the matrix is positive definite): for pivot j every other lane subtracts f = A_ij / A_jj times
    // row j -- broadcast from lane j with v_readlane, eight entries at a time -- from the columns
    // k > j of its row and from its right-hand side.  A wave issues the same instructions for the
    // lanes above the pivot as a Cholesky factorisation does for the lanes below it alone, so the
    // elimination costs what the factorisation cost, and when it ends z_i = b_i / A_ii: no transposed
    // factor through LDS, no back substitution (they were 4-5.6 k cycles of the 10-16 k per slot).
    // The solve only preconditions the (inexact) Newton step; rows >= n_act are identity rows.
    // ------------------------------------------------------------------
    auto gj_home = [&](auto NTag, double a, int n_act) -> bool {
        constexpr int N = decltype(NTag)::value;
        const int q = wave, i = lane;
        const double* Wq = Wm + (size_t)q * NA * LD;
        const double* rq = rhs + q * NP;
        bool ok = true;
        const bool live = i < n_act;
        const double ci_ = live ? cc[i] : 0.0;
        double A[N];
        {
            // W is kept as upper triangle + diagonal: entry (i, j) sits at [min][max].  All N loads are
            // issued back to back (clamped lane index, selected afterwards)
            const int ic = min(i, N - 1);
            double wr[N];
#pragma unroll
            for (int j = 0; j < N; ++j) wr[j] = Wq[min(j, ic) * LD + max(j, ic)];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double cj = wave_bcast(ci_, j);        // = c_j for j < n_act, else 0
                double xv = ci_ * wr[j] * cj;                // 0 in the rows and columns >= n_act
                if (j == i) xv = live ? xv + a : 1.0;
                A[j] = xv;
            }
        }
        double b = live ? rq[i] : 0.0;
        double dinv_i = 1.0;
        MXE_STAMPH(1);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double piv = wave_bcast(A[j], j);
            if (!(piv > 0.0)) ok = false;
            double inv = __builtin_amdgcn_rcp(piv);
            inv = fma(fma(-piv, inv, 1.0), inv, inv);
            if (i == j) dinv_i = inv;
            const double f = (i != j) ? A[j] * inv : 0.0;    // multiplier of row j for this lane's row
            b = fma(-f, wave_bcast(b, j), b);
#pragma unroll
            for (int k0 = j + 1; k0 < N; k0 += 8) {
                double rk[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) if (k0 + r < N) rk[r] = wave_bcast(A[k0 + r], j);     // row j, column k
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 8; ++r) if (k0 + r < N) A[k0 + r] = fma(-f, rk[r], A[k0 + r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        MXE_STAMPH(3);
        if (ok && live) zz[q * NP + i] = b * dinv_i;
        MXE_STAMPH(4);
#ifdef MXE_PROFILE_HOME
        prof_acc[6] += 1;                        // solves (slot 6 is a count in this build)
#endif
        return ok;
    };

    long long guard = 0;
    const long long guard_max = (long long)(dynamic ? x.n_queue : 1) * p.n_alpha * (p.maxiter + 64) + 64;

    while (guard++ < guard_max) {
        // (a slot that finishes its piece takes the next one from the queue right away, in step 4)
        if (!(s_act[0] | s_act[1] | s_act[2] | s_act[3])) break;      // uniform: LDS flags behind a barrier

        // ---- 1. home wave: right-hand side, active block, factorise, solve, step ----
        if (wave < MCC) {
            const int q = wave, k = lane;
            Slot t;
            load_slot(t);
            int okflag = 0;
            double dk = 0.0;
            double dtot = 0.0;                       // total step from v of the trial point (dk: operand of the row pass)
            if (t.active && t.scratch) {
                dk = vv[q * NP + k];                 // evaluation from scratch: the operand is v
            } else if (t.active && t.okprev == 3) {
                // backtracking: the last trial v - delta made Q worse; the next one is v - delta / 2,
                // reached from the trial state in LDS by the step -delta / 2.  No factorisation, and
                // no round spent on restoring the state from v.
                dtot = 0.5 * dlc[q * NP + k];
                dk = -dtot;
                okflag = 4;
            } else if (t.active) {
                rhs[q * NP + k] = (k < ns) ? fma(t.alpha * vv[q * NP + k], ci[k], rho[q * NP + k]) : 0.0;
                const double thr = p.theta * t.alpha / fmax(t.wmax, 1e-300);
                const unsigned long long m = __ballot(k < ns && cc[k] * cc[k] > thr);
                int na = (p.theta > 0.0) ? __popcll(m) : ns;
                na = max(1, min(na, NA));
                t.nact = na;
                wave_sync();
                MXE_STAMPH(0);
                // damping loop: raise mu until the factorisation succeeds and Bryan's bound holds
                while (true) {
                    const double a = t.alpha + t.mu;
                    bool ok;
                    if (na <= 16) ok = gj_home(std::integral_constant<int, 16>{}, a, na);
                    else if (na <= 20) ok = gj_home(std::integral_constant<int, 20>{}, a, na);
                    else if (na <= 24) ok = gj_home(std::integral_constant<int, 24>{}, a, na);
                    else if (na <= 28) ok = gj_home(std::integral_constant<int, 28>{}, a, na);
                    else if (NA <= 32 || na <= 32) ok = gj_home(std::integral_constant<int, 32>{}, a, na);
                    else ok = gj_home(std::integral_constant<int, (NA > 32 ? NA : 32)>{}, a, na);
                    if (ok) {
                        double z = 0.0, nrm = 0.0;
                        if (k < na) { z = zz[q * NP + k]; nrm = z * (rhs[q * NP + k] - a * z); }
                        else if (k < ns) z = rhs[q * NP + k] / a;
                        nrm = wave_sum(nrm);
                        if (nrm <= t.steplim) {
                            okflag = 1; dk = (k < ns) ? cc[k] * z : 0.0;
#ifndef MXE_X_NO_PREDICTOR
                            // Predictor along the alpha path.  The first Newton iterate of an alpha, started
                            // from the solution of the previous one, misses the new solution by a defect
                            // e = O(h^2) (h = the step in log alpha) whose coefficient changes slowly along
                            // the path, and the defect of the PREVIOUS alpha is known exactly: it is what the
                            // later iterations of that alpha added.  Adding it to the first step leaves
                            // O(h^3): most alphas then need two iterations instead of three.
                            // Safeguards: only a correction smaller than half the Newton step is used, and
                            // the corrected step must not increase Q (else it is halved like a shortened one).
                            if (t.niter == 0 && t.ia > 0 && t.mu == 0.0) {
                                const double ek = ecor[q * NP + k];
                                const double ne = wave_sum(ek * ek), nd = wave_sum(dk * dk);
                                if (ne > 0.0 && ne <= 0.25 * nd) { dk -= ek; okflag = 5; }
                            }
#endif
                            break;
                        }
#ifndef MXE_X_NO_STEP_SCALE
                        // An undamped Newton step that violates Bryan's bound is shortened onto it
                        // (same direction, a descent direction of Q) instead of being recomputed with
                        // damping: a second factorisation in this round would keep the other three
                        // slots of the workgroup waiting.  It is accepted like a damped step (Q must
                        // not increase); if it is not, the damped path below takes over.
                        if (t.mu == 0.0 && nrm < 1e300) {
                            const double sc = sqrt(t.steplim / nrm);
                            okflag = 2; dk = (k < ns) ? cc[k] * z * sc : 0.0; break;
                        }
#endif
                    }
                    t.mu = (t.mu == 0.0) ? p.mu_first * t.alpha : t.mu * p.mu_grow;
                    if (!(t.mu <= p.mu_max * t.alpha)) break;
                }
            }
            if (okflag != 4) dtot = okflag ? dk : 0.0;
            dlc[q * NP + k] = dtot;
            vecI[k * MCC + q] = dk;
            t.okprev = okflag;
            store_slot(t);
            MXE_STAMPH(0);
        }
        __syncthreads();
        MXE_STAMPW(2);

        // ---- 2. row pass (V^T once): u, w, H of the four trial points, in place ----
        // du = V delta of the four slots as v_mfma_f64_4x4x4 (four independent 4x4x4 blocks per
        // instruction): block b = omega rows 4b .. 4b+3 of a 16-row tile, columns = the four slots,
        // K = four singular directions.  Operand / result lanes (probed, tools/mfma_4x4x4_layout.hip):
        //   A[b][i][k] lane 16k + 4b + i,   B[b][k][j] lane 16k + 4b + j,   D[b][i][j] lane 16i + 4b + j
        // so A is one 8-byte load of V^T per lane (four 128-B row segments per instruction), B one
        // LDS read of the steps shared by every tile, and every lane ends up with ONE (row, slot)
        // element per tile for the exp / entropy part (all 64 lanes busy, no per-slot loop).
        {
            // the home waves are done with W / L of the previous iteration: zero it for the tile sums of step 3
            {
                static_assert((MCC * NA * LD) % 2 == 0, "W is zeroed with 16-byte stores");
                double2* Wz = reinterpret_cast<double2*>(Wm);
#pragma unroll
                for (int idx = 0; idx < (MCC * NA * LD / 2 + T - 1) / T; ++idx)
                    if (tid + idx * T < MCC * NA * LD / 2) Wz[tid + idx * T] = double2{0.0, 0.0};
            }
            const int j = lane & 3;                              // slot of this lane's results
            const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4);      // result row inside the tile
            const int ak = lane >> 4;                            // operand k inside the chunk
            const bool scr_j = s_scr[j] != 0;
            const bool pm_j = s_kind[j] != 0;
            const double* Dj = p.D + (size_t)((s_elem[j] >= 0) ? s_elem[j] : any_elem) * nwp;
            double pS = 0.0, pdH = 0.0, pHn = 0.0, pwm = 0.0, pdu = 0.0;
            const int nblk = nwp >> 5;                           // blocks of 32 omega rows = two 16-row tiles
            const int nchunk = (ns + 3) >> 2;                    // chunks of four singular directions
            constexpr int TB = 8;                                // tiles per batch (accumulators)
            constexpr int RD = 4;                                // ring depth in chunks
            constexpr int NCHK = NP / 4;
            // The two tiles of a block interleave: tile parity = omega parity, so that a lane's operands of
            // both come from ONE 16-byte load (V^T[4 kc + ak][32 blk + 2 (lane & 15) .. + 1]).  The loads of
            // this pass wait for L2 latency with a bounded number in flight (vmcnt): wider loads, not more.
            for (int b0 = wave; b0 < nblk; b0 += NWV * (TB / 2)) {
                // tile tt of the batch: block b0 + NWV (tt >> 1), parity tt & 1; past the end: block b0 again,
                // results dropped
                double acc[TB], Dv[TB], uo[TB], wo[TB];
                int rowt[TB];
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) {
                    const int blk = (b0 + NWV * (tt >> 1) < nblk) ? b0 + NWV * (tt >> 1) : b0;
                    const int row = 32 * blk + 2 * drow + (tt & 1);
                    rowt[tt] = row;
                    acc[tt] = 0.0;
                    Dv[tt] = Dj[row];
                    uo[tt] = ui[row * MCC + j];
                    wo[tt] = wi[row * MCC + j];
                }
                // V^T operand: row 4 kc + ak of V^T; the blocks of a batch are 32 NWV rows apart (V^T is
                // padded behind its last row for a partial batch)
                const double* ap = Vt + (size_t)ak * nwp + 32 * b0 + 2 * (lane & 15);
                const double* bp = vecI + ak * MCC + j;
                double xr[RD][TB];
                auto loadA = [&](double (&xv)[TB], int kc) {
                    const double* src = ap + (size_t)(4 * kc) * nwp;
#pragma unroll
                    for (int pp = 0; pp < TB / 2; ++pp) {
                        const double2 x2 = *reinterpret_cast<const double2*>(src + 32 * NWV * pp);
                        xv[2 * pp] = x2.x; xv[2 * pp + 1] = x2.y;
                    }
                };
#pragma unroll
                for (int r = 0; r < RD - 1; ++r) loadA(xr[r], min(r, NCHK - 1));
                for (int kc = 0; kc < nchunk; kc += RD) {
#pragma unroll
                    for (int r = 0; r < RD; ++r) {
                        loadA(xr[(r + RD - 1) % RD], min(kc + r + RD - 1, NCHK - 1));
                        const double bv = bp[min(kc + r, NCHK - 1) * 4 * MCC];
                        if (kc + r < nchunk) {
#pragma unroll
                            for (int tt = 0; tt < TB; ++tt)
                                acc[tt] = __builtin_amdgcn_mfma_f64_4x4x4f64(xr[r][tt], bv, acc[tt], 0, 0, 0);
                        }
                    }
                }
                MXE_STAMPW(0);
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) {
                    if (b0 + NWV * (tt >> 1) < nblk) {            // uniform
                        const int row = rowt[tt];
                        const double vd = acc[tt];
                        const double uq = scr_j ? vd : uo[tt] - vd;
                        const double tq = scr_j ? 0.0 : wo[tt] * vd;
                        pdH = fma(tq, tq, pdH);
                        pdu = fmax(pdu, scr_j ? 0.0 : fabs(vd));  // padded rows of V^T are zero
                        const double Di = Dv[tt];
                        const double ep = fast_exp(uq);
                        const double Hp = Di * ep;
                        double Hq = Hp, wq = Hp, Sq = Hp - Di - Hp * uq;
                        if (pm_j) {
                            const double Hm = Di * recip_exp(ep);
                            Hq = Hp - Hm; wq = Hp + Hm;
                            Sq += Hm - Di + Hm * uq;
                        }
                        if (row >= nw) { Hq = 0.0; wq = 0.0; Sq = 0.0; }
                        ui[row * MCC + j] = uq; wi[row * MCC + j] = wq; Hi[row * MCC + j] = Hq;
                        wiF[row * MCC + j] = (float)wq;
                        pS += Sq;
                        pHn = fma(Hq, Hq, pHn);
                        pwm = fmax(pwm, wq);                      // NaN-ignoring; non-finite states are caught through Q
                    }
                }
            }
            // sums over the lanes of equal slot (lane & 3): rotate by 4 and 8 inside the rows of
            // 16 lanes, then across the rows
            pS = slot_sum(pS); pdH = slot_sum(pdH); pHn = slot_sum(pHn);
            pwm = slot_max(pwm); pdu = slot_max(pdu);
            if (lane < MCC) {
                red[wave * 32 + lane * 8 + 0] = pS; red[wave * 32 + lane * 8 + 1] = pdH;
                red[wave * 32 + lane * 8 + 2] = pHn; red[wave * 32 + lane * 8 + 3] = pwm;
                red[wave * 32 + lane * 8 + 4] = pdu;
            }
        }
        MXE_STAMPW(3);
        __syncthreads();                         // Hi, wi, ui and the partial sums complete
        MXE_STAMPW(7);

        // ---- 3. fused pass (V once): h = V^T H and W = V_a^T diag(w) V_a, both on the matrix pipes ----
        {
            // The Gram matrix only preconditions the Newton step (the residual rho that defines the
            // answer comes from h, binary64): its tiles are accumulated on the binary32 matrix pipe
            // (v_mfma_f32_16x16x4_f32: 32 cycles per 16x16x4 instead of 64), operands converted from the
            // same binary64 registers of V that feed h; w comes as the binary32 copy the row pass wrote.
            // Relative error ~1e-6 of sqrt(W_ii W_jj), next to the decoupling threshold theta = 1e-5
            // that the Newton matrix carries anyway (MC_GRAM_ERR in the stopping estimate).
            // h_q = V^T H_q of the four slots is v_mfma_f64_4x4x4 (four blocks per instruction: block b =
            // columns 16 t + 4 b .. + 3 of V, the four slots as columns, K = the 4 omega rows of the
            // group) from the SAME operand registers: A[b][i][k] (lane 16 k + 4 b + i) = V[i0 + k][16 t + 4 b + i]
            // is exactly f[t]; B[b][k][j] (lane 16 k + 4 b + j) = H[i0 + k][slot j] is one 8-byte LDS read;
            // D[b][i][j] lands on lane 16 i + 4 b + j: one accumulator per tile and no cross-lane sum.
            typedef float g4 __attribute__((ext_vector_type(4)));
            g4 acc[MCC][NPAIR];
            double hp[4];
#pragma unroll
            for (int c = 0; c < MCC; ++c)
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) acc[c][pr] = g4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) hp[t] = 0.0;
            const int kq = lane >> 4, cn = lane & 15;
            const int n_groups = nwp >> 2;           // 4 omega rows per MFMA; multiple of 16
            const double* Vl = V + (size_t)kq * NP + cn;
            struct HW { double h; float4 w; };       // H of slot (lane & 3), w of the four slots, row i0 + kq
            struct OP { float ff[NT]; float a[MCC][NT]; };    // binary32 Gram operands of one row group
            auto loadHW = [&](HW& hw, const double* hsrc, const float* wsrc) {
                hw.h = hsrc[0];
                hw.w = *reinterpret_cast<const float4*>(wsrc);
            };
            auto prep = [&](OP& o, const double (&f)[4], const HW& hw) {
                const float wq[MCC] = {hw.w.x, hw.w.y, hw.w.z, hw.w.w};
#pragma unroll
                for (int t = 0; t < NT; ++t) o.ff[t] = (float)f[t];
#pragma unroll
                for (int c = 0; c < MCC; ++c)
#pragma unroll
                    for (int t = 0; t < NT; ++t) o.a[c][t] = o.ff[t] * wq[c];
            };
            auto mma = [&](const OP& o, const double (&f)[4], const HW& hw) {
#ifdef MXE_X_NO_GRAM      // timing experiment only (results are wrong)
                acc[0][0][0] += o.a[0][0] + o.a[1][1] + o.a[2][0] + o.a[3][1];
#else
#pragma unroll
                for (int c = 0; c < MCC; ++c) {
                    int pr = 0;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int nt = mt; nt < NT; ++nt) {
                            acc[c][pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[c][mt], o.ff[nt], acc[c][pr], 0, 0, 0);
                            ++pr;
                        }
                }
#endif
#pragma unroll
                for (int t = 0; t < 4; ++t) hp[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[t], hw.h, hp[t], 0, 0, 0);
            };
            // The waves take groups wave, wave + NWV, ...  Software pipeline, no branch inside: the 12 + 4
            // MFMAs of group j back to back, then the vector instructions that build the binary32
            // operands of group j + 1 (on gfx950 a wave's vector instructions do not issue in the shadow
            // of its own MFMAs, tools/mfma_shadow.hip; a row group costs 464 cycles of MFMA plus its
            // vector and memory instructions, ~720 in all).  V comes through a ring of DEPTH register sets
            // (DEPTH - 1 row groups in flight), H / w from LDS two groups ahead.  n_groups is a multiple
            // of DEPTH * NWV (n_omega_pad is a multiple of 128) and every load is unconditional: the
            // look-ahead past the end reads the zero rows behind V and the padding behind H in LDS, and
            // what is prepared from them is never multiplied.
            constexpr int ST = NWV;
            constexpr int DEPTH = (NWV == 4 && MXE_X_WGPC == 1) ? 8 : 4;
            static_assert(DEPTH % 4 == 0, "the H / w ring of four is indexed statically across trips");
            int g = wave;
            {
                double fr[DEPTH][4];
                HW hr[4];
                OP op[2];
                // V through buffer loads: one resource descriptor for the data set's V (wave-uniform), one
                // 32-bit lane offset, the position of the row group as a SCALAR offset -- the loop carries
                // no 64-bit vector address arithmetic (it was four v_add_co / v_addc pairs with their
                // hazard slots per row group)
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, 0x7fffffff, 0x00020000);
                const int loff = (kq * NP + cn) * 8;                  // bytes
                auto loadV = [&](double (&f)[4], int soff_bytes) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const u2v v2 = __builtin_amdgcn_raw_buffer_load_b64(vrsrc, loff + 128 * t, soff_bytes, 0);
                        f[t] = __hiloint2double((int)v2.y, (int)v2.x);
                    }
                };
                constexpr size_t VSTEP = (size_t)4 * ST * NP;        // doubles per group step (V)
                constexpr int HSTEP = 4 * ST * MCC;                  // ... (H, w in LDS)
                int vp = 4 * __builtin_amdgcn_readfirstlane(g) * NP * 8;          // byte offset of the wave's first row group (uniform)
                const double* hb = Hi + (size_t)(4 * g + kq) * MCC + (lane & 3);
                const float* wb = wiF + (size_t)(4 * g + kq) * MCC;
#pragma unroll
                for (int j = 0; j < DEPTH - 1; ++j) loadV(fr[j], vp + j * (int)(VSTEP * 8));
                loadHW(hr[0], hb, wb);
                loadHW(hr[1], hb + HSTEP, wb + HSTEP);
                prep(op[0], fr[0], hr[0]);
                for (; g < n_groups; g += DEPTH * ST, vp += DEPTH * (int)(VSTEP * 8), hb += DEPTH * HSTEP, wb += DEPTH * HSTEP) {
#pragma unroll
                    for (int j = 0; j < DEPTH; ++j) {
#ifndef MXE_X_NO_VLOAD     // timing experiment only (results are wrong)
                        loadV(fr[(j + DEPTH - 1) % DEPTH], vp + (j + DEPTH - 1) * (int)(VSTEP * 8));
#endif
                        loadHW(hr[(j + 2) & 3], hb + (j + 2) * HSTEP, wb + (j + 2) * HSTEP);
                        // blocks, not a mix (measured, tools/mfma_shadow.hip: a vector instruction placed
                        // BETWEEN two MFMAs costs 14 cycles, behind the block 7): memory and vector
                        // instructions of the step first, then the 16 MFMAs back to back
                        prep(op[(j + 1) & 1], fr[(j + 1) % DEPTH], hr[(j + 1) & 3]);
                        __builtin_amdgcn_sched_barrier(0);
                        mma(op[j & 1], fr[j], hr[j & 3]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            MXE_STAMPW(1);
            // h: lane 16 i + 4 b + j holds column 16 t + 4 b + i of slot j; the waves are summed in step 4
            {
                const int hj = lane & 3, hcol = 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) hpart[(wave * MCC + hj) * NP + 16 * t + hcol] = hp[t];
            }
            MXE_STAMPW(6);
            // Gram tiles: every wave adds its partial tiles into the slots' W with LDS atomics
            // (ds_add_f64; W was zeroed at the start of the row pass), one barrier instead of four
            // rotating read-modify-write phases
            {
                typedef __attribute__((address_space(3))) double lds_double;
#pragma unroll
                for (int c = 0; c < MCC; ++c) {
                    double* Wq = Wm + (size_t)c * NA * LD;
                    int pr = 0;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                        for (int nt = mt; nt < NT; ++nt) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                // C/D layout of v_mfma_f32_16x16x4_f32: row 4 (l >> 4) + r, col l & 15
                                const int row = 16 * mt + 4 * kq + r, col = 16 * nt + cn;
                                __builtin_amdgcn_ds_atomic_fadd_f64((lds_double*)(Wq + row * LD + col), (double)acc[c][pr][r]);
                            }
                            ++pr;
                        }
                }
            }
            __syncthreads();
        }
        MXE_STAMPW(4);

        // ---- 4. home wave: rho, sums, accept / converge / advance, results ----
        if (wave < MCC) {
            const int q = wave, k = lane;
            double h = 0.0;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) h += hpart[(wv * MCC + q) * NP + k];
            const double r = (k < ns) ? cc[k] * h - gh[q * NP + k] : 0.0;
            const double r2 = wave_sum(r * r);
            double sS = 0.0, sdH = 0.0, sHn = 0.0, swm = 0.0, sdu = 0.0;
#pragma unroll
            for (int wv = 0; wv < NWV; ++wv) {
                sS += red[wv * 32 + q * 8 + 0]; sdH += red[wv * 32 + q * 8 + 1];
                sHn += red[wv * 32 + q * 8 + 2]; swm = fmax(swm, red[wv * 32 + q * 8 + 3]);
                sdu = fmax(sdu, red[wv * 32 + q * 8 + 4]);
            }
            Slot t;
            load_slot(t);
            if (t.active) {
                rho[q * NP + k] = r;
                const double chi2t = r2 + t.cperp, St = sS;
                const double Qt = 0.5 * chi2t - t.alpha * St;
                const bool finite = fabs(Qt) <= 1.7e308;
                bool finish_alpha = false; int conv = 0;
                if (t.scratch) {
                    // state restored from v (or first evaluation of the piece); damping kept
                    ++t.nevals;
                    if (finite) { t.scratch = 0; t.chi2 = chi2t; t.S = St; t.Hn2 = sHn; t.wmax = swm; t.Q = Qt; }
                    else finish_alpha = true;                   // cannot even evaluate: give up on this alpha
                } else if (!t.okprev) {
                    finish_alpha = true;                        // the damping loop ran out of range
                } else if (!finite || ((t.mu > 0.0 || (t.okprev >= 2 && t.okprev <= 4)) && Qt > t.Q + 1e-12 * fabs(t.Q)) ||   // (margin: rounding of Q)
                           // a predicted step may overshoot like any undamped Newton step (measured: a strict test
                           // rejects 20 % of them and costs more than the predictor gains); only a gross increase
                           // of Q -- an extrapolation gone wrong on a coarse alpha mesh -- rejects it
                           (t.okprev == 5 && Qt > 4.0 * fabs(t.Q) + 1.0) ||
                           // a full Newton step may overshoot (a cold start does, by factors of hundreds in Q, and
                           // recovers quadratically); one that multiplies Q by a million (alpha meshes with steps of
                           // a decade) does not come back
                           (t.okprev == 1 && Qt > 1e6 * (fabs(t.Q) + 1.0))) {
                    ++t.nevals;
                    if (finite && t.bt < 3 && !(t.mu == 0.0 && t.muh > 0.0)) {
                        // a shortened / damped / halved step that made Q worse: halve it (step 1)
                        ++t.bt;
                        t.okprev = 3;
                    } else {
                        // not finite, or still worse after three halvings: more damping, restore from v.  Where
                        // an earlier iteration of this piece needed damping, the search starts one notch below
                        // that level instead of climbing from mu_first again (alphas far below the physical range
                        // need it at every iteration)
                        t.mu = (t.mu == 0.0) ? fmax(p.mu_first * t.alpha, t.muh / p.mu_grow) : t.mu * p.mu_grow;
                        t.scratch = 1; t.bt = 0;
                        if (!(t.mu <= p.mu_max * t.alpha)) finish_alpha = true;
                    }
                } else {
                    // accepted
                    ++t.nevals;
                    // convergence in squares (no division, no square root in this serial section):
                    // relH^2 = sdH / Hn2 against tol_h^2.
                    // estimate of the NEXT Newton correction after a full step: the weights
                    // change by at most expm1(max|du|) relatively, and so does the Jacobian;
                    // the decoupled directions add the relative error theta of the Newton matrix.
                    // expm1 by its series up to x^4 for x <= 1 (relative error < 1e-2, an estimate), no
                    // estimate beyond
                    double fac2 = 1.0;
                    if (p.stop_estimate && t.mu == 0.0 && t.okprev == 1 && sdu <= 1.0) {
                        const double em1 = sdu * fma(sdu, fma(sdu, fma(sdu, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0);
                        const double fac = em1 + p.theta + MC_GRAM_ERR;
                        fac2 = fmin(1.0, fac * fac);
                    }
                    const double relH2_min = fac2 * sdH;            // min(relH, relH_next)^2 * Hn2
                    const double tol2Hn = p.tol_h * p.tol_h * t.Hn2;
                    vv[q * NP + k] -= dlc[q * NP + k];
                    if (t.niter == 0) t.capp = (t.okprev == 5) ? 2 : (t.okprev == 1 && t.mu == 0.0) ? 1 : 0;
                    else eacc[q * NP + k] -= dlc[q * NP + k];     // what the later iterations add to the first iterate
                    t.chi2 = chi2t; t.S = St; t.Hn2 = sHn; t.wmax = swm;
                    t.Qprev = t.Q; t.Q = Qt; t.muh = t.mu; t.mu = 0.0;
                    ++t.niter;
                    const bool newton_step = t.okprev != 4;       // a halved step says nothing about convergence
                    t.bt = 0;
                    if (newton_step && p.tol_h > 0.0 && relH2_min < tol2Hn && t.niter > p.miniter) { conv = 1; finish_alpha = true; }
                    else if (p.tol_relq > 0.0 && fabs(fabs(t.Qprev - t.Q) / t.Q) < p.tol_relq && t.niter > p.miniter) { conv = 1; finish_alpha = true; }
                    else if (t.niter >= p.maxiter) finish_alpha = true;
                }
                if (finish_alpha) {
                    const size_t prob = (size_t)t.prob0 + t.ia;
                    if (p.out_H) {
                        // H of the point just evaluated (the accepted one unless the alpha failed)
                        double* Ho = p.out_H + prob * nw;
                        for (int i = lane; i < nw; i += 64) Ho[i] = Hi[i * MCC + q];
                    }
                    if (p.out_v) p.out_v[prob * NP + lane] = vv[q * NP + lane];
                    if (lane == 0) {
                        p.out_chi2[prob] = t.chi2; p.out_S[prob] = t.S; p.out_Q[prob] = t.Q;
                        p.out_niter[prob] = t.niter; p.out_conv[prob] = conv;
                        p.out_nevals[prob] = t.nevals; p.out_nact[prob] = t.nact;
                    }
                    {
                        // defect of this alpha's first Newton iterate -> predictor of the next alpha, scaled
                        // with the square of the ratio of the steps in log alpha
                        double e = 0.0;
                        if (conv && t.capp > 0 && t.ia > 0 && t.ia + 1 < t.clen) {
                            const double a0 = alpha_at(t, t.ia - 1), a1 = t.alpha, a2 = alpha_at(t, t.ia + 1);
                            const double q0 = a1 / a0, q1 = a2 / a1;       // a logarithmic mesh: equal ratios, no log
                            const double rr = (fabs(q1 - q0) < 1e-9 * q0) ? 1.0 : log(q1) / log(q0);
                            // the extrapolation is an expansion in the step h of log alpha: fine meshes only
                            // (|h| <= MC_PRED_HMAX, i.e. alpha ratios between 0.74 and 1.35)
                            const bool fine = q0 > 0.74 && q0 < 1.35 && q1 > 0.74 && q1 < 1.35;
                            if (fine)
                            e = ((t.capp == 2 ? ecor[q * NP + k] : 0.0) + eacc[q * NP + k]) * rr * rr;
                            if (!(fabs(e) < 1e300)) e = 0.0;
                        }
                        ecor[q * NP + k] = e; eacc[q * NP + k] = 0.0;
                    }
                    ++t.ia;
                    t.niter = 0; t.nevals = 0; t.mu = 0.0; t.bt = 0; t.capp = 0;
                    t.Qprev = __builtin_nan("");
                    if (t.ia >= t.clen) {
                        t.active = 0;
                        if (dynamic) {           // next piece from the queue (most expensive first)
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
                store_slot(t);
            }
        }
        __syncthreads();                         // slot flags, v, rho visible to the next round
        MXE_STAMPW(5);
#ifdef MXE_PROFILE
        ++prof_rounds;
#endif
    }
#ifdef MXE_PROFILE
    if (lane == 0 && p.prof && blockIdx.x < 1024) {
        long long* pr = p.prof + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int r = 0; r < 7; ++r) pr[r] = prof_acc[r];
        pr[7] = (wave == 0) ? prof_rounds : prof_acc[7];     // wave 0: rounds; others: wait at the row-pass barrier
    }
#endif
}

} // namespace mxe
