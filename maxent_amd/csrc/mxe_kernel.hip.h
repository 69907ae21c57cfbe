// mxe_kernel.hip.h -- the alpha-chain solver kernel (gfx950, wave64, fp64)
//
// One workgroup of NW wavefronts solves one chain = the warm-started alpha
// scan of one matrix element (reference: maxent_loop.py:241-245 around
// levenberg_minimizer.py:123-248).  All state of the chain lives in LDS and
// registers; HBM/L2 traffic is the shared singular basis V (streamed once per
// mat-vec / Gram pass, L2 resident), the element's D and ghat, and the
// per-alpha results.
//
// Mathematics (whitened singular basis, see DESIGN.md and oracle/sform.py):
//   u = V v,  H = D e^u | D(e^u - e^-u),  w = H | D(e^u + e^-u)
//   h = V^T H,  rho = c*h - ghat,  chi2 = |rho|^2 + c_perp
//   S = sum(H - D - H u) | sum(H+ - D - H+ u) + sum(H- - D + H- u)
//   g = c*rho + alpha v,  W = V^T diag(w) V
//   Newton step (Bryan):  (c W c + (alpha+mu) I) z = rho + alpha v / c,
//   delta = c*z, accepted when delta^T W delta <= step_max * sum(D) and the
//   trial point is finite; converged when |w * V delta| / |H| < tol_h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mxe {

constexpr int WAVE = 64;
constexpr int GRAM_R = 4;        // rows staged per wave per Gram tile

struct KParams {
    int n_omega;        // number of frequencies
    int n_omega_pad;    // multiple of 64
    int n_s;            // kept singular values (<= NP)
    int NP;             // padded n_s (row stride of V): 64
    int n_alpha;
    int n_chain;
    // basis arrays, indexed by data-set id
    const double* V;     // [n_ds][n_omega][NP]      zero padded columns
    const double* Vt;    // [n_ds][NP][n_omega_pad]  zero padded
    const double* c;     // [n_ds][NP]               padded with 1
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
    const double* alpha;    // [n_chain][n_alpha]
    const double* v0;       // [n_chain][NP]   whitened basis
    // outputs, problem p = chain*n_alpha + i
    double* out_v;      // [P][NP]  whitened basis
    double* out_H;      // [P][n_omega]
    double* out_chi2;   // [P]
    double* out_S;
    double* out_Q;
    int* out_niter;
    int* out_conv;
    int* out_nevals;
    // options
    int maxiter, miniter;
    double tol_h, tol_d, tol_relq, step_max, mu_first, mu_grow, mu_max;
};

__device__ __forceinline__ void wave_sync() {
    // orders LDS traffic between the lanes of one wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}

__device__ __forceinline__ double wave_bcast(double x, int lane) {
    return __shfl(x, lane, WAVE);
}

// block-wide sum of NV values per thread; result valid in every thread.
template <int NW, int NV>
__device__ __forceinline__ void block_sum(double (&x)[NV], double* red /*[NW*NV]*/) {
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = wave_sum(x[q]);
    if (NW == 1) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[wave * NV + q] = x[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) s += red[wv * NV + q];
        x[q] = s;
    }
}

template <int NW> __device__ __forceinline__ void block_sync() {
    if (NW == 1) wave_sync(); else __syncthreads();
}

// position of singular index k inside a staged Gram row: blocks of BS padded
// to an odd stride so that the 8 block bases fall on distinct LDS banks.
template <int BS> struct GramLayout {
    static constexpr int BSP = (BS % 2) ? BS : BS + 1;
    static constexpr int ROW = 8 * BSP;
    __device__ static __forceinline__ int pos(int k) { return (k / BS) * BSP + (k % BS); }
};

template <int NW, int BS>
__global__ __launch_bounds__(64 * NW)
void chain_kernel(const KParams p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = WAVE * NW;
    using GL = GramLayout<BS>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.x;
    if (chain >= p.n_chain) return;

    const int NP = p.NP, LD = NP + 1, ns = p.n_s;
    const int nw = p.n_omega, nwp = p.n_omega_pad;

    // ---- LDS carve ----
    double* Wm   = lds;                    // [NP][LD]  upper+diag: W ; strict lower: L
    double* dinv = Wm + NP * LD;           // [NP] 1/L_jj
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
    double* Wd   = rhot + NP;
    double* hpart = Wd + NP;               // [NW][NP]
    double* red  = hpart + NW * NP;        // [NW*8]
    double* u    = red + NW * 8;           // [nwp]
    double* ut   = u + nwp;
    double* w    = ut + nwp;
    double* wt   = w + nwp;
    double* Hs   = wt + nwp;
    double* stage = Hs + nwp;              // [NW][2][GRAM_R][GL::ROW]

    const int elem = p.chain_elem[chain];
    const int ds = p.elem_ds[elem];
    const int kind = p.elem_kind[elem];
    const double* __restrict__ V  = p.V  + (size_t)ds * nw * NP;
    const double* __restrict__ Vt = p.Vt + (size_t)ds * NP * nwp;
    const double* __restrict__ Dg = p.D + (size_t)elem * nwp;
    const double cperp = p.cperp[elem];
    const double step_lim = p.step_max * p.sumD[elem];

    if (tid < NP) {
        cc[tid] = p.c[ds * NP + tid];
        ci[tid] = p.cinv[ds * NP + tid];
        gh[tid] = p.ghat[(size_t)elem * NP + tid];
        v[tid]  = p.v0[(size_t)chain * NP + tid];
        dl[tid] = 0.0;
    }
    block_sync<NW>();

    // ------------------------------------------------------------------
    // evaluation pass.  trial u = u_base - V*dl (dl == 0, u_base == nullptr:
    // u = V v from scratch).  Fills ut, wt, Hs, rhot; returns chi2, S,
    // |w o V dl|^2 (with the OLD w) and |H_trial|^2.
    // ------------------------------------------------------------------
    auto eval_pass = [&](const double* vec, bool from_scratch,
                         double& chi2, double& S, double& dH2, double& Hn2) {
        double part[3] = {0.0, 0.0, 0.0};   // S, dH2, Hn2
        for (int i = tid; i < nwp; i += T) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const double* col = Vt + i;
            int k = 0;
            for (; k + 3 < ns; k += 4) {
                a0 = fma(col[(size_t)(k + 0) * nwp], vec[k + 0], a0);
                a1 = fma(col[(size_t)(k + 1) * nwp], vec[k + 1], a1);
                a2 = fma(col[(size_t)(k + 2) * nwp], vec[k + 2], a2);
                a3 = fma(col[(size_t)(k + 3) * nwp], vec[k + 3], a3);
            }
            for (; k < ns; ++k) a0 = fma(col[(size_t)k * nwp], vec[k], a0);
            const double vd = (a0 + a1) + (a2 + a3);
            double ui;
            if (from_scratch) ui = vd;
            else {
                ui = u[i] - vd;
                const double t = w[i] * vd;
                part[1] = fma(t, t, part[1]);
            }
            const double Di = Dg[i];
            double Hi, wi, Si;
            if (kind == 0) {
                const double e = exp(ui);
                Hi = Di * e; wi = Hi;
                Si = Hi - Di - Hi * ui;
            } else {
                const double ep = exp(ui), em = exp(-ui);
                const double Hp = Di * ep, Hm = Di * em;
                Hi = Hp - Hm; wi = Hp + Hm;
                Si = (Hp - Di - Hp * ui) + (Hm - Di + Hm * ui);
            }
            if (i >= nw) { Hi = 0.0; wi = 0.0; Si = 0.0; }
            ut[i] = ui; wt[i] = wi; Hs[i] = Hi;
            part[0] += Si;
            part[2] = fma(Hi, Hi, part[2]);
        }
        block_sync<NW>();                    // Hs complete
        // h = V^T H : lane = singular index, rows split over the waves
        {
            const int rows_per = (nw + NW - 1) / NW;
            const int r0 = wave * rows_per;
            const int r1 = min(nw, r0 + rows_per);
            double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
            const double* Vc = V + lane;
            int i = r0;
            for (; i + 3 < r1; i += 4) {
                b0 = fma(Vc[(size_t)(i + 0) * NP], Hs[i + 0], b0);
                b1 = fma(Vc[(size_t)(i + 1) * NP], Hs[i + 1], b1);
                b2 = fma(Vc[(size_t)(i + 2) * NP], Hs[i + 2], b2);
                b3 = fma(Vc[(size_t)(i + 3) * NP], Hs[i + 3], b3);
            }
            for (; i < r1; ++i) b0 = fma(Vc[(size_t)i * NP], Hs[i], b0);
            hpart[wave * NP + lane] = (b0 + b1) + (b2 + b3);
        }
        block_sync<NW>();
        double r2 = 0.0;
        if (tid < NP) {
            double h = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) h += hpart[wv * NP + tid];
            const double r = (tid < ns) ? cc[tid] * h - gh[tid] : 0.0;
            rhot[tid] = r;
            r2 = r * r;
        }
        double x4[4] = {part[0], part[1], part[2], r2};
        block_sum<NW, 4>(x4, red);
        S = x4[0]; dH2 = x4[1]; Hn2 = x4[2]; chi2 = x4[3] + cperp;
    };

    auto accept_trial = [&]() {
        for (int i = tid; i < nwp; i += T) { u[i] = ut[i]; w[i] = wt[i]; }
        if (tid < NP) { v[tid] -= dl[tid]; rho[tid] = rhot[tid]; }
        block_sync<NW>();
    };

    // ------------------------------------------------------------------
    // Gram matrix W = V^T diag(w) V -> Wm (upper triangle + diagonal)
    // lane (br, bc) of every wave owns the BS x BS block (br, bc); the waves
    // split the omega rows; partial blocks are summed through LDS.
    // ------------------------------------------------------------------
    auto gram = [&]() {
        const int br = lane >> 3, bc = lane & 7;
        double acc[BS][BS];
#pragma unroll
        for (int j = 0; j < BS; ++j)
#pragma unroll
            for (int k = 0; k < BS; ++k) acc[j][k] = 0.0;
        double* xs = stage + (size_t)wave * 2 * GRAM_R * GL::ROW;
        double* ys = xs + GRAM_R * GL::ROW;
        const int n_tiles = (nw + GRAM_R - 1) / GRAM_R;
        const int mypos = GL::pos(lane);
        const bool stager = lane < 8 * BS;
        for (int t = wave; t < n_tiles; t += NW) {
            const int i0 = t * GRAM_R;
            double val[GRAM_R], wv_[GRAM_R];
#pragma unroll
            for (int r = 0; r < GRAM_R; ++r) {
                const int i = i0 + r;
                const bool ok = (i < nw);
                val[r] = ok ? V[(size_t)i * NP + lane] : 0.0;
                wv_[r] = ok ? w[i] : 0.0;
            }
            wave_sync();                     // previous tile fully consumed
            if (stager) {
#pragma unroll
                for (int r = 0; r < GRAM_R; ++r) {
                    ys[r * GL::ROW + mypos] = val[r];
                    xs[r * GL::ROW + mypos] = val[r] * wv_[r];
                }
            }
            wave_sync();
#pragma unroll
            for (int r = 0; r < GRAM_R; ++r) {
                double x[BS], y[BS];
#pragma unroll
                for (int j = 0; j < BS; ++j) x[j] = xs[r * GL::ROW + br * GL::BSP + j];
#pragma unroll
                for (int k = 0; k < BS; ++k) y[k] = ys[r * GL::ROW + bc * GL::BSP + k];
#pragma unroll
                for (int j = 0; j < BS; ++j)
#pragma unroll
                    for (int k = 0; k < BS; ++k) acc[j][k] = fma(x[j], y[k], acc[j][k]);
            }
        }
        // reduce over the waves into Wm (upper + diagonal)
        for (int wv = 0; wv < NW; ++wv) {
            if (wave == wv && br <= bc) {
#pragma unroll
                for (int j = 0; j < BS; ++j)
#pragma unroll
                    for (int k = 0; k < BS; ++k) {
                        const int row = br * BS + j, col = bc * BS + k;
                        if (row <= col && col < NP) {
                            if (wv == 0) Wm[row * LD + col] = acc[j][k];
                            else Wm[row * LD + col] += acc[j][k];
                        }
                    }
            }
            block_sync<NW>();
        }
    };

    // symmetric mat-vec  out = W x  (W in upper+diag of Wm), threads < NP
    auto symv = [&](const double* x, double* out) {
        if (tid < NP) {
            double s0 = 0.0, s1 = 0.0;
            const int i = tid;
            int j = 0;
            for (; j < i && j < ns; ++j) s0 = fma(Wm[j * LD + i], x[j], s0);   // column i
            for (j = i; j < ns; ++j) s1 = fma(Wm[i * LD + j], x[j], s1);       // row i
            out[i] = (i < ns) ? s0 + s1 : 0.0;
        }
    };

    // ------------------------------------------------------------------
    // wave 0: Cholesky of A = c W c + a I (left-looking, lane = row), with
    // the right-hand side carried along as an extra row, then the back
    // substitution.  L goes to the strict lower triangle of Wm, 1/L_jj to
    // dinv.  Result z in zz[].  Returns false on a non-positive pivot.
    // ------------------------------------------------------------------
    auto chol_solve = [&](double a) -> bool {
        bool ok = true;
        if (wave == 0) {
            const int i = lane;
            const double ci_ = (i < ns) ? cc[i] : 0.0;
            double yacc = 0.0;                 // forward substitution of rhs, lane j holds y_j
            for (int j = 0; j < ns; ++j) {
                double s;
                if (i > j)       s = ci_ * Wm[j * LD + i] * cc[j];
                else if (i == j) s = fma(ci_ * Wm[j * LD + j], ci_, a);
                else             s = 0.0;
                double r = rhs[j];             // rhs row (uniform)
                double s0 = 0.0, s1 = 0.0, r0 = 0.0;
                if (i >= j && i < ns) {
                    int k = 0;
                    for (; k + 1 < j; k += 2) {
                        s0 = fma(Wm[i * LD + k],     Wm[j * LD + k],     s0);
                        s1 = fma(Wm[i * LD + k + 1], Wm[j * LD + k + 1], s1);
                    }
                    if (k < j) s0 = fma(Wm[i * LD + k], Wm[j * LD + k], s0);
                }
                // rhs row: sum_k y_k L_jk, y_k lives in lane k -> wave reduction
                {
                    const double t = (i < j) ? yacc * Wm[j * LD + i] : 0.0;
                    r0 = wave_sum(t);
                }
                s -= (s0 + s1);
                const double piv = wave_bcast(s, j);
                if (!(piv > 0.0)) { ok = false; break; }
                const double inv = 1.0 / sqrt(piv);
                if (i == j) { dinv[j] = inv; yacc = (r - r0) * inv; }
                if (i > j && i < ns) Wm[i * LD + j] = s * inv;
                wave_sync();
            }
            if (ok) {
                // back substitution L^T z = y ; lane i holds residual r_i
                double ri = (i < ns) ? yacc : 0.0;
                for (int j = ns - 1; j >= 0; --j) {
                    const double zj = wave_bcast(ri, j) * dinv[j];
                    if (i == j) zz[j] = zj;
                    if (i < j) ri = fma(-Wm[j * LD + i], zj, ri);
                }
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
    // initial state: u = V v0
    // ------------------------------------------------------------------
    double chi2, S, dH2, Hn2;
    eval_pass(v, true, chi2, S, dH2, Hn2);
    accept_trial();                 // dl == 0: v unchanged
    int nevals_pending = 1;

    for (int ia = 0; ia < p.n_alpha; ++ia) {
        const double alpha = p.alpha[(size_t)chain * p.n_alpha + ia];
        int n_iter = 0, conv = 0, nevals = nevals_pending;
        nevals_pending = 0;
        double Qprev = __builtin_nan("");
        double Q = 0.5 * chi2 - alpha * S;
        bool failed = false;

        for (int it = 0; it < p.maxiter && !failed; ++it) {
            if (tid < NP) {
                const double vv = v[tid], r = rho[tid];
                g[tid]   = (tid < ns) ? fma(cc[tid], r, alpha * vv) : 0.0;
                rhs[tid] = (tid < ns) ? fma(alpha * vv, ci[tid], r) : 0.0;
            }
            gram();                 // ends with a block sync
            // reference-style criteria (convergence_methods.py:81-122)
            bool stop = false;
            if (p.tol_d > 0.0) {
                symv(g, Wd);
                double m[1] = {0.0};
                block_sync<NW>();
                if (tid < ns) m[0] = fabs(Wd[tid]);
                // max over the block via sum of a 0/1 test is not enough: do a max reduce
                double mx = m[0];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, WAVE));
                if (NW > 1) {
                    __syncthreads();
                    if (lane == 0) red[wave] = mx;
                    __syncthreads();
                    mx = red[0];        // only wave 0 holds tid < NP
                    __syncthreads();
                }
                if (mx < p.tol_d) stop = true;
            }
            if (p.tol_relq > 0.0 && it > 0) {
                if (fabs(fabs(Qprev - Q) / Q) < p.tol_relq) stop = true;
            }
            if (stop && it >= p.miniter) { conv = 1; break; }

            // ---- damped Newton step with Bryan's step bound ----
            double mu = 0.0;
            double chi2t = 0.0, St = 0.0, dH2t = 0.0, Hn2t = 0.0;
            bool accepted = false;
            while (true) {
                const bool okc = chol_solve(alpha + mu);
                bool good = okc;
                if (okc) {
                    if (tid < NP) dl[tid] = (tid < ns) ? cc[tid] * zz[tid] : 0.0;
                    block_sync<NW>();
                    symv(dl, Wd);
                    block_sync<NW>();
                    double nrm = 0.0;
                    if (tid < NP) nrm = dl[tid] * Wd[tid];
                    double x1[1] = {nrm};
                    block_sum<NW, 1>(x1, red);
                    if (!(x1[0] <= step_lim)) good = false;
                    if (good) {
                        eval_pass(dl, false, chi2t, St, dH2t, Hn2t);
                        ++nevals;
                        const double Qt = 0.5 * chi2t - alpha * St;
                        if (!(fabs(Qt) <= 1.7e308)) good = false;   // NaN / inf
                    }
                }
                if (good) { accepted = true; break; }
                mu = (mu == 0.0) ? p.mu_first * alpha : mu * p.mu_grow;
                if (!(mu <= p.mu_max * alpha)) break;
            }
            if (!accepted) { failed = true; break; }
            const double relH = sqrt(dH2t / Hn2);
            accept_trial();
            chi2 = chi2t; S = St; Hn2 = Hn2t;
            Qprev = Q;
            Q = 0.5 * chi2 - alpha * S;
            ++n_iter;
            if (p.tol_h > 0.0 && relH < p.tol_h && n_iter > p.miniter) { conv = 1; break; }
        }

        // ---- results of this alpha (MaxEntResult fields, maxent_result.py:835-967)
        const size_t prob = (size_t)chain * p.n_alpha + ia;
        if (p.out_H) {
            double* Ho = p.out_H + prob * nw;
            for (int i = tid; i < nw; i += T) {
                const double Di = Dg[i], ui = u[i];
                Ho[i] = (kind == 0) ? Di * exp(ui) : Di * exp(ui) - Di * exp(-ui);
            }
        }
        if (p.out_v && tid < NP) p.out_v[prob * NP + tid] = v[tid];
        if (tid == 0) {
            p.out_chi2[prob] = chi2;
            p.out_S[prob] = S;
            p.out_Q[prob] = Q;
            p.out_niter[prob] = n_iter;
            p.out_conv[prob] = conv;
            p.out_nevals[prob] = nevals;
        }
    }
}

} // namespace mxe
