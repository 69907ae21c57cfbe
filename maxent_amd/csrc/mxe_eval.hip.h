// mxe_eval.hip.h -- the cost function and its derivatives at caller-supplied points
//
// One workgroup (4 wavefronts) per problem, everything binary64.  This is the device side of
//     CostFunction.__call__ / f / d / dd          cost_functions/cost_function.py:73-85
//     MaxEntCostFunction.f / dH / d / ddH / dd    cost_functions/maxent_cost_function.py:68-165
//     BryanCostFunction.f / d / dd                cost_functions/bryan_cost_function.py:57-128
//     NormalChi2, NormalEntropy, PlusMinusEntropy, NormalH_of_v, PlusMinusH_of_v   functions.py:336-796
// in the singular-space form of SURVEY.md section 8 (whitened basis):
//     u = V v,  H = D e^u | D (e^u - e^-u),  w = H | D (e^u + e^-u),  h = V^T H,
//     rho = c h - ghat,  chi2 = |rho|^2 + c_perp,  S,  Q = eta chi2 / 2 - alpha S,
//     g = eta c rho + alpha v            (= dQ/dv "without the factor W": Bryan's d, d with dA_projection = 1)
//     W  = V^T diag(w) V                 (every derivative of the reference is built from g, W and M = diag(c^2):
//                                          d = W g, dd = W M W + alpha W (default); d = g, dd = M W (Bryan))
//     W2 = V^T diag((V g) H) V           (the d^2 H / dv dv term of d_dv = True)
// It is also the checker of the solver itself (MODE_AUDIT): at the v the chain kernel returned, the
// exact Newton correction  (eta c W c + alpha I) z = eta rho + alpha v / c,  delta = c z, over ALL n_s
// directions, and its size in the metric of the stopping rule,  ||w * V delta||_2 / ||H||_2  -- to first
// order the distance of the returned H from the minimiser.
#pragma once
#include "mxe_kernel.hip.h"

namespace mxe {

struct EvalParams {
    int nw, nwp, ns, NP;
    const double* V;        // [n_ds][nwp][NP]
    const double* Vt;       // [n_ds][NP][nwp]
    const double* c;        // [n_ds][NP]
    const int* elem_ds;
    const int* elem_kind;
    const double* ghat;     // [n_elem][NP]
    const double* cperp;    // [n_elem]
    const double* D;        // [n_elem][nwp]
    const int* elem;        // [P / elem_div] element of the problem
    int elem_div;           // problems per entry of elem[] (n_alpha for the audit of a launch, else 1)
    const double* alpha;    // [P]
    const double* x;        // input: v (whitened basis) [P][x_stride] or H [P][x_stride]
    int x_stride;
    int input_is_H;         // 1: x is the hidden image H itself (u = H_of_v.inv(H)); g then has no alpha v term
    double eta;             // chi2_factor
    int want_gram, want_gram2, want_audit;
    // outputs (device; any may be null)
    double* Q; double* chi2; double* S;         // [P]
    double* H; double* u; double* w; double* q; // [P][nw]   q = V g = dQ/dH
    double* h; double* g;                       // [P][NP]  whitened basis
    double* W; double* W2;                      // [P][NP][NP] whitened basis, full symmetric
    double* corr; double* gmax;                 // [P] audit: ||w * V delta|| / ||H||, max |g_k| / (|eta c rho|_k + |alpha v|_k)
};

// block-wide sum of NV values through LDS (256 threads); result in every thread
template <int NV>
__device__ __forceinline__ void eval_block_sum(double (&x)[NV], double* red /*[4 * NV]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = wave_sum(x[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[wave * NV + q] = x[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = (red[q] + red[NV + q]) + (red[2 * NV + q] + red[3 * NV + q]);
}

// W = V^T diag(wsh) V into Bm (full, symmetric) with v_mfma_f64_16x16x4_f64: the omega rows split over
// the four waves, the upper-triangular tiles of one tile row per sweep of V (as in logdet_kernel)
template <int NT>
__device__ __forceinline__ void eval_gram(const double* __restrict__ V, const double* wsh, double* Bm,
                                          int nwp, int ns)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int NP = 16 * NT, LD = NP + 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, cn = lane & 15;
    const int n_groups = nwp >> 2;
    const int ntile = (ns + 15) >> 4;
    for (int i = tid; i < NP * LD; i += 256) Bm[i] = 0.0;
    __syncthreads();
    for (int mt = 0; mt < ntile; ++mt) {
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        for (int g = wave; g < n_groups; g += 4) {
            const double* row = V + (size_t)(4 * g + kq) * NP + cn;
            const double am = row[16 * mt] * wsh[4 * g + kq];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t >= mt && t < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, row[16 * t], acc[t], 0, 0, 0);
        }
        for (int ph = 0; ph < 4; ++ph) {         // fixed order: the result does not depend on timing
            if (wave == ph) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (t >= mt && t < ntile) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Bm[(16 * mt + kq + 4 * r) * LD + 16 * t + cn] += acc[t][r];
                    }
            }
            __syncthreads();
        }
    }
    // mirror: the tiles hold rows <= columns (tile-wise); make the matrix full
    for (int idx = tid; idx < NP * NP; idx += 256) {
        const int i = idx / NP, j = idx % NP;
        if ((i >> 4) > (j >> 4)) Bm[i * LD + j] = Bm[j * LD + i];
    }
    __syncthreads();
}

template <int NT>
__global__ __launch_bounds__(256)
void eval_kernel(const EvalParams ep)
{
    constexpr int NP = 16 * NT, LD = NP + 1;
    extern __shared__ double sm[];
    double* Bm  = sm;                    // [NP][LD]
    double* wsh = Bm + NP * LD;          // [nwp]  w (then the weights of W2)
    double* Hsh = wsh + ep.nwp;          // [nwp]
    double* vsh = Hsh + ep.nwp;          // [NP]   v
    double* gsh = vsh + NP;              // [NP]   g
    double* zsh = gsh + NP;              // [NP]   rhs / z / delta
    double* hp  = zsh + NP;              // [4][NP] partial h
    double* red = hp + 4 * NP;           // [16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t prob = blockIdx.x;
    const int nw = ep.nw, nwp = ep.nwp, ns = ep.ns;
    const int e = ep.elem[prob / ep.elem_div];
    const int ds = ep.elem_ds[e], kind = ep.elem_kind[e];
    const double a = ep.alpha[prob], eta = ep.eta;
    const double* V  = ep.V + (size_t)ds * nwp * NP;
    const double* Vt = ep.Vt + (size_t)ds * NP * nwp;
    const double* cc = ep.c + (size_t)ds * NP;
    const double* Dp = ep.D + (size_t)e * nwp;
    const double* xp = ep.x + prob * ep.x_stride;

    for (int k = tid; k < NP; k += 256) vsh[k] = (!ep.input_is_H && k < ns) ? xp[k] : 0.0;
    __syncthreads();
    // ---- u, H, w, S ----
    double sums[2] = {0.0, 0.0};         // S, sum H^2
    for (int i = tid; i < nwp; i += 256) {
        double Hq = 0.0, wq = 0.0, uq = 0.0;
        if (i < nw) {
            const double Di = Dp[i];
            if (ep.input_is_H) {
                // H_of_v.inv (functions.py:749-755, 790-796) with its safelog (functions.py:53-56)
                Hq = xp[i];
                if (kind == 0) { wq = Hq; const double r = Hq / Di; uq = log(fabs(r) > 1e-100 ? r : 1e-100); }
                else {
                    const double d2 = 2.0 * Di; wq = sqrt(fma(Hq, Hq, d2 * d2));
                    const double r = (Hq + wq) / d2; uq = log(fabs(r) > 1e-100 ? r : 1e-100);
                }
                if (kind == 0) sums[0] += Hq - Di - Hq * uq;
                else {
                    const double Hp = 0.5 * (wq + Hq), Hm = 0.5 * (wq - Hq);
                    sums[0] += (Hp - Di - Hp * uq) + (Hm - Di + Hm * uq);
                }
            } else {
                double acc0 = 0.0, acc1 = 0.0;
                int k = 0;
                for (; k + 1 < ns; k += 2) {
                    acc0 = fma(Vt[(size_t)k * nwp + i], vsh[k], acc0);
                    acc1 = fma(Vt[(size_t)(k + 1) * nwp + i], vsh[k + 1], acc1);
                }
                if (k < ns) acc0 = fma(Vt[(size_t)k * nwp + i], vsh[k], acc0);
                uq = acc0 + acc1;
                const double ep_ = exp(uq);
                const double Hp = Di * ep_;
                if (kind == 0) { Hq = Hp; wq = Hp; sums[0] += Hp - Di - Hp * uq; }
                else {
                    const double Hm = Di * exp(-uq);
                    Hq = Hp - Hm; wq = Hp + Hm;
                    sums[0] += (Hp - Di - Hp * uq) + (Hm - Di + Hm * uq);
                }
            }
            sums[1] = fma(Hq, Hq, sums[1]);
            if (ep.H) ep.H[prob * nw + i] = Hq;
            if (ep.u) ep.u[prob * nw + i] = uq;
            if (ep.w) ep.w[prob * nw + i] = wq;
        }
        Hsh[i] = Hq; wsh[i] = wq;
    }
    eval_block_sum<2>(sums, red);
    const double Sval = sums[0], Hn2 = sums[1];
    // ---- h = V^T H (row-major V: lane = column) ----
    for (int k0 = 0; k0 < NP; k0 += 64) {
        const int k = k0 + lane;
        double acc = 0.0;
        for (int i = wave; i < nw; i += 4) acc = fma(V[(size_t)i * NP + k], Hsh[i], acc);
        hp[wave * NP + k] = acc;
    }
    __syncthreads();
    double part[1] = {0.0};              // |rho|^2
    double gm = 0.0;
    for (int k = tid; k < NP; k += 256) {
        double hk = 0.0, rk = 0.0, gk = 0.0;
        if (k < ns) {
            hk = (hp[k] + hp[NP + k]) + (hp[2 * NP + k] + hp[3 * NP + k]);
            rk = cc[k] * hk - ep.ghat[(size_t)e * NP + k];
            const double t1 = eta * cc[k] * rk, t2 = a * vsh[k];
            gk = t1 + t2;
            const double sc = fabs(t1) + fabs(t2);
            gm = (sc > 0.0) ? fabs(gk) / sc : 0.0;
            part[0] = rk * rk;
            // right-hand side of the Newton system in the z variable (delta = c z)
            zsh[k] = eta * rk + a * vsh[k] / cc[k];
        } else {
            zsh[k] = 0.0;
        }
        gsh[k] = gk;
        if (ep.h) ep.h[prob * NP + k] = hk;
        if (ep.g) ep.g[prob * NP + k] = gk;
    }
    eval_block_sum<1>(part, red);
    gm = wave_max(gm);
    __syncthreads();
    if (lane == 0) red[wave] = gm;
    __syncthreads();
    gm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    const double chi2 = part[0] + ep.cperp[e];
    if (tid == 0) {
        if (ep.chi2) ep.chi2[prob] = chi2;
        if (ep.S) ep.S[prob] = Sval;
        if (ep.Q) ep.Q[prob] = 0.5 * eta * chi2 - a * Sval;
        if (ep.gmax) ep.gmax[prob] = gm;
    }
    if (ep.q) {
        for (int i = tid; i < nw; i += 256) {
            double q = 0.0;
            for (int k = 0; k < ns; ++k) q = fma(Vt[(size_t)k * nwp + i], gsh[k], q);
            ep.q[prob * nw + i] = q;
        }
    }
    if (!(ep.want_gram || ep.want_gram2 || ep.want_audit)) return;
    // ---- W ----
    if (ep.want_gram || ep.want_audit) {
        eval_gram<NT>(V, wsh, Bm, nwp, ns);
        if (ep.W)
            for (int idx = tid; idx < NP * NP; idx += 256) {
                const int i = idx / NP, j = idx % NP;
                ep.W[prob * NP * NP + idx] = (i < ns && j < ns) ? Bm[i * LD + j] : 0.0;
            }
    }
    // ---- audit: exact Newton correction over all n_s directions ----
    if (ep.want_audit) {
        __syncthreads();
        for (int idx = tid; idx < ns * ns; idx += 256) {
            const int i = idx / ns, j = idx % ns;
            if (i >= j) {
                double b = eta * cc[i] * Bm[i * LD + j] * cc[j];
                if (i == j) b += a;
                Bm[i * LD + j] = b;          // lower triangle
            }
        }
        __syncthreads();
        bool ok = true;
        for (int j = 0; j < ns; ++j) {           // right-looking Cholesky
            const double piv = Bm[j * LD + j];
            if (!(piv > 0.0)) ok = false;
            const double d = sqrt(piv);
            __syncthreads();
            for (int i = j + tid; i < ns; i += 256) Bm[i * LD + j] /= d;
            __syncthreads();
            const int m = ns - j - 1;
            for (int idx = tid; idx < m * m; idx += 256) {
                const int i = j + 1 + idx / m, k = j + 1 + idx % m;
                if (k <= i) Bm[i * LD + k] = fma(-Bm[i * LD + j], Bm[k * LD + j], Bm[i * LD + k]);
            }
            __syncthreads();
        }
        if (wave == 0) {                         // two triangular solves, column oriented, one wave
            for (int j = 0; j < ns; ++j) {
                if (lane == (j & 63)) zsh[j] /= Bm[j * LD + j];
                wave_sync();
                const double zj = zsh[j];
                for (int i = lane; i < ns; i += 64) if (i > j) zsh[i] = fma(-Bm[i * LD + j], zj, zsh[i]);
                wave_sync();
            }
            for (int j = ns - 1; j >= 0; --j) {
                if (lane == (j & 63)) zsh[j] /= Bm[j * LD + j];
                wave_sync();
                const double zj = zsh[j];
                for (int i = lane; i < j; i += 64) zsh[i] = fma(-Bm[j * LD + i], zj, zsh[i]);
                wave_sync();
            }
            for (int i = lane; i < ns; i += 64) zsh[i] *= cc[i];       // delta = c z
        }
        __syncthreads();
        double s2[1] = {0.0};
        for (int i = tid; i < nw; i += 256) {
            double du = 0.0;
            for (int k = 0; k < ns; ++k) du = fma(Vt[(size_t)k * nwp + i], zsh[k], du);
            const double t = wsh[i] * du;
            s2[0] = fma(t, t, s2[0]);
        }
        eval_block_sum<1>(s2, red);
        if (tid == 0 && ep.corr) ep.corr[prob] = ok ? sqrt(s2[0] / Hn2) : __builtin_nan("");
    }
    // ---- W2 = V^T diag((V g) H'') V,  H'' = d^2 H / du^2 = H for both entropies ----
    if (ep.want_gram2 && ep.W2) {
        __syncthreads();
        for (int i = tid; i < nwp; i += 256) {
            double q = 0.0;
            if (i < nw) for (int k = 0; k < ns; ++k) q = fma(Vt[(size_t)k * nwp + i], gsh[k], q);
            wsh[i] = q * Hsh[i];
        }
        __syncthreads();
        eval_gram<NT>(V, wsh, Bm, nwp, ns);
        for (int idx = tid; idx < NP * NP; idx += 256) {
            const int i = idx / NP, j = idx % NP;
            ep.W2[prob * NP * NP + idx] = (i < ns && j < ns) ? Bm[i * LD + j] : 0.0;
        }
    }
}

// ---- entropy of a hidden image given directly (NormalEntropy / PlusMinusEntropy .f/.d/.dd, functions.py:508-520,
//      544-564, with their safelog): one workgroup per image
__global__ __launch_bounds__(256)
void entropy_kernel(int kind, int n, const double* __restrict__ H, const double* __restrict__ D,
                    double* __restrict__ outS, double* __restrict__ outd, double* __restrict__ outdd)
{
    __shared__ double red[4];
    const size_t p = blockIdx.x;
    auto slog = [](double x) { return log(fabs(x) > 1e-100 ? x : 1e-100); };
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 256) {
        const double h = H[p * n + i], d = D[i];
        double Sv, dS, wsum;
        if (kind == 0) {
            Sv = h - d - h * slog(h / d);
            dS = -(slog(h) - slog(d));
            wsum = h;
        } else {
            const double r = sqrt(h * h + 4.0 * d * d);
            const double hp = 0.5 * (r + h), hm = 0.5 * (r - h);
            Sv = (hp - d - hp * slog(hp / d)) + (hm - d - hm * slog(hm / d));
            dS = -(slog(hp) - slog(d));
            wsum = hp + hm;
        }
        if (!(fabs(wsum) > 1e-100)) wsum = 1e-100;
        acc[0] += Sv;
        if (outd) outd[p * n + i] = dS;
        if (outdd) outdd[p * n + i] = -1.0 / wsum;
    }
    eval_block_sum<1>(acc, red);
    if (threadIdx.x == 0 && outS) outS[p] = acc[0];
}

} // namespace mxe

// ---- the default analyzer on the device: LineFitAnalyzer's alpha (linefit_analyzer.py:28-87, 151-183) --
// Two-piece fit of log chi2 (log alpha): a line through the first i points, a constant (p2_deg = 0) or a
// line (p2_deg = 1) through the rest, i = 2 .. n-3 chosen for the smallest summed squared misfit; the answer
// is the index of the alpha closest to the intersection.  NaN chi2 are left out.  One wavefront per scan:
// lane t evaluates the break points t, t + 64, ... with the two-pass formulas of the host analyzer
// (maxent_amd/analyzers.py: _linfit_sse), then the H of the chosen alpha is copied out, so that a caller
// (or a gather between GPUs) moves one row per scan instead of all of them.
namespace mxe {

struct LineSSE { double slope, icpt, sse; int n; };

// least-squares line (or constant) through the points lo .. hi-1 that are not NaN, by the whole wavefront:
// lane t takes the points lo + t, lo + t + 64, ...; two-pass formulas (mean first, then centred sums)
__device__ inline LineSSE line_sse(const double* x, const double* y, int lo, int hi, bool line)
{
    const int lane = threadIdx.x & 63;
    LineSSE r; r.slope = 0.0; r.icpt = 0.0; r.sse = 0.0; r.n = 0;
    double sx = 0.0, sy = 0.0, cnt = 0.0;
    for (int k = lo + lane; k < hi; k += 64) if (y[k] == y[k]) { sx += x[k]; sy += y[k]; cnt += 1.0; }
    sx = wave_sum(sx); sy = wave_sum(sy); cnt = wave_sum(cnt);
    r.n = (int)cnt;
    if (r.n < 1) return r;
    const double xm = sx / cnt, ym = sy / cnt;
    double sxx = 0.0, sxy = 0.0;
    for (int k = lo + lane; k < hi; k += 64) if (y[k] == y[k]) { const double dx = x[k] - xm; sxx += dx * dx; sxy += dx * (y[k] - ym); }
    sxx = wave_sum(sxx); sxy = wave_sum(sxy);
    if (line && r.n >= 2 && sxx != 0.0) { r.slope = sxy / sxx; r.icpt = ym - r.slope * xm; }
    else r.icpt = ym;
    double e = 0.0;
    for (int k = lo + lane; k < hi; k += 64) if (y[k] == y[k]) { const double d = y[k] - (r.slope * x[k] + r.icpt); e += d * d; }
    r.sse = wave_sum(e);
    return r;
}

// One workgroup of FOUR waves per scan (round 5; one wave until then: 19.4 us per launch of 256 scans, a serial chain of
// running sums, candidate fits and row copies in one wave).  Every number that decides a pick is formed by the arithmetic of the
// one-wave version in the same order -- the running sums per quantity, the exact misfits per candidate -- so the picks are the
// same; what is spread over the waves is WHICH quantity / candidate / analyzer a wave takes, and the row copies.
__global__ __launch_bounds__(256)
void linefit_kernel(const double* __restrict__ alpha, const double* __restrict__ chi2, const double* __restrict__ H,
                    int n_alpha, int nw, int p2_deg, double* __restrict__ out_sel /*[n_chain][nw] or null*/,
                    double* __restrict__ out_idx /*[n_chain], as doubles (they travel in the result pack)*/,
                    // the two other default analyzers of the reference, when out3_idx is given (mxe_select3_launch):
                    const double* __restrict__ S = nullptr, double gamma = 0.2,
                    double* __restrict__ out3_idx = nullptr /*[3][n_chain]: line fit, chi2 curvature, entropy*/,
                    double* __restrict__ out3_sel = nullptr /*[3][n_chain][nw]: the H rows of the three*/)
{
    // Break points are first ranked with running sums (O(n) for all of them; centred, like the host
    // analyzer's fit_piecewise), then only those within a whisker of the best are evaluated exactly with the
    // two-pass formulas -- the same two stages, with the same tolerance, as the host code.
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = n_alpha;
    double* x = sm;                  // log alpha
    double* y = x + n;               // log chi2
    double* cs = y + n;              // [6][n + 1] running sums of w, w x0, yc, w x0^2, x0 yc, yc^2
    double* ap = cs + 6 * (n + 1);   // [n] approximate misfit of break point i
    double* x10 = ap + n;            // [n] gamma log10 alpha   (chi2 curvature)
    double* y10 = x10 + n;           // [n] log10 chi2
    double* wbest = y10 + n;         // [4] smallest approximate misfit seen by a wave
    double* wfit = wbest + 4;        // [4][6] a wave's best exact candidate: misfit, index, the two lines
    __shared__ int s_pick[3];
    const size_t c = blockIdx.x;
    for (int k = tid; k < n; k += 256) {
        const double a = alpha[c * n + k], q = chi2[c * n + k];
        x[k] = log(a); y[k] = log(q);
        x10[k] = gamma * log10(a); y10[k] = log10(q);
    }
    if (tid < 3) s_pick[tid] = -1;
    __syncthreads();
    const bool fit = n > 4;
    double xm = 0.0, ym = 0.0;
    if (fit) {
        // means for the centring (every wave for itself: the same sums in the same order)
        double sx = 0.0, sy = 0.0, sw = 0.0;
        for (int k = lane; k < n; k += 64) { sx += x[k]; if (y[k] == y[k]) { sy += y[k]; sw += 1.0; } }
        sx = wave_sum(sx); sy = wave_sum(sy); sw = wave_sum(sw);
        xm = sx / n; ym = sy / fmax(sw, 1.0);
        // running sums of the six quantities, quantity q by wave q & 3: lane t holds element c0 + t of a chunk of 64,
        // inclusive scan over the lanes (six shuffle steps), carry from chunk to chunk
        for (int q = wave; q < 6; q += 4) {
            double carry = 0.0;
            if (lane == 0) cs[q * (n + 1)] = 0.0;
            for (int c0 = 0; c0 < n; c0 += 64) {
                const int k = c0 + lane;
                const bool in = k < n, ok = in && y[in ? k : 0] == y[in ? k : 0];
                const double w = ok ? 1.0 : 0.0, x0 = in ? x[k] - xm : 0.0, yc = ok ? y[k] - ym : 0.0;
                double v = q == 0 ? w : q == 1 ? w * x0 : q == 2 ? yc : q == 3 ? w * x0 * x0 : q == 4 ? x0 * yc : yc * yc;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double up = __shfl_up(v, off);
                    if (lane >= off) v += up;
                }
                v += carry;
                if (in) cs[q * (n + 1) + k + 1] = v;
                carry = __shfl(v, 63);
            }
        }
    }
    __syncthreads();
    auto sse = [&](int lo, int hi, bool line) -> double {
        const double m = cs[hi] - cs[lo], s1 = cs[(n + 1) + hi] - cs[(n + 1) + lo], s2 = cs[2 * (n + 1) + hi] - cs[2 * (n + 1) + lo];
        const double sxx = cs[3 * (n + 1) + hi] - cs[3 * (n + 1) + lo], sxy = cs[4 * (n + 1) + hi] - cs[4 * (n + 1) + lo];
        const double syy = cs[5 * (n + 1) + hi] - cs[5 * (n + 1) + lo];
        if (!(m >= 1.0)) return __builtin_nan("");
        const double vyy = syy - s2 * s2 / m;
        if (!line || m < 2.0) return vyy;
        const double vxx = sxx - s1 * s1 / m, vxy = sxy - s1 * s2 / m;
        return vxx > 0.0 ? vyy - vxy * vxy / vxx : vyy;
    };
    if (fit) {
        double best = __builtin_inf();
        for (int i = 2 + tid; i < n - 2; i += 256) {
            const double a = sse(0, i, true) + sse(i, n, p2_deg == 1);
            ap[i] = a;
            if (a == a && fabs(a) < 1.7e308 && a < best) best = a;
        }
        best = -wave_max(-best);
        if (lane == 0) wbest[wave] = best;
    }
    __syncthreads();
    if (fit) {
        const double best = fmin(fmin(wbest[0], wbest[1]), fmin(wbest[2], wbest[3]));
        const double tol = best + 1e-9 * (fabs(cs[5 * (n + 1) + n]) + 1e-300) + 1e-6 * fabs(best);
        const bool rank_ok = best < 1.7e308 && n > 8;
        // exact misfit of the candidates, candidate number j by wave j & 3, each by its whole wave; the smallest wins,
        // the lowest index among equals (np.nanargmin)
        double ebest = __builtin_inf(); int ebest_i = -1, seen = 0;
        LineSSE abest, bbest;
        abest.slope = abest.icpt = bbest.slope = bbest.icpt = 0.0;
        for (int i = 2; i < n - 2; ++i) {
            const bool cand = rank_ok ? (ap[i] == ap[i] && ap[i] <= tol) : true;      // wave-uniform
            if (!cand) continue;
            const bool mine = (seen & 3) == wave;
            ++seen;
            if (!mine) continue;
            const LineSSE a = line_sse(x, y, 0, i, true), b = line_sse(x, y, i, n, p2_deg == 1);
            if (a.n < 1 || b.n < 1) continue;
            const double m = a.sse + b.sse;
            if (m == m && m < ebest) { ebest = m; ebest_i = i; abest = a; bbest = b; }
        }
        if (lane == 0) {
            double* f = wfit + wave * 6;
            f[0] = ebest; f[1] = (double)ebest_i; f[2] = abest.slope; f[3] = abest.icpt; f[4] = bbest.slope; f[5] = bbest.icpt;
        }
    }
    __syncthreads();
    if (wave == 0 && fit) {
        double ebest = __builtin_inf(); int ebest_i = -1, wv = -1;
        for (int w = 0; w < 4; ++w) {
            const double m = wfit[w * 6]; const int i = (int)wfit[w * 6 + 1];
            if (i >= 0 && (m < ebest || (m == ebest && i < ebest_i))) { ebest = m; ebest_i = i; wv = w; }
        }
        if (ebest_i >= 0) {
            const double* f = wfit + wv * 6;
            const double xc = (f[5] - f[3]) / (f[2] - f[4]);
            double dbest = __builtin_inf(); int di = -1;
            for (int k = lane; k < n; k += 64) {
                const double d = fabs(x[k] - xc);
                if (d == d && (d < dbest || (d == dbest && k < di))) { dbest = d; di = k; }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                const double od = __shfl_xor(dbest, off); const int oi = __shfl_xor(di, off);
                if (oi >= 0 && (di < 0 || od < dbest || (od == dbest && oi < di))) { dbest = od; di = oi; }
            }
            if (lane == 0) s_pick[0] = di;
        }
    }
    // ---- Chi2CurvatureAnalyzer (reference analyzers/chi2_curvature_analyzer.py:25-49, 101-131): the alpha of the
    //      largest curvature y'' / (1 + y'^2)^(3/2) of y = log10 chi2 over x = gamma log10 alpha, second-order
    //      central differences on the (non-uniform) mesh, NaN ignored, the first of equal maxima.          (wave 1)
    // ---- EntropyAnalyzer (entropy_analyzer.py:72-103): the alpha where (dS / dlog alpha)^2 is smallest.   (wave 2)
    if (out3_idx && (wave == 1 || wave == 2)) {
        double vbest = (wave == 1) ? -__builtin_inf() : __builtin_inf();
        int vi = -1;
        for (int k = 1 + lane; k < n - 1; k += 64) {
            double val;
            if (wave == 1) {
                const double hp = x10[k + 1] - x10[k], hm = x10[k] - x10[k - 1];
                const double der2 = (y10[k + 1] - 2 * y10[k] + y10[k - 1]) / (hp * hm);
                const double der1 = ((y10[k + 1] - y10[k]) / hp + (y10[k] - y10[k - 1]) / hm) / 2;
                const double q = 1 + der1 * der1;
                val = der2 / (q * sqrt(q));
                if (val == val && (val > vbest || vi < 0)) { vbest = val; vi = k; }          // (ascending k per lane: the first maximum)
            } else {
                const double dS = (S[c * n + k + 1] - S[c * n + k - 1]) / (x[k + 1] - x[k - 1]);
                val = dS * dS;
                if (val == val && (val < vbest || vi < 0)) { vbest = val; vi = k; }
            }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(vbest, off); const int oi = __shfl_xor(vi, off);
            const bool better = (wave == 1) ? (ov > vbest) : (ov < vbest);
            if (oi >= 0 && (vi < 0 || better || (ov == vbest && oi < vi))) { vbest = ov; vi = oi; }
        }
        if (lane == 0) s_pick[wave] = vi;
    }
    __syncthreads();
    const int idx = s_pick[0];
    if (tid == 0) out_idx[c] = (double)idx;
    if (out_sel) {
        const double* row = H + (c * n + (idx >= 0 ? idx : 0)) * nw;
        for (int k = tid; k < nw; k += 256) out_sel[c * nw + k] = (idx >= 0) ? row[k] : __builtin_nan("");
    }
    if (!out3_idx) return;
    const size_t nc = gridDim.x;
    if (tid < 3) out3_idx[tid * nc + c] = (double)s_pick[tid];
    if (out3_sel) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int pk = s_pick[a];
            const double* row = H + (c * n + (pk >= 0 ? pk : 0)) * nw;
            double* dst = out3_sel + ((size_t)a * nc + c) * nw;
            for (int k = tid; k < nw; k += 256) dst[k] = (pk >= 0) ? row[k] : __builtin_nan("");
        }
    }
}

} // namespace mxe
