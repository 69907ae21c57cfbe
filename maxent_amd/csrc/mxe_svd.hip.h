// mxe_svd.hip.h -- kernel matrix staging on the device
//
//   TauKernel._fill_values      kernels.py:244-271   -> tau_kernel_fill
//   get_preblur                 preblur.py:31-58     -> preblur_rows / preblur_cols / preblur_matrix
//   PreblurKernel._fill_values  kernels.py:384-393   -> preblur_product   (K' = K diag(delta) B)
//   KernelSVD.svd + reduce_singular_space kernels.py:53-122 -> svd_kernel
//
// One launch serves a whole batch of blur widths b (a "b-scan", doc/guide/preblur_example.py): one
// workgroup of 16 wavefronts per b does the decomposition, everything it touches (0.8 MB of K^T,
// 0.25 MB of R) stays in the L2 of its XCD.
//
// SVD algorithm (Drmac / Veselic style preconditioned one-sided Jacobi, all binary64):
//   1. Householder QR with column pivoting of K (n_tau x n_omega), stopped at the first step whose
//      largest remaining column norm is below eps * (largest column norm of K): K P = Q R with R of
//      r ~ 60 rows (the kernel's singular values decay exponentially; what is left is rounding noise,
//      LAPACK's singular values below eps * sigma_max are noise too).
//   2. one-sided Jacobi (Hestenes) on the r rows of R in round-robin order, r/2 disjoint pairs per
//      round, one wavefront per pair: R = J diag(S) W^T.  Rows of R are graded by the pivoting, which
//      is what makes Jacobi converge in ~10 sweeps here (on K itself: > 30, measured with the numpy model).
//   3. U = Q J, V = P W, sorted by S, truncated at S >= threshold (absolute, like the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mxe {

constexpr int SVD_T = 1024;          // threads of the decomposition workgroup
constexpr int SVD_NWAVE = SVD_T / 64;
constexpr int SVD_RCAP = 128;        // most rows of R kept (= the solver's n_s limit)
#ifndef MXE_X_SVD_SWEEPS
#define MXE_X_SVD_SWEEPS 40
#endif
constexpr int SVD_MAX_SWEEPS = MXE_X_SVD_SWEEPS;

// K^T[j][i] = K(tau_i, omega_j)  (column-major K: one column = one contiguous run of n_tau values)
__global__ __launch_bounds__(256)
void tau_kernel_fill(const double* __restrict__ tau, const double* __restrict__ omega, double beta,
                     int n_tau, int n_omega, double* __restrict__ Kt)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_tau * n_omega) return;
    const int j = idx / n_tau, i = idx - j * n_tau;
    const double w = omega[j], t = tau[i];
    // two algebraically equal forms, each overflow-free on its half-axis (kernels.py:256-263)
    Kt[idx] = (w >= 0.0) ? -exp(-w * t) / (exp(-beta * w) + 1.0)
                         : -exp(w * (beta - t)) / (1.0 + exp(beta * w));
}

__device__ __forceinline__ double gauss_blur(double wi, double wj, double b)
{
    const double d = wj - wi;
    return exp(-d * d / 2.0 / (b * b)) / sqrt(2.0 * 3.141592653589793 * b * b);
}

// r1[i] = sum_k delta_k G[k][i]   (first normalisation of get_preblur: rows)
__global__ __launch_bounds__(256)
void preblur_rows(const double* __restrict__ omega, const double* __restrict__ delta, double b,
                  int n_omega, double* __restrict__ r1)
{
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < n_omega; k += 256) s += delta[k] * gauss_blur(omega[k], omega[i], b);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) r1[i] = red[0];
}

// d[j] = sum_c B1[j][c] delta_c with B1[j][c] = G[j][c] / r1[j]   (second normalisation: columns)
__global__ __launch_bounds__(256)
void preblur_cols(const double* __restrict__ omega, const double* __restrict__ delta, double b,
                  int n_omega, const double* __restrict__ r1, double* __restrict__ dcol)
{
    __shared__ double red[256];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int c = threadIdx.x; c < n_omega; c += 256) s += gauss_blur(omega[j], omega[c], b) / r1[j] * delta[c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) dcol[j] = red[0];
}

// B[j][l] = G[j][l] / r1[j] / d[l]   (the matrix get_preblur returns)
__global__ __launch_bounds__(256)
void preblur_matrix(const double* __restrict__ omega, double b, int n_omega,
                    const double* __restrict__ r1, const double* __restrict__ dcol, double* __restrict__ B)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_omega * n_omega) return;
    const int j = idx / n_omega, l = idx - j * n_omega;
    B[idx] = gauss_blur(omega[j], omega[l], b) / r1[j] / dcol[l];
}

// K'^T[l][i] = sum_j K^T[j][i] delta_j B[j][l]; block = one l, threads over i (coalesced K^T rows,
// the B column is block uniform)
__global__ __launch_bounds__(256)
void preblur_product(const double* __restrict__ Kt, const double* __restrict__ delta,
                     const double* __restrict__ B, int n_tau, int n_omega, double* __restrict__ Kbt)
{
    extern __shared__ double bcol[];          // [n_omega]  delta_j B[j][l]
    const int l = blockIdx.x;
    for (int j = threadIdx.x; j < n_omega; j += blockDim.x) bcol[j] = delta[j] * B[(size_t)j * n_omega + l];
    __syncthreads();
    for (int i = threadIdx.x; i < n_tau; i += blockDim.x) {
        double a0 = 0.0, a1 = 0.0;
        int j = 0;
        for (; j + 1 < n_omega; j += 2) {
            a0 = fma(Kt[(size_t)j * n_tau + i], bcol[j], a0);
            a1 = fma(Kt[(size_t)(j + 1) * n_tau + i], bcol[j + 1], a1);
        }
        if (j < n_omega) a0 = fma(Kt[(size_t)j * n_tau + i], bcol[j], a0);
        Kbt[(size_t)l * n_tau + i] = a0 + a1;
    }
}

__device__ __forceinline__ double svd_wave_sum(double x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

struct SvdParams {
    int m;              // n_tau   (rows of K)
    int n;              // n_omega (columns of K)
    int ns_max;         // capacity of the outputs
    double threshold;   // absolute cut on S
    // per batch item (stride = item index):
    double* A;          // [n][m]   K^T, destroyed (columns reduced in place)
    double* Vh;         // [RCAP][m] unit Householder vectors
    double* Rm;         // [RCAP][n] R, then diag(S) W^T
    double* Jt;         // [RCAP][RCAP] accumulated rotations (rows)
    double* Qc;         // [RCAP][m] columns of Q
    double* cn2;        // [n] remaining squared column norms
    int* perm;          // [n]
    double* out_U;      // [m][ns_max]
    double* out_S;      // [ns_max]
    double* out_V;      // [n][ns_max]
    int* out_info;      // [4]: n_s, rank of the QR stage, Jacobi sweeps, status (0 ok, 1 sweeps exhausted, 2 n_s > ns_max)
};

__global__ __launch_bounds__(SVD_T)
void svd_kernel(const SvdParams p)
{
    extern __shared__ double sm[];
    const int item = blockIdx.x;
    const int m = p.m, n = p.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* A  = p.A  + (size_t)item * n * m;
    double* Vh = p.Vh + (size_t)item * SVD_RCAP * m;
    double* Rm = p.Rm + (size_t)item * SVD_RCAP * n;
    double* Jt = p.Jt + (size_t)item * SVD_RCAP * SVD_RCAP;
    double* Qc = p.Qc + (size_t)item * SVD_RCAP * m;
    double* cn2 = p.cn2 + (size_t)item * n;
    int* perm = p.perm + (size_t)item * n;
    double* out_U = p.out_U + (size_t)item * m * p.ns_max;
    double* out_S = p.out_S + (size_t)item * p.ns_max;
    double* out_V = p.out_V + (size_t)item * n * p.ns_max;
    int* out_info = p.out_info + (size_t)item * 4;

    double* vk = sm;                        // [m] current Householder vector
    double* redv = vk + m;                  // [SVD_NWAVE] reduction values
    int* redi = reinterpret_cast<int*>(redv + SVD_NWAVE);   // [SVD_NWAVE] reduction indices
    double* s2 = reinterpret_cast<double*>(redi + SVD_NWAVE);   // [RCAP] squared singular values
    int* order = reinterpret_cast<int*>(s2 + SVD_RCAP);     // [RCAP]
    __shared__ int sh_piv, sh_stop, sh_rot;
    __shared__ double sh_nrm0;

    // ---- column norms ----
    for (int j = wave; j < n; j += SVD_NWAVE) {
        double s = 0.0;
        for (int i = lane; i < m; i += 64) { const double x = A[(size_t)j * m + i]; s = fma(x, x, s); }
        s = svd_wave_sum(s);
        if (lane == 0) { cn2[j] = s; perm[j] = j; }
    }
    __syncthreads();

    // ---- 1. Householder QR with column pivoting, early stop ----
    const int kmax = min(min(m, n), SVD_RCAP);
    int r = 0;
    for (int k = 0; k < kmax; ++k) {
        // pivot = argmax_{j >= k} cn2[j] (lowest index on ties)
        double best = -1.0; int bi = k;
        for (int j = k + tid; j < n; j += SVD_T) { const double c = cn2[j]; if (c > best) { best = c; bi = j; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { redv[wave] = best; redi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double bb = redv[0]; int ii = redi[0];
            for (int w2 = 1; w2 < SVD_NWAVE; ++w2)
                if (redv[w2] > bb || (redv[w2] == bb && redi[w2] < ii)) { bb = redv[w2]; ii = redi[w2]; }
            if (k == 0) sh_nrm0 = bb;
            const double eps = 2.220446049250313e-16;
            sh_stop = !(bb > eps * eps * sh_nrm0) || !(bb > 0.0);
            sh_piv = ii;
        }
        __syncthreads();
        if (sh_stop) break;
        const int piv = sh_piv;
        if (piv != k) {
            for (int i = tid; i < m; i += SVD_T) {
                const double x = A[(size_t)k * m + i];
                A[(size_t)k * m + i] = A[(size_t)piv * m + i];
                A[(size_t)piv * m + i] = x;
            }
            if (tid == 0) {
                const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
                cn2[piv] = cn2[k];
            }
        }
        __syncthreads();
        // Householder vector of column k, rows k..m-1 (wave 0)
        if (wave == 0) {
            double s = 0.0;
            for (int i = k + lane; i < m; i += 64) { const double x = A[(size_t)k * m + i]; s = fma(x, x, s); }
            s = svd_wave_sum(s);
            const double x0 = A[(size_t)k * m + k];
            const double nrm = sqrt(s);
            const double alpha = (x0 >= 0.0) ? -nrm : nrm;
            // v = x - alpha e0 ; |v|^2 = s - 2 alpha x0 + alpha^2 = 2 (s - alpha x0)
            const double vn = sqrt(2.0 * (s - alpha * x0));
            const double inv = (vn > 0.0) ? 1.0 / vn : 0.0;
            for (int i = lane; i < m; i += 64) {
                double v = 0.0;
                if (i >= k) v = ((i == k) ? (x0 - alpha) : A[(size_t)k * m + i]) * inv;
                vk[i] = v;
                Vh[(size_t)k * m + i] = v;
            }
            for (int i = k + lane; i < m; i += 64) A[(size_t)k * m + i] = (i == k) ? alpha : 0.0;
        }
        __syncthreads();
        // apply H_k = I - 2 v v^T to the columns j > k, one wavefront per column, four columns in
        // flight per wavefront (the loads of a column come from L2: latency, not bandwidth);
        // new remaining norms
        for (int j0 = k + 1 + wave; j0 < n; j0 += 4 * SVD_NWAVE) {
            constexpr int MR = 16;                       // rows per lane held in registers: m <= 64 * MR
            double s[4], rem[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s[q] = 0.0;
                const int j = min(j0 + q * SVD_NWAVE, n - 1);
                const double* col = A + (size_t)j * m;
                for (int i = k + lane; i < m; i += 64) s[q] = fma(vk[i], col[i], s[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = 2.0 * svd_wave_sum(s[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rem[q] = 0.0;
                const int j = j0 + q * SVD_NWAVE;
                if (j < n) {
                    double* col = A + (size_t)j * m;
                    for (int i = k + lane; i < m; i += 64) {
                        const double x = fma(-s[q], vk[i], col[i]);
                        col[i] = x;
                        if (i > k) rem[q] = fma(x, x, rem[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rem[q] = svd_wave_sum(rem[q]);
                const int j = j0 + q * SVD_NWAVE;
                if (lane == 0 && j < n) cn2[j] = rem[q];
            }
            (void)MR;
        }
        __syncthreads();
        r = k + 1;
    }
    __syncthreads();

    // ---- R (r x n, row-major) and J^T = I ----
    const int rr = (r + 1) & ~1;              // even number of rows (a zero row pads an odd r)
    for (int idx = tid; idx < rr * n; idx += SVD_T) {
        const int kk = idx / n, j = idx - kk * n;
        Rm[idx] = (kk < r && j >= kk) ? A[(size_t)j * m + kk] : 0.0;
    }
    for (int idx = tid; idx < rr * SVD_RCAP; idx += SVD_T) {
        const int kk = idx / SVD_RCAP, l = idx - kk * SVD_RCAP;
        Jt[idx] = (kk == l) ? 1.0 : 0.0;
    }
    __syncthreads();

    // ---- 2. one-sided Jacobi on the rows of R, round-robin pairs, one wavefront per pair ----
    int sweeps = 0, status = 0;
    if (rr >= 2) {
        status = 1;
        for (int sweep = 0; sweep < SVD_MAX_SWEEPS; ++sweep) {
            if (tid == 0) sh_rot = 0;
            __syncthreads();
            for (int round = 0; round < rr - 1; ++round) {
                for (int pi = wave; pi < rr / 2; pi += SVD_NWAVE) {
                    int a, b;
                    if (pi == 0) { a = rr - 1; b = round % (rr - 1); }
                    else { a = (round + pi) % (rr - 1); b = (round - pi + 2 * (rr - 1)) % (rr - 1); }
                    const int pp = min(a, b), qq = max(a, b);
                    double* x = Rm + (size_t)pp * n;
                    double* y = Rm + (size_t)qq * n;
                    double aa = 0.0, bb = 0.0, gg = 0.0;
                    for (int j = lane; j < n; j += 64) {
                        const double xv = x[j], yv = y[j];
                        aa = fma(xv, xv, aa); bb = fma(yv, yv, bb); gg = fma(xv, yv, gg);
                    }
                    aa = svd_wave_sum(aa); bb = svd_wave_sum(bb); gg = svd_wave_sum(gg);
                    const double eps = 2.220446049250313e-16;
                    if (aa > 0.0 && bb > 0.0 && fabs(gg) > eps * sqrt(aa * bb)) {
                        const double zeta = (bb - aa) / (2.0 * gg);
                        const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                        for (int j = lane; j < n; j += 64) {
                            const double xv = x[j], yv = y[j];
                            x[j] = cs * xv - sn * yv;
                            y[j] = sn * xv + cs * yv;
                        }
                        double* jx = Jt + (size_t)pp * SVD_RCAP;
                        double* jy = Jt + (size_t)qq * SVD_RCAP;
                        for (int l = lane; l < r; l += 64) {
                            const double xv = jx[l], yv = jy[l];
                            jx[l] = cs * xv - sn * yv;
                            jy[l] = sn * xv + cs * yv;
                        }
                        if (lane == 0) atomicAdd(&sh_rot, 1);
                    }
                }
                __syncthreads();
            }
            sweeps = sweep + 1;
            const int nrot = sh_rot;
            __syncthreads();
            if (nrot == 0) { status = 0; break; }
        }
    }

    // ---- 3. singular values, order, truncation ----
    for (int kk = wave; kk < rr; kk += SVD_NWAVE) {
        double s = 0.0;
        for (int j = lane; j < n; j += 64) { const double x = Rm[(size_t)kk * n + j]; s = fma(x, x, s); }
        s = svd_wave_sum(s);
        if (lane == 0) s2[kk] = s;
    }
    __syncthreads();
    if (tid < rr) {
        const double mine = s2[tid];
        int rank = 0;
        for (int l = 0; l < rr; ++l) { const double o = s2[l]; if (o > mine || (o == mine && l < tid)) ++rank; }
        order[rank] = tid;
    }
    __syncthreads();
    int ns = 0;
    for (int kk = 0; kk < rr; ++kk) if (sqrt(s2[order[kk]]) >= p.threshold && s2[order[kk]] > 0.0) ns = kk + 1; else break;
    if (ns > p.ns_max) { ns = p.ns_max; status = 2; }

    // Q = H_0 ... H_{r-1} [I_r ; 0], columns in Qc[l][i]
    for (int idx = tid; idx < r * m; idx += SVD_T) {
        const int l = idx / m, i = idx - l * m;
        Qc[idx] = (i == l) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int k = r - 1; k >= 0; --k) {
        for (int i = tid; i < m; i += SVD_T) vk[i] = Vh[(size_t)k * m + i];
        __syncthreads();
        for (int l = wave; l < r; l += SVD_NWAVE) {
            double* col = Qc + (size_t)l * m;
            double s = 0.0;
            for (int i = k + lane; i < m; i += 64) s = fma(vk[i], col[i], s);
            s = 2.0 * svd_wave_sum(s);
            for (int i = k + lane; i < m; i += 64) col[i] = fma(-s, vk[i], col[i]);
        }
        __syncthreads();
    }
    // outputs
    for (int kk = tid; kk < p.ns_max; kk += SVD_T) out_S[kk] = (kk < ns) ? sqrt(s2[order[kk]]) : 0.0;
    for (int idx = tid; idx < n * p.ns_max; idx += SVD_T) {
        const int j = idx / p.ns_max, kk = idx - j * p.ns_max;
        double val = 0.0;
        if (kk < ns) { const int row = order[kk]; val = Rm[(size_t)row * n + j] / sqrt(s2[row]); }
        out_V[(size_t)perm[j] * p.ns_max + kk] = val;
    }
    for (int idx = tid; idx < m * p.ns_max; idx += SVD_T) {
        const int i = idx / p.ns_max, kk = idx - i * p.ns_max;
        double val = 0.0;
        if (kk < ns) {
            const double* jrow = Jt + (size_t)order[kk] * SVD_RCAP;
            for (int l = 0; l < r; ++l) val = fma(Qc[(size_t)l * m + i], jrow[l], val);
        }
        out_U[idx] = val;
    }
    if (tid == 0) { out_info[0] = ns; out_info[1] = r; out_info[2] = sweeps; out_info[3] = status; }
}

} // namespace mxe
