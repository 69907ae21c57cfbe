/* maxent_hip.h -- C-ABI of libmaxent_hip.so (MI355X / gfx950)
 *
 * Drop-in boundary for the alpha-scan inner solver of TRIQS/maxent.
 * The reference (pure Python, /root/reference/python) has no FFI; the natural
 * hook for a batched device solver is the body of
 *     MaxEntLoop.run                      maxent_loop.py:144-302  (one alpha scan)
 *     ElementwiseMaxEnt.run_diagonal /
 *                       run_offdiagonal   elementwise_maxent.py:223-268 (all elements)
 * i.e. everything between "K.reduce_singular_space()" (maxent_loop.py:184) and
 * "result.add_result(...)" (maxent_loop.py:258-266).  The entry points below
 * replace exactly that span:
 *
 *   reference                                              | this library
 *   -------------------------------------------------------+---------------------------
 *   KernelSVD.U/.S/.V after reduce_singular_space          | mxe_ctx_create(U,S,V)
 *     kernels.py:53-122                                    |
 *   NormalChi2(K, G, err)            functions.py:336-377  | mxe_dataset_add (U,err) +
 *   TauMaxEnt.set_cov rotation       tau_maxent.py:253-325 |   mxe_elements_set (G, D, kind)
 *   NormalEntropy / PlusMinusEntropy functions.py:491-564  |
 *   NormalH_of_v / PlusMinusH_of_v   functions.py:720-796  |
 *   for alpha in alpha_mesh:                               | mxe_solve_chains
 *      cost_function.set_alpha       maxent_loop.py:243    |   (one chain = one element's
 *      minimizer.minimize            maxent_loop.py:245    |    warm-started alpha scan)
 *        LevenbergMinimizer.minimize levenberg_minimizer.py:123-248
 *        MaxEntCostFunction.f/d/dd   maxent_cost_function.py:68-165
 *        BryanCostFunction.f/d/dd    bryan_cost_function.py:57-128
 *      Q_min = cost_function(v)      maxent_loop.py:246    | out_v/out_H/out_chi2/out_S/out_Q
 *      minimizer.n_iter_last/.converged  maxent_loop.py:254-256 | out_niter/out_converged
 *      self.probability(Q_min)       maxent_loop.py:258-264 | mxe_logdet (the determinant of
 *        NormalLogProbability        probabilities.py:60-85 |   the posterior curvature)
 *   PreblurA_of_H.f  A = B H         functions.py:999-1001 | mxe_apply_output_map
 *   result.analyze(analyzers): LineFitAnalyzer,            | mxe_select3_launch / mxe_select3_fetch
 *     Chi2CurvatureAnalyzer, EntropyAnalyzer               |   (alpha_index and the H row of each,
 *     maxent_result.py:793-822, analyzers/               |    for every scan of the launch)
 *   CostFunction.__call__ / .f / .d / .dd at a given v     | mxe_eval_batch, mxe_entropy, mxe_audit
 *     cost_function.py:73-85, maxent_cost_function.py:68-165
 *   TauKernel / PreblurKernel fill + KernelSVD.svd         | mxe_kernel_svd (optional: the host
 *     kernels.py:53-122,244-271,384-393                    |   numpy path stays the default)
 *   the arrays of MaxEntResult (numpy allocations)         | mxe_host_alloc / mxe_host_free (optional:
 *     maxent_result.py:835-967                             |   page-locked destinations, one DMA per fetch)
 *
 * Conventions: plain C, no C++ types; every function returns 0 (MXE_OK) or a
 * negative error code and never throws or aborts; the caller owns every host
 * buffer (C-contiguous, row-major); the library owns device memory inside
 * mxe_ctx.  Calls on one ctx are not re-entrant; different ctxs (devices) may
 * be driven from different threads.  All arithmetic is IEEE binary64 unless
 * mxe_opts.precision asks for the binary32 streaming variant.
 */
#ifndef MAXENT_HIP_H
#define MAXENT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MXE_OK               0
#define MXE_ERR_ARG         -1   /* bad argument (NULL, size, index)          */
#define MXE_ERR_HIP         -2   /* a HIP runtime call failed (mxe_last_hip_error) */
#define MXE_ERR_NODEVICE    -3   /* no usable gfx950 device                   */
#define MXE_ERR_STATE       -4   /* call order (e.g. solve before elements)   */
#define MXE_ERR_LIMIT       -5   /* n_s > 128 (fp32: > 64); mxe_eval_batch / mxe_audit: w and H of a problem
                                     exceed the LDS (n_omega > ~7900); the alpha scans themselves take any
                                     n_omega (state in device memory where the LDS does not hold it) */
#define MXE_ERR_NUMERIC     -6   /* whitening failed (non-positive error bar) / SVD sweeps exhausted */
#define MXE_ERR_NOMEM       -7   /* host allocation failed                     */

#define MXE_PRECISION_F64      0 /* all arithmetic IEEE binary64 (default)      */
#define MXE_PRECISION_F32      1 /* omega-space streaming arithmetic in binary32 */

#define MXE_ENTROPY_NORMAL     0 /* NormalEntropy + NormalH_of_v              */
#define MXE_ENTROPY_PLUSMINUS  1 /* PlusMinusEntropy + PlusMinusH_of_v        */

typedef struct mxe_ctx mxe_ctx;

/* Options of the per-alpha minimiser.  Fill with mxe_opts_default() first.
 * Counterparts: LevenbergMinimizer.__init__ (levenberg_minimizer.py:92-121)
 * and convergence_methods.py:81-122. */
typedef struct mxe_opts {
    int32_t maxiter;      /* max Newton iterations per alpha (reference: 1000)            */
    int32_t miniter;      /* reference: 0                                                 */
    double  tol_h;        /* converged when the Newton correction satisfies
                             ||dH||_2 / ||H||_2 < tol_h   (0 = off); with stop_estimate
                             the ESTIMATED next correction is tested as well             */
    double  tol_d;        /* MaxDerivativeConvergenceMethod: max|W g| < tol_d (0 = off; reference default
                             1e-4).  The maximum runs over the coupled block (rows and columns of the
                             directions with c_k^2 max(w) > decouple_tol * alpha): the gradient
                             components of the decoupled directions are zeroed by their own diagonal
                             Newton step to relative decouple_tol and add nothing at that level; with
                             decouple_tol = 0 it is the reference's max over all n_s              */
    double  tol_relq;     /* RelativeFunctionChangeConvergenceMethod: |Q0-Q1|/|Q1| <
                             tol_relq (0 = off; reference default 1e-16)                  */
    double  step_max;     /* step bound  delta^T W delta <= step_max * sum(D)  (0.2)      */
    double  mu_first;     /* first non-zero damping, in units of alpha (1e-3)             */
    double  mu_grow;      /* growth factor of the damping on a rejected step (4)          */
    double  mu_max;       /* give up on the alpha when mu/alpha exceeds this (1e20)       */
    double  decouple_tol; /* theta: singular directions with c_k^2 max(w) <= theta*alpha
                             take the diagonal Newton step (1e-5; 0 = full n_s block)      */
    int32_t waves_per_chain; /* one-chain-per-workgroup layout: wavefronts per chain;
                                0 = choose from the chain count; else 1,2,4,8             */
    int32_t chains_per_wg;   /* 0 = auto; 1 = one chain per workgroup; 4 = four chains of
                                one data set per workgroup in lock-step (shared V loads)   */
    int32_t alpha_split;     /* cut every alpha scan into this many cold-started pieces
                                (more chains to fill the GPU; results are path independent);
                                0 = auto: about two pieces per chain slot of the GPU, a piece
                                of a normal-entropy scan counting twice, none shorter than two
                                alphas (binary32 variant: at most 16, none shorter than six);
                                1 = never.  In either case a piece of a normal-entropy scan
                                that would start among the scan's smallest alphas (the last 6 %
                                of its logarithmic range, where a cold start from the default
                                model can take hundreds of iterations) solves the last alpha above
                                that range cold instead and walks down the mesh to its own alpha with a
                                loose tolerance (lock-step layout: every alpha of that range is a piece
                                of its own; one-chain layout: joined to the piece before) */
    int32_t stop_estimate;   /* 1 (default): after a full (undamped) Newton step the next
                                correction is estimated as (expm1(max|du|) + decouple_tol) * ||dH||/||H||
                                (the relative change of the weights w bounds the relative
                                change of the Jacobian) and tol_h is applied to it, which
                                saves the last, verifying iteration; 0: tol_h is applied to
                                the correction just taken only                              */
    int32_t precision;       /* MXE_PRECISION_F64 (default) or MXE_PRECISION_F32: V, u = V v, w, H,
                                exp, h = V^T H and the Gram matrix in binary32 (fp32 MFMA); the
                                n_act x n_act Newton system, the residual and all scalars stay
                                binary64.  An alpha also stops when its Newton correction has
                                reached the rounding floor (it no longer shrinks).  n_s <= 64.
                                Where V^T fits the LDS as binary32 the launch runs in the lock-step
                                kernel chain_kernel_lv (mxe_opts.lds_basis), else one chain per
                                workgroup.  A request, not a promise: binary32 is asked for as the cheaper
                                arithmetic, and a launch for which it is not is PROMOTED to binary64 -- a
                                job with an alpha that couples more than 32 directions (error bars far
                                below the noise of the data: the lock-step build with the 64-row block),
                                and, with wg_per_cu = 0, a batch that fills the GPU at two workgroups per
                                CU (the binary64 kernel that runs that way: 0.81 against 1.24 ms on the
                                16 x 16 x 100 batch; wg_per_cu = 1 keeps chain_kernel_lv), and a frequency
                                mesh whose basis does not fit the LDS as binary32 (n_omega > 512: one chain
                                per workgroup in binary32 takes 3.6-7.8 ms where the binary64 lock-step
                                kernel takes 0.5-1.7; lds_basis = 2 or chains_per_wg = 1 keep that kernel);
                                mxe_last_launch_info names the kernel that ran.  For the fp32-vs-fp64 tolerance sweep of BASELINE config 5
                                (tools/cfg5_tolerance_sweep.py)                                 */
    int32_t wg_per_cu;       /* lock-step layout: workgroups per CU.  0 = auto (2 where the kernel has a
                                build for it: n_s <= 64 active block 32, n_omega <= 512 -- u then lives in
                                registers and two workgroups of 73 KB share a CU, so that the serial
                                sections of one overlap the streaming passes of the other), 1, 2   */
    double  chi2_factor;     /* eta in Q = eta chi2 / 2 - alpha S (CostFunction(chi2_factor=...),
                                cost_function.py:60, bryan_cost_function.py:71); default 1       */
    int32_t lds_basis;       /* lock-step layout with V^T resident in the LDS as binary32 (chain_kernel_lv; it needs
                                n_s * (n_omega_pad + 4) * 4 B + 47 KB <= 160 KB: the BASELINE grids just fit).
                                0 = auto: it IS the launch for precision = MXE_PRECISION_F32 (where the basis fits) and is
                                NOT used by binary64 launches;
                                1 = opt-in: also the first pass of every binary64 launch (every alpha to 1e-5 in binary32,
                                then one binary64 Newton step per alpha in the lock-step kernel, all alphas side by side) --
                                correct, measured NOT faster than the plain binary64 launch, hence not auto;
                                2 = never                                                                        */
    int32_t in_flight;       /* batches of this size the caller keeps in flight on the GPU (launches of several
                                contexts enqueued without waiting in between): 0 or 1 = one -- the scans are cut into
                                enough cold-started pieces to fill the GPU on their own --, n > 1: into 1 / n as many
                                (a cold start costs 4-17 evaluations: with n batches side by side the GPU is full
                                without them; bench.py --in-flight 4: 0.83 -> 0.65 ms per batch of 25 600 alpha-solves).
                                The latency of ONE batch grows (1.7 ms for n = 4)                                */
} mxe_opts;

/* ---- library / device ------------------------------------------------- */
const char* mxe_version(void);
/* first 16 hex digits of the SHA-256 over the library's sources (the .hip and .hip.h files of csrc/, this header) as the Makefile
   saw them when it built this binary: bench.py only trusts a counter profile under profiles/ that records the same hash */
const char* mxe_source_hash(void);
/* Page-locked host memory for result arrays (the destination of mxe_chains_fetch / mxe_fetch_rows / mxe_select3_fetch may be
   any host memory; into a block of this call the copy is one DMA at the rate of the link -- all H of the BASELINE batch,
   102 MB: 3 ms against 11-22 ms into pageable memory).  The library keeps a few freed blocks for the next call.  NULL when
   the runtime cannot pin that much.  What they replace in the reference: numpy's allocation of MaxEntResult's arrays
   (maxent_result.py:835-967). */
void* mxe_host_alloc(size_t bytes);
void  mxe_host_free(void* block);
const char* mxe_strerror(int code);
int  mxe_device_count(int* n_devices);
void mxe_opts_default(mxe_opts* opts);

/* ---- context: one per device; holds the truncated SVD of the kernel ---- */
/* U: n_tau x n_s, S: n_s, V: n_omega x n_s (row-major, as KernelSVD.U/.S/.V
 * return them, kernels.py:66-94).  U is only used as the default data set 0
 * together with `err` given to mxe_dataset_add; it may be NULL if every data
 * set brings its own (rotated) U. */
int  mxe_ctx_create(int device, int n_tau, int n_omega, int n_s,
                    const double* U, const double* S, const double* V,
                    mxe_ctx** out);
void mxe_ctx_destroy(mxe_ctx* ctx);
const char* mxe_last_hip_error(mxe_ctx* ctx);

/* ---- data sets: a (U, err) pair = one whitened singular basis ---------- */
/* U_rot: n_rows x n_s left factor in the (possibly covariance-rotated) data
 * space (Kernel.transform, kernels.py:160-180); NULL = the ctx's U (then
 * n_rows must equal n_tau).  err: n_rows standard deviations
 * (TauMaxEnt.set_error / set_cov, tau_maxent.py:227-288).
 * Returns the data-set id (>= 0) in *id. */
int  mxe_dataset_add(mxe_ctx* ctx, int n_rows, const double* U_rot,
                     const double* err, int* id);
int  mxe_dataset_clear(mxe_ctx* ctx);

/* ---- elements: the matrix elements of G that will be continued -------- */
/* dataset_of_elem[e]: data-set id; G: concatenated data vectors, element e
 * has n_rows(dataset) entries starting at G_offset[e] (already rotated into
 * the data set's space); D: n_elem x n_omega default models INCLUDING
 * delta-omega (default_models.py:61-63); entropy[e]: MXE_ENTROPY_*. */
int  mxe_elements_set(mxe_ctx* ctx, int n_elem, const int32_t* dataset_of_elem,
                      const double* G, const int64_t* G_offset,
                      const double* D, const int32_t* entropy);
/* New DATA for the elements that are set (same number, same data sets, default models and entropies): G as above.  Only the
 * projections of the data change on the device; the staged chains (mxe_chains_upload: their cut into pieces, the start
 * states) do not depend on G and stay ready -- what every iteration of a self-consistency loop does to an
 * ElementwiseMaxEnt on fixed grids (reference: set_G_tau_data again, elementwise_maxent.py:373-395, then run()).
 * MXE_ERR_STATE when no elements are set, MXE_ERR_ARG when n_elem differs. */
int  mxe_elements_update_data(mxe_ctx* ctx, int n_elem, const double* G, const int64_t* G_offset);

/* ---- the hot path ------------------------------------------------------ */
/* n_chain warm-started alpha scans of n_alpha values each.
 *   elem_of_chain[c]          element index
 *   alpha_scaled[c*n_alpha+i] alpha * scale_alpha (maxent_loop.py:216-243),
 *                             in the order they are to be visited
 *   v0[c*n_s + k]             start vector in the caller's singular basis
 *                             (maxent_loop.py:196-203)
 * Outputs (host pointers, any may be NULL), problem p = c*n_alpha + i:
 *   out_v[p*n_s+k]  optimum in the caller's singular basis
 *   out_H[p*n_omega+j] hidden image H(v)   out_chi2/out_S/out_Q[p]
 *   out_niter[p] iterations (minimizer.n_iter_last)   out_converged[p] 0/1
 *   (an alpha whose minimisation FAILED -- damping exhausted, nothing finite to evaluate: converged 0
 *    before maxiter -- reports v, chi2, S, Q of its last accepted iterate and H = NaN)
 *   out_nevals[p] cost evaluation passes spent on p
 * Blocking: returns after the results are in the host buffers. */
int  mxe_solve_chains(mxe_ctx* ctx, int n_chain, int n_alpha,
                      const int32_t* elem_of_chain, const double* alpha_scaled,
                      const double* v0, const mxe_opts* opts,
                      double* out_v, double* out_H, double* out_chi2,
                      double* out_S, double* out_Q, int32_t* out_niter,
                      int32_t* out_converged, int32_t* out_nevals);

/* Device-resident variant used by bench.py and by multi-GPU drivers:
 * mxe_chains_upload stages the chain description once; mxe_chains_launch
 * enqueues one pass of the solver on the ctx stream (inputs already in HBM),
 * bracketed by HIP events; mxe_sync waits for it; mxe_chains_fetch copies the
 * results to host buffers (same layout as mxe_solve_chains).
 * mxe_result_device_ptrs exposes the device result buffers (for an RCCL
 * gather driven by the caller); layouts as above except v, which is in the
 * whitened basis with row stride mxe_ns_padded() (use mxe_chains_fetch for
 * caller-basis v).  H, chi2, S, Q are contiguous in that order in ONE
 * allocation of P*n_omega + 3*P doubles starting at d_H, so that one
 * collective moves all per-alpha results. */
int  mxe_chains_upload(mxe_ctx* ctx, int n_chain, int n_alpha,
                       const int32_t* elem_of_chain, const double* alpha_scaled,
                       const double* v0, const mxe_opts* opts);
int  mxe_chains_launch(mxe_ctx* ctx);
int  mxe_sync(mxe_ctx* ctx);
/* Blocking.  After a launch in the lock-step layout: the alphas that did not converge there are solved again in
 * the one-chain layout, from the state they were left in, and their records are overwritten (iteration and
 * evaluation counts: both passes).  The lock-step kernel forms its Gram matrices from binary16 products (21 bits):
 * an inexact Newton matrix that is harmless where the system is well conditioned and stalls where it is not (few
 * data points, small alpha); it gives an alpha up after 32 iterations, and the one-chain kernel (binary64 Gram
 * matrix) takes it from there with the rest of mxe_opts.maxiter.  n_resolved (may be NULL): how many alphas that
 * was.  Alphas that couple more than the 32 directions the lock-step kernel has a build for (error bars far below
 * 1e-5 of the data: the smallest alphas of a scan) are not in its pieces at all when they are the smaller part of the
 * batch: after mxe_chains_launch their records read NaN / not converged / 0 iterations, and this call solves them
 * as one warm chain per scan from the last alpha before them -- a launch is complete after mxe_chains_finish, not
 * before.  A no-op after a launch in the one-chain layout or when everything converged.  mxe_solve_chains calls it;
 * what it replaces in the reference is nothing but the remaining iterations of levenberg_minimizer.py:155-243. */
int  mxe_chains_finish(mxe_ctx* ctx, int32_t* n_resolved);
int  mxe_chains_fetch(mxe_ctx* ctx, double* out_v, double* out_H,
                      double* out_chi2, double* out_S, double* out_Q,
                      int32_t* out_niter, int32_t* out_converged,
                      int32_t* out_nevals);
/* The two blocks of per-alpha scalars queued behind the launch on the context's stream, so that a caller with jobs in flight on
 * several contexts finds them in its memory when the kernel has finished instead of copying them out, job after job, when all
 * kernels are done: out_chi2_S_Q [3][P] (chi2 | S | Q), out_niter_converged_nevals [3][P].  Both should be page-locked
 * (mxe_host_alloc), or the runtime stages them.  mxe_chains_fetch given exactly these destinations waits for the stream and
 * copies nothing; mxe_chains_finish, when it solves anything again, makes the next fetch copy afresh.  The destinations must
 * stay allocated until a call that waits for the stream has returned.  Replaces nothing of the reference (its per-alpha
 * scalars are Python floats as they are made, maxent_loop.py:330-353): plumbing of the asynchronous boundary. */
int  mxe_chains_prefetch(mxe_ctx* ctx, double* out_chi2_S_Q, int32_t* out_niter_converged_nevals);
/* log det(I + M W / alpha~) at the returned point of every problem of the last launch,
 * [n_chain][n_alpha], over all n_s kept singular directions (M = S U^T diag(1/err^2) U S,
 * W = V^T diag(w) V).  It is the only expensive term of NormalLogProbability
 * (probabilities.py:60-85: two n_omega x n_omega slogdet per alpha in the reference):
 * log p = -1/2 logdet - Q - log(alpha~) with the default norm and prior. */
int  mxe_logdet(mxe_ctx* ctx, double* out_logdet);
/* diagnostic: size of the coupled (active) block of the last Newton iteration of
 * every problem, [n_chain][n_alpha] (see mxe_opts.decouple_tol) */
int  mxe_chains_fetch_nact(mxe_ctx* ctx, int32_t* out_nact);
int  mxe_result_device_ptrs(mxe_ctx* ctx, void** d_H, void** d_chi2,
                            void** d_S, void** d_Q, void** d_v,
                            void** d_niter, void** d_converged);
int  mxe_ns_padded(mxe_ctx* ctx);
/* two result allocations (0, 1): the next launches / fetches / pointer queries use
 * `which`, so that a driver can gather buffer k while the solver fills buffer 1-k */
int  mxe_set_result_buffer(mxe_ctx* ctx, int which);
/* duration of the last mxe_chains_launch in ms (HIP events on the ctx stream) */
int  mxe_last_kernel_ms(mxe_ctx* ctx, float* ms);
/* mxe_timing_mark records an event on the ctx stream; mxe_ms_since_mark waits for the end of the
 * last launch and returns the device time from the mark to it: the duration of a run of launches
 * that were enqueued back to back, without a host synchronisation in between */
int  mxe_timing_mark(mxe_ctx* ctx);
int  mxe_ms_since_mark(mxe_ctx* ctx, float* ms);
/* the HIP stream (hipStream_t) the ctx launches on, for callers that order their own device work
 * (an RCCL gather of the result buffers) against it with events instead of host synchronisation */
void* mxe_stream(mxe_ctx* ctx);
/* name of the kernel instantiation the last mxe_chains_launch ran, as a profiler shows it
 * (e.g. "mxe::chain_kernel_mc<32, 4>"); owned by the ctx */
const char* mxe_last_kernel_name(mxe_ctx* ctx);
/* kernel geometry of the last launch: waves per chain, workgroups, LDS bytes */
int  mxe_last_launch_info(mxe_ctx* ctx, int* waves_per_chain, int* n_workgroups,
                          int* lds_bytes);
/* Depth of the last lock-step launch in ROUNDS (one Newton iteration of the four chain slots of a workgroup): maximum and mean over
 * its workgroups.  The reference's scan is one serial chain of n_alpha x ~16 iterations (maxent_loop.py:241-245 around
 * levenberg_minimizer.py:155); here a launch that does not fill the GPU is as long as its deepest workgroup, and this is that depth
 * as the kernel counted it (bench.py: scaling_projection.bound divides the measured time by it).  [0]: the launch, or the binary32
 * first pass of a two-pass launch (mxe_opts.lds_basis); [1]: the second pass (0 when there is none).  One-chain layout: zeros.
 * Blocking. */
int  mxe_launch_depth(mxe_ctx* ctx, int32_t* max_rounds /*[2]*/, double* mean_rounds /*[2]*/);
/* The schedule of the staged chains (after mxe_chains_upload): n_solo = workgroups of a two-per-CU lock-step launch that get a CU to
 * themselves for the longest pieces (the tails of the normal-entropy scans; the serial alpha loop they replace: maxent_loop.py:241-245).
 * That schedule rests on which workgroups share a CU -- b and b + gridDim / 2 --, an observed placement that HIP does not promise: the
 * library probes it once per device (a 25 us kernel that records XCC_ID / HW_ID) and drops the solo workgroups where it does not
 * hold.  placement_rule: 0 = the upload wanted none, 1 = probed and holds, 2 = does not hold (n_solo = 0).  Results never depend
 * on it; MXE_FORCE_NO_SOLO_RULE in the environment forces 2. */
int  mxe_schedule_info(mxe_ctx* ctx, int* n_solo, int* placement_rule);

/* ---- the cost function and its derivatives at caller-supplied points --- */
/* Device side of CostFunction.__call__ / f / d / dd (cost_function.py:73-85),
 * MaxEntCostFunction.f/dH/d/ddH/dd (maxent_cost_function.py:68-165), BryanCostFunction.f/d/dd
 * (bryan_cost_function.py:57-128) and of the component functions NormalChi2, NormalEntropy /
 * PlusMinusEntropy, NormalH_of_v / PlusMinusH_of_v (functions.py:336-796), for a batch of P points.
 *   elem_of_problem[p]  element (its G, err, D, entropy) the point belongs to
 *   alpha_scaled[p]     alpha * scale_alpha (>= 0)
 *   x                   input_is_H = 0: v, [P][n_s], the caller's singular basis
 *                       input_is_H = 1: the hidden image H itself, [P][n_omega] (the component functions
 *                       chi2(H), S(H) and H_of_v.inv: u = log(H/D) | log((H + sqrt(H^2 + 4 D^2)) / 2D))
 *   chi2_factor         eta in Q = eta chi2 / 2 - alpha S
 * Outputs (host, any may be NULL):
 *   out_Q, out_chi2, out_S [P];  out_H, out_u (= V v), out_w (= dH/du), out_q (= V g = dQ/dH) [P][n_omega];
 *   out_h = V^T H, out_g = eta (M h - b) + alpha v  [P][n_s]     (input_is_H: no alpha v term)
 *   out_W  = V^T diag(w) V,  out_W2 = V^T diag((V g) H) V        [P][n_s][n_s]
 * from which every derivative of the reference follows with M = S U^T diag(1/err^2) U S:
 *   default (dA_projection = 2):  d = W g        dd = W M' W + alpha W       (M' = eta M)
 *   dA_projection = 1:            d = g          dd = M' W + alpha I
 *   dA_projection = 0:            d = V g        dd = V M' W + alpha V       (n_omega rows)
 *   d_dv = True:                  d = W g        dd = W M' W + alpha W + W2
 *   BryanCostFunction:            d = g          dd = M' W
 *   dQ/dH = V g;  dchi2/dH = 2 V (M h - b);  dS/dH = -u;  d2S/dH2 = -diag(1/w);  dH/dv = diag(w) V */
int  mxe_eval_batch(mxe_ctx* ctx, int P, const int32_t* elem_of_problem, const double* alpha_scaled,
                    const double* x, int input_is_H, double chi2_factor,
                    double* out_Q, double* out_chi2, double* out_S,
                    double* out_H, double* out_u, double* out_w, double* out_q,
                    double* out_h, double* out_g, double* out_W, double* out_W2);
/* NormalEntropy / PlusMinusEntropy as functions of a hidden image given directly (functions.py:508-520,
 * 544-564): S, dS/dH and the diagonal of d2S/dH2 for P images H [P][n_omega] and one default model D
 * [n_omega] (including delta-omega).  No context needed; outputs may be NULL. */
int  mxe_entropy(int device, int kind, int n_omega, int P, const double* H, const double* D,
                 double* out_S, double* out_dS, double* out_ddS);
/* Audit of the last launch, every problem, binary64, all n_s directions (no active-subspace cut, no
 * binary16 tiles): at the returned v the exact Newton correction delta of Bryan's system is computed and
 *   out_corr[p] = ||w * V delta||_2 / ||H||_2   -- to first order the relative L2 distance of the
 *                 returned H from the minimiser (the quantity mxe_opts.tol_h bounds by an estimate)
 *   out_gmax[p] = max_k |g_k| / (|c_k rho_k| + |alpha v_k|)  -- the gradient against its cancelling terms
 * [n_chain][n_alpha]; either may be NULL. */
int  mxe_audit(mxe_ctx* ctx, double* out_corr, double* out_gmax);

/* ---- the default analyzer's alpha on the device; selected rows ---------- */
/* LineFitAnalyzer (analyzers/linefit_analyzer.py:28-87,151-183: the kink of log chi2 over log alpha,
 * linefit_deg = p2_deg) for every scan of the last launch, enqueued on the ctx stream behind it; the H
 * of the chosen alpha is copied next to chi2 / S / Q, which makes the COMPACT result pack
 *     chi2 [P] | S [P] | Q [P] | H_selected [n_chain][n_omega] | index [n_chain] (as doubles)
 * contiguous on the device (P = n_chain * n_alpha).  index = -1: no fit (fewer than five alphas, chi2 NaN). */
int  mxe_select_launch(mxe_ctx* ctx, int p2_deg);
int  mxe_select_fetch(mxe_ctx* ctx, int32_t* out_index, double* out_H_selected);
/* The three default analyzers of the reference in one launch (it also fills the compact pack like mxe_select_launch):
 *   0  LineFitAnalyzer (as above),
 *   1  Chi2CurvatureAnalyzer (analyzers/chi2_curvature_analyzer.py:25-49,101-131): the alpha of the largest curvature
 *      y'' / (1 + y'^2)^(3/2) of y = log10 chi2 over x = gamma log10 alpha, NaN ignored, the first of equal maxima,
 *   2  EntropyAnalyzer (analyzers/entropy_analyzer.py:72-103): the alpha where (dS / dlog alpha)^2 is smallest;
 * mxe_select3_fetch: out_index [3][n_chain] (-1: nothing to choose from), out_H_selected [3][n_chain][n_omega] (may be
 * NULL), both in ONE device-to-host copy. */
int  mxe_select3_launch(mxe_ctx* ctx, int p2_deg, double gamma);
int  mxe_select3_fetch(mxe_ctx* ctx, int32_t* out_index, double* out_H_selected);
/* The same, the rows of the analyzers first .. first + count - 1 only (out_H_selected [count][n_chain][n_omega]; out_index
 * [3][n_chain] or NULL): a caller whose default analyzer is the line fit takes its row behind the launch and those of the
 * other two when somebody looks at them (result.analyzer_results[...]['A_out'], maxent_result.py:600-640 of the reference
 * reads one analyzer's A_out at a time).  The rows stay valid until the next mxe_select3_launch / mxe_chains_upload. */
int  mxe_select3_fetch_rows(mxe_ctx* ctx, int32_t* out_index, int first, int count, double* out_H_selected);
/* The copies of mxe_select3_fetch_rows queued behind the selection kernel (see mxe_chains_prefetch): the indices into memory of
 * the context, the rows into out_H_selected (page-locked; to stay allocated until a call that waits for the stream has
 * returned).  mxe_select3_fetch_rows with the same first / count / out_H_selected then waits and converts the indices. */
int  mxe_select3_prefetch_rows(mxe_ctx* ctx, int first, int count, double* out_H_selected);
/* n_rows hidden images of the last launch by problem index (chain * n_alpha + i), [n_rows][n_omega]:
 * what an analyzer needs (one row per scan) without moving all of H */
int  mxe_fetch_rows(mxe_ctx* ctx, int n_rows, const int32_t* problem_index, double* out_H);

/* ---- several GPUs: shard by matrix element, one gather ------ */
/* The (element, alpha) problems are independent given U, S, V: element e goes to rank e mod n_ranks
 * (mxe_shard_plan), every rank stages the basis itself and solves its shard with the same entry points;
 * afterwards ONE gather brings the result packs to the root's device (and, if asked, to its host):
 * RCCL send / recv over xGMI, called directly from this library (librccl.so.1 is loaded on first use).
 *   ranks in separate processes:  rank 0 calls mxe_comm_unique_id and hands the 128 bytes to the others
 *                                 (file, socket, environment: the caller's business); every rank calls
 *                                 mxe_comm_init(ctx, n_ranks, rank, id) and, per step, mxe_gather.
 *   ranks in one process:         mxe_comm_init_local(ctxs, n) (rank = position; contexts on the same
 *                                 device -- used for tests on one GPU -- gather with device copies
 *                                 instead of RCCL) and, per step, mxe_gather_local.
 * what: MXE_GATHER_COMPACT (the pack above; needs mxe_select_launch first) or MXE_GATHER_FULL (all H
 * in front of it).  counts[r]: doubles rank r contributes (checked against the rank's own launch).
 * The gather is enqueued on the ctx streams; with recv_host != NULL the root copies the gathered packs,
 * rank after rank, to the host and waits for it.  recv_host may be NULL on the other ranks. */
#define MXE_GATHER_COMPACT 0
#define MXE_GATHER_FULL    1
int  mxe_shard_plan(int n_elem, int n_ranks, int32_t* rank_of_elem, int32_t* local_index, int32_t* n_local);
int  mxe_comm_unique_id(char* out_id128);
int  mxe_comm_init(mxe_ctx* ctx, int n_ranks, int rank, const char* id128);
int  mxe_comm_init_local(mxe_ctx** ctxs, int n);
int  mxe_comm_destroy(mxe_ctx* ctx);
/* test plumbing for a one-GPU box (communicators of mxe_comm_init): with on != 0 the root's own pack goes through
   ncclGroupStart / ncclSend / ncclRecv (to itself) / ncclGroupEnd instead of a device copy, and mxe_comm_allreduce
   runs ncclAllReduce with the single rank -- the RCCL calls of the multi-GPU path execute on one device */
int  mxe_comm_set_loopback(mxe_ctx* ctx, int on);
int  mxe_gather(mxe_ctx* ctx, int root, int what, const int64_t* counts, double* recv_host);
/* sum (op = 0) or maximum (op = 1) of n <= 64 doubles over the ranks, result on every rank; with it the
 * ranks of separate processes agree on a timing or wait for each other (a barrier is a sum of zeros) */
int  mxe_comm_allreduce(mxe_ctx* ctx, double* inout_host, int n, int op);
int  mxe_gather_local(mxe_ctx** ctxs, int n, int root, int what, const int64_t* counts, double* recv_host);

/* ---- output map A = B H (PreblurA_of_H.f, functions.py:999-1001) ------- */
/* B: n_omega x n_omega row-major.  Applies to the device-resident H of the
 * last solve; result n_problem x n_omega to host. */
int  mxe_apply_output_map(mxe_ctx* ctx, const double* B, double* out_A);

/* ---- kernel matrix staging on the device ------------- */
/* TauKernel._fill_values (kernels.py:244-271), get_preblur (preblur.py:31-58),
 * PreblurKernel._fill_values (kernels.py:384-393: K' = K diag(delta) B), KernelSVD.svd
 * and reduce_singular_space (kernels.py:53-122) for a batch of n_b blur widths in one
 * go (a b-scan; preblur_b[ib] <= 0: the plain TauKernel).  tau: n_tau, omega / delta:
 * n_omega (delta = the trapezoid weights of the mesh, omega_meshes.py:54-62).
 * The decomposition is a pivoted-QR preconditioned one-sided Jacobi SVD in binary64
 * (one workgroup per blur width); singular values S >= threshold (ABSOLUTE, as in the
 * reference) are kept, at most ns_max (<= 128).  Outputs, host, row-major, per item ib:
 *   out_K [ib][n_tau][n_omega]   the (blurred) kernel matrix; may be NULL
 *   out_U [ib][n_tau][ns_max], out_S [ib][ns_max], out_V [ib][n_omega][ns_max]
 *         (columns >= out_ns[ib] are zero), out_ns [ib]
 *   out_info [ib][3]: rank kept by the QR stage, Jacobi sweeps, status (may be NULL)
 *   out_ms: device time of the whole batch (may be NULL)
 * Returns MXE_ERR_LIMIT if more than ns_max singular values pass the threshold,
 * MXE_ERR_NUMERIC if the Jacobi sweeps did not converge. */
int  mxe_kernel_svd(int device, int n_tau, int n_omega, const double* tau,
                    const double* omega, const double* delta, double beta,
                    int n_b, const double* preblur_b, double threshold, int ns_max,
                    double* out_K, double* out_U, double* out_S, double* out_V,
                    int32_t* out_ns, int32_t* out_info, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* MAXENT_HIP_H */
