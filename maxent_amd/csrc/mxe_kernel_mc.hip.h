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
#ifndef MXE_X_KERNARG_RELOAD
#define MXE_X_KERNARG_RELOAD 0       // (measured, not taken: see chain_kernel_mc)
#endif
#include "mxe_kernel.hip.h"

namespace mxe {

constexpr int MCC = 4;            // chain slots per workgroup == wavefronts per workgroup

struct MCExtra {
    const int* wg_chains;         // static layout: [n_wg][MCC] chain ids, -1 = empty slot (dynamic layout: see below)
    int n_wg;
    const int* queue;             // dynamic layout: chain ids in the order they are handed out
    int n_queue;                  //   (0 = static layout)
    int* counter;                 //   next queue position (zeroed before every launch)
    int stagger;                  // WGPC = 2: the second half of the grid starts this many units of 4096 cycles late
    int n_solo;                   // WGPC = 2, dynamic layout, gridDim = 2 x CUs: workgroups b and b + gridDim / 2 share a CU
                                  //   (tools/wg_placement.hip); the first n_solo workgroups get their CU to themselves --
                                  //   their partners leave at once -- and the first pieces of the queue (the most expensive)
    double* gstate;               // GSTATE builds: the omega-space state of the workgroups, gstate_stride doubles each
    size_t gstate_stride;         //   (mc_gstate_doubles)
    // (dynamic layout, n_queue > 0: wg_chains is not a table of chain ids but, when not null, [n_wg] ints that RECEIVE the rounds --
    //  Newton iterations of its four slots in lock-step -- every workgroup ran: the depth of a launch that does not fill the GPU,
    //  mxe_launch_depth.  A field of its own changed the kernel-argument layout and cost the <32, 2> build 0.25 %.)
};

// omega-space state of one workgroup of a GSTATE build, in doubles: u [nwp][4] | H [nwp][4] + look-ahead | sw [nwp][4] floats + look-ahead
constexpr int MC_GSTATE_PAD = 1024;
inline size_t mc_gstate_doubles(int nwp) { return (size_t)nwp * 4 + ((size_t)nwp * 4 + MC_GSTATE_PAD) + ((size_t)nwp * 4 + MC_GSTATE_PAD) / 2; }

// NA    capacity of the active block: 32 or 48
// WGPC  workgroups per CU the kernel is built for:
//       1  u, H, sw of the four slots in LDS (any n_omega the 160 KB hold), 512 registers per lane;
//       2  (n_omega_pad <= 512) u lives in registers -- every lane keeps the eight (row, slot) elements it
//          updates in the row pass --, the partial h of the waves are summed with LDS atomics, the fused pass
//          keeps four instead of eight row groups of V in flight: 73 KB of LDS and 256 registers, so that
//          two workgroups share a CU and the serial sections of one (the solve of the home waves, the accept
//          step; vector instructions only) run beside the streaming passes of the other, which wait for L2.
// rows of V behind the last omega row (and LDS entries behind H / sw) that the look-ahead of the
// fused pass may read without using them: (DEPTH + 2) groups * 4 waves * 4 rows, rounded up
constexpr double MC_GRAM_ERR = 0.0;              // allowance for the inexact Gram tiles in the stopping estimate
constexpr int MC_INIT_STRIDE = 3 * 256 + 64 + 8;       // KParams::init_tab, NA = 32: three tile pairs, h, four scalars (+ pad)
constexpr int MC_LOOKAHEAD_ROWS = 512;
constexpr int MC_LOOKAHEAD_LDS = (8 + 2) * 4 * 4 * 4 + 64;       // entries

#ifndef MXE_X_DEPTH2
#define MXE_X_DEPTH2 2        // V ring of the fused pass at two workgroups per CU (4: 1.33 ms and 300 MB of spill stores per launch; 2: 1.29 ms, none)
#endif
// A led piece solves its leading alpha cold to 1e-3 and then walks down the mesh: every alpha on the way gets Newton
// rounds until the estimated next correction is below MXE_X_WALK_TOL, at most MXE_X_WALK_ITERS of them -- in practice
// ONE (a path follower with one corrector step).  Measured on the shards of the BASELINE batch (slowest rank of 8 /
// rank 0): 6e-2 ... 4e-1 all 0.526 / 0.483 ms; with the leading alpha itself only at the walk's tolerance 3e-2 was as
// fast and 6e-2 ran away on the scan with the hardest tail (2.4 ms): the walk needs a start ON the path.
#ifndef MXE_X_WALK_TOL
#define MXE_X_WALK_TOL 1e-1
#endif
#ifndef MXE_X_LEAD_TOL
#define MXE_X_LEAD_TOL 2e-2     // the leading alpha of a walking piece: a start ON the path.  Measured (r03, slowest rank of cfg4 / 8,
                                // cfg2, cfg3): 1e-3 0.492 / 0.434 / 0.465 ms, 1e-2 0.473 / 0.418 / 0.445, 3e-2 0.454 / 0.386 / 0.414, 6e-2 runs away
                                // on the scan with the hardest tail (rank 5 of cfg4 / 8: 1.6 ms, as in r02); 2e-2 keeps a factor 3 to that
#endif
// The walk of a led piece does not visit every alpha of the mesh: it lands where alpha has fallen by at most this factor since the
// last landing -- one corrector round per landing.  Measured safe over a factor 1.5 (r02: at most 11 evaluations for such a warm
// step; over a factor 2 single scans took 100-300).  On the BASELINE mesh that is every third alpha: cfg2 0.389 -> 0.344 ms,
// cfg3 0.411 -> 0.372 ms (profiles/r04_experiments.txt: fixed strides 2 and 3 on one box); 1.0 = every alpha, as until r04
#ifndef MXE_X_WALK_RATIO
#define MXE_X_WALK_RATIO 1.55
#endif
// ... but only from a landing that is ON the path: ONE corrector round that ended with an estimated next correction below this (the
// walk's own tolerance is MXE_X_WALK_TOL = 1e-1).  Without the condition the scan with the hardest tail of the BASELINE batch
// (element 221: rank 5 of cfg4 / 8) ran away at every stride > 1: 0.39 -> 0.91 ms for that rank; with it (1e-2: never taken, 3e-2
// and 1e-1: the same) that rank stays at 0.40 ms and cfg2 / cfg3 go 0.389 / 0.415 -> 0.359 / 0.379 ms (profiles/r04_experiments.txt)
#ifndef MXE_X_WALK_SKIP_TOL
#define MXE_X_WALK_SKIP_TOL 3e-2
#endif
#ifndef MXE_X_WALK_ITERS
#define MXE_X_WALK_ITERS 4
#endif
#ifndef MXE_X_DEPTH1
#define MXE_X_DEPTH1 4        // V ring of the fused pass at one workgroup per CU (8: the shards of an 8-GPU job 0.464 / 0.508 ms, 4: 0.455 / 0.497, 2: 0.463 / 0.508)
#endif
#ifndef MXE_X_ILL_ITERS
#define MXE_X_ILL_ITERS 8      // iterations an alpha gets after its solve first met a small pivot, before it is handed to the one-chain kernel
#endif
// Wave priorities (s_setprio).  Two workgroups share a CU and run out of phase, so a SIMD holds one wave of each: a home wave
// in its serial section (dependent chains: every instruction waits for the one before) beside a wave of the partner that
// streams -- and the arbiter, left alone, lets the streaming wave's instructions (a 16-cycle MFMA, a burst of loads) go first
// as often as not.  The serial section is the critical path of a round: with priority 2 for it cfg4 0.863 -> 0.823 ms; the fused
// pass at 1 (its matrix instructions and the split that feeds them, over the partner's row pass) 0.823 -> 0.815; the row pass
// at 1 instead: 0.833 (worse); levels 3 / 2 / 0 the same as 2 / 1 / 0.  One workgroup per CU with eight waves (two per SIMD, all in
// the same phase): the waves still inside the loop of the fused pass go before those that have left it -- cfg2 0.393 -> 0.383,
// cfg3 0.421 -> 0.405, the shards of 4 / 8 GPUs 0.495 / 0.449 -> 0.481 / 0.438 ms (same box, twice).  A launch of two workgroups
// per CU that does NOT fill the GPU (one rank's shard of a two-GPU job: the <32, 2, lead> build) is bound by the depth of its
// chains, where the partner's delayed passes are on the critical path too: 0.554 -> 0.579 ms with them -- not there
// (profiles/r03_h_experiments.txt).
#ifndef MXE_X_HOME_PRIO
#define MXE_X_HOME_PRIO 2
#endif
#ifndef MXE_X_FUSED_PRIO
#define MXE_X_FUSED_PRIO 1
#endif
#ifndef MXE_X_SLOW_STEPS
#define MXE_X_SLOW_STEPS 0     // full Newton steps in a row that do not halve the correction before an alpha is handed over (0: never).
                               // Built for VERDICT r02 item 7 and measured with 3: an iteration of this kernel costs 1 / 25 of one of the one-chain
                               // kernel the alpha is handed to, so crawling to the limit of 32 is the cheaper way -- stress set 2401 -> 6953
                               // alphas finished there, solve wall in sum 1466 -> 1595 ms; sigma = 2e-6 batch 11.0 -> 13.0 ms (profiles/r03_e_na64.txt)
#endif
#ifndef MXE_X_RD1
#define MXE_X_RD1 4        // (8: the shards of an 8-GPU job 0.593 / 0.653 -> 0.617 / 0.688 ms)
#endif
#ifndef MXE_X_UL
#define MXE_X_UL 0            // u elements per lane kept in LDS instead of registers (0: all eight in registers)
#endif
// LEAD  pieces may be led by an earlier alpha of their scan (KParams::chain_lead); a build of its own, because the
//       two extra instructions of start_piece cost the schedule without such pieces 1.6 % (register allocation)
// NWV   wavefronts per workgroup: 4 (wave q is the home of slot q and nothing else) or, for launches that do not fill
//       the GPU (WGPC = 1: single scans, small matrices, one rank's shard of a multi-GPU job), 8 -- waves 4 .. 7 are
//       helpers that take half of the rows of the two streaming passes.  Such a launch is as long as its deepest chain
//       of rounds, and at one wave per SIMD the passes wait for L2 and for the wave's own issue rate (a v_fma_f64 every
//       9 cycles from one wave, every 4.75 from two: tools/mfma_f64_rate.hip); a second wave per SIMD halves them.
// GSTATE  frequency meshes whose state does not fit the LDS (n_omega_pad > ~1500): u, H and sw of the four slots live in
//       device memory (MCExtra::gstate, one slice per workgroup; the L2 holds them).  Their access pattern is the one the
//       LDS arrays have -- [row][slot], 512 contiguous bytes per tile in the row pass, 128 per row group in the fused pass
//       -- and adds ~190 B per omega row and round to the 1024 B of V and V^T.  Waves of a workgroup share the CU's
//       L1, the workgroup barriers order the accesses (as in chain_kernel<.., GST>).
// the kernel's first argument, either as the compiler keeps it (RELOAD = false) or read through the kernarg segment pointer, which
// refresh() makes opaque (see chain_kernel_mc)
template <bool RELOAD> struct KernargRef;
template <> struct KernargRef<false> {
    const KParams& r;
    __device__ __forceinline__ explicit KernargRef(const KParams& a) : r(a) {}
    __device__ __forceinline__ const KParams& operator*() const { return r; }
    __device__ __forceinline__ void refresh() {}
};
template <> struct KernargRef<true> {
    typedef const KParams __attribute__((address_space(4))) K4;
    K4* q;
    __device__ __forceinline__ explicit KernargRef(const KParams&) : q((K4*)__builtin_amdgcn_kernarg_segment_ptr()) {}
    __device__ __forceinline__ K4& operator*() const { return *q; }
    __device__ __forceinline__ void refresh() { asm volatile("" : "+s"(q)); }
};

template <int NA, int WGPC, bool LEAD = false, int NWV = 4, bool GSTATE = false>
__global__ __launch_bounds__(64 * NWV, WGPC)
#if MXE_X_KERNARG_RELOAD
void chain_kernel_mc(const KParams p_arg, const MCExtra x)
#else
void chain_kernel_mc(const KParams p, const MCExtra x)
#endif
{
    // MXE_X_KERNARG_RELOAD (an experiment, off): at two workgroups per CU the kernel's arguments are read from the kernarg segment
    // where they are used (scalar loads, re-issued every round: the pointer is made opaque at the top of the loop) instead of being
    // held in scalar registers for the life of the kernel -- 175 of them are spilled to the lanes of three vector registers and come
    // back through 461 v_readlane (static; whole tuples of eight for one option), a third of the vector instructions of the accept
    // section.  Executed, that is 1.9 % of the kernel's vector instructions (SQ_INSTS_VALU 247.5 M -> 242.9 M per launch) and the
    // full batch gains 0-0.25 % (0.8139 -> 0.8118, 0.8108 -> 0.8107 ms; the two-GPU shard 0.553 -> 0.547): the issue slots it frees
    // are not what the kernel waits for.  A lone workgroup LOSES (it waits for the loads: cfg2 0.353 -> 0.359 ms).
    // profiles/r04_experiments.txt 9.
#if MXE_X_KERNARG_RELOAD
    KernargRef<WGPC == 2> kargs(p_arg);
#define p (*kargs)
#endif
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = 64 * NWV;
    static_assert(WGPC == 1 || WGPC == 2, "one or two workgroups per CU");
    static_assert(!GSTATE || WGPC == 1, "device-memory state: the one-workgroup-per-CU build");
    static_assert(NWV == 4 || (NWV == 8 && WGPC == 1), "helper waves only in the one-workgroup-per-CU build");
    constexpr bool UREG = (WGPC == 2);            // u in registers, h summed with atomics
    constexpr bool PRIO = (WGPC == 1) || !LEAD;   // wave priorities (MXE_X_HOME_PRIO): not in the two-workgroups-per-CU build of launches that do not fill the GPU, see there
    constexpr int NP = 64;
    constexpr int NT = NA / 16;                   // 16-column tiles of the Gram block
    constexpr int NPAIR = NT * (NT + 1) / 2;
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= x.n_wg) return;
    const int half_grid = (int)gridDim.x / 2;
    if (WGPC == 2 && x.n_solo > 0 && (int)blockIdx.x >= half_grid && (int)blockIdx.x < half_grid + x.n_solo) return;
    const int ns = p.n_s, nw = p.n_omega, nwp = p.n_omega_pad;
    const bool dynamic = x.n_queue > 0;

    // ---- LDS carve ----
    // Per-chain omega state [row][chain]: u (binary64), H (binary64) and sw = sqrt(w * sc2) as binary32
    // (sc2: a power of two per slot that puts the Gram operand sw * V into the binary16 range, see
    // step 3); w itself is not kept -- the row pass gets the old w back as sw^2 / sc2, good to
    // binary32, for the norm of the step that the stopping rule looks at.  The Gram tiles of the four
    // slots are kept in the accumulator layout of the MFMA ([slot][tile pair][register][lane]) as
    // binary64: the four waves add their binary32 partial tiles with ds_add_f64 -- ds_add_f32 takes
    // 770 cycles per instruction on gfx950 against 34 (tools/lds_atomic_rate.hip).
    double* vv   = lds;                          // [MCC][NP]   v
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
    // copies of the partial h: one per wave, or (UREG) one per PAIR of waves, summed with atomics.  Two additions onto zero give
    // a + b whichever comes first, and the two pair sums are added in a fixed order (the accept step): the h of a round does not
    // depend on the order in which the waves arrive.  With ONE copy for the four waves (rounds 2-4) it did, in the last bit, and a
    // launch of the full batch did not repeat bit for bit -- half of its chi2 moved by ~1e-10 from run to run.  (The Gram tiles
    // are sums of four binary32 values in binary64: exact, so in any order, unless the partials differ by more than 2^27.)
    constexpr int HPW = UREG ? 2 : NWV;
    static_assert(!UREG || NWV == 4, "the pair copies of h are laid out for four waves");
    double* hpart = vecI + NP * MCC;             // [HPW][MCC chains][NP]
    double* red  = hpart + HPW * MCC * NP;       // [NWV waves][32]
    double* ssc  = red + NWV * 32;               // [MCC][NP]   binary32 solve: power-of-two scale of row / column k, ~ 1 / sqrt(c_k^2 wmax + alpha)
    double* csc  = ssc + MCC * NP;               // [MCC][NP]   ... times c_k
    double* gs   = GSTATE ? x.gstate + (size_t)blockIdx.x * x.gstate_stride : nullptr;
    double* ui   = GSTATE ? gs : csc + MCC * NP;  // [nwp][MCC]   (WGPC = 1 only)
    // (the look-ahead of the fused pass reads up to MC_LOOKAHEAD_LDS entries past the end of Hi and of swF:
    //  they land in swF and Wt, are never used, and need no padding)
    // WGPC = 2: u of a lane's eight (row, slot) elements: the first 8 - UL in registers, the last UL in LDS
    // ([UL][256], one 8-byte slot per thread: what the 80 KB of a half CU leave room for)
    constexpr int UL = UREG ? MXE_X_UL : 0;
    double* Hi   = ui + (UREG ? (size_t)UL * T : (size_t)nwp * MCC);  // [nwp][MCC]
    float*  swF  = reinterpret_cast<float*>(Hi + (size_t)nwp * MCC + (GSTATE ? MC_GSTATE_PAD : 0));    // [nwp][MCC]
    double* Wt   = GSTATE ? csc + MCC * NP : reinterpret_cast<double*>(swF + (size_t)nwp * MCC);  // [MCC][NPAIR][4][64]
    static_assert(MCC * NPAIR * 256 * 2 >= MC_LOOKAHEAD_LDS, "the look-ahead stays inside the allocation");
    __shared__ int s_elem[MCC], s_kind[MCC], s_act[MCC], s_scr[MCC];

    // ---- slot state.  It is owned by the home wave, which loads it from LDS at the
    //      start of its two sections of a round and stores it back at their end, so
    //      that no register is pinned by it during the two streaming passes. ----
    struct Slot {
        double alpha, mu, chi2, S, Hn2, wmax, Q, Qprev, cperp, steplim, muh;    // muh: damping the last damped step of this piece needed
        double sc2;        // the power of two the sw in LDS (and the Gram tiles computed from it) carry
        double pred;       // what the stopping estimate of this alpha's previous Newton step predicted for the square of the
                           //   next correction (times |H|^2); 0: nothing predicted
        int elem, prob0, clen, ia, niter, nevals, nact, active, scratch, okprev, bt, capp;
        int lead;          // LEAD builds: alphas of the scan the piece walks through before its own first one (see start_piece)
        int wide;          // 1 + the iteration of this alpha at which the binary32 elimination first met a small pivot (0: never)
        double dHp;        // square of the correction of this alpha's previous step if that was a full Newton step (times |H|^2), else 0
        int slow;          // consecutive full Newton steps of this alpha that did not halve the correction
    };
    // the alphas of a slot's piece (a dependent global load in the accept step costs its full latency)
    constexpr int ACAP = UREG ? 32 : 128;
    __shared__ double s_alpha[MCC][ACAP];
    __shared__ double s_sd[MCC][14];
    __shared__ float s_zp[MCC][2][16];          // binary32 solve: z of the scaled system by column parity
    __shared__ __attribute__((aligned(16))) float s_zr[MCC][NA > 32 ? 64 : 4];     // ... of the one-row-per-lane solve (NA > 32)
    __shared__ double s_scw[MCC][2];             // row pass: 1 / sc2 of the sw it reads, sc2 of the sw it writes
    __shared__ int s_si[MCC][15];
    auto load_slot = [&](Slot& t) {
        const double* d = s_sd[wave]; const int* n = s_si[wave];
        t.alpha = d[0]; t.mu = d[1]; t.chi2 = d[2]; t.S = d[3]; t.Hn2 = d[4]; t.wmax = d[5];
        t.Q = d[6]; t.Qprev = d[7]; t.cperp = d[8]; t.steplim = d[9]; t.muh = d[10]; t.sc2 = d[11]; t.pred = d[12];
        t.elem = n[0]; t.prob0 = n[1]; t.clen = n[2]; t.ia = n[3]; t.niter = n[4]; t.nevals = n[5];
        t.nact = n[6]; t.active = n[7]; t.scratch = n[8]; t.okprev = n[9]; t.bt = n[10]; t.capp = n[11];
        t.lead = 0;
        if constexpr (LEAD) t.lead = n[12];
        t.wide = n[13]; t.slow = n[14]; t.dHp = d[13];
    };
    auto store_slot = [&](const Slot& t) {
        if (lane == 0) {
            double* d = s_sd[wave]; int* n = s_si[wave];
            d[0] = t.alpha; d[1] = t.mu; d[2] = t.chi2; d[3] = t.S; d[4] = t.Hn2; d[5] = t.wmax;
            d[6] = t.Q; d[7] = t.Qprev; d[8] = t.cperp; d[9] = t.steplim; d[10] = t.muh; d[11] = t.sc2; d[12] = t.pred;
            n[0] = t.elem; n[1] = t.prob0; n[2] = t.clen; n[3] = t.ia; n[4] = t.niter; n[5] = t.nevals;
            n[6] = t.nact; n[7] = t.active; n[8] = t.scratch; n[9] = t.okprev; n[10] = t.bt; n[11] = t.capp;
            if constexpr (LEAD) n[12] = t.lead;
            n[13] = t.wide; n[14] = t.slow; d[13] = t.dHp;
            s_act[wave] = t.active; s_scr[wave] = t.scratch;
        }
        wave_sync();
    };
    auto start_piece = [&](Slot& t, int c) {     // home wave: take chain (piece) c into this slot
        t.elem = p.chain_elem[c];
        t.cperp = p.cperp[t.elem];
        t.steplim = p.step_max * p.sumD[t.elem];
        t.prob0 = p.chain_prob0[c]; t.clen = p.chain_len[c];
        // a piece may be led by earlier (larger) alphas of its scan: chain_lead = how many entries before the piece's
        // first alpha the walk starts (0: none) -- at the last alpha above the range where a cold start from the default
        // model is expensive and unsafe.  The piece solves that alpha cold, walks down the mesh to its own first alpha
        // with a loose tolerance (alphas number -lead .. -1: starting points only, no records), and goes on as usual
        const int lead = LEAD ? p.chain_lead[c] : 0;
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
        ecor[wave * NP + lane] = 0.0; eacc[wave * NP + lane] = 0.0;
        if (lane == 0) { s_elem[wave] = t.elem; s_kind[wave] = p.elem_kind[t.elem]; }
        // Every piece of a class starts from the same vector: what its evaluation gives -- H, S, h, the Gram
        // tiles -- is tabulated, only rho = c h - ghat is the piece's own.  The slot is in the state "evaluated at v0"
        // right away (scratch = 2): its first round is a Newton step, not the evaluation of the start vector.
        if (NA == 32 && p.init_tab) {
            const int ic = p.chain_init[c];
            if (ic >= 0) {
                const double* T = p.init_tab + (size_t)ic * MC_INIT_STRIDE;
                double* Wq = Wt + (size_t)wave * NPAIR * 256;
                for (int i = lane; i < NPAIR * 256; i += 64) Wq[i] = T[i];
                const double r = (lane < ns) ? p.c[p.elem_ds[t.elem] * NP + lane] * T[NPAIR * 256 + lane] - gh[wave * NP + lane] : 0.0;
                rho[wave * NP + lane] = r;
                const double r2 = wave_sum(r * r);
                const double* sc = T + NPAIR * 256 + NP;
                t.S = sc[0]; t.Hn2 = sc[1]; t.wmax = sc[2]; t.sc2 = sc[3];
                t.chi2 = r2 + t.cperp;
                t.Q = 0.5 * t.chi2 - t.alpha * t.S;
                t.scratch = 2;
            }
        }
    };

    auto alpha_at = [&](const Slot& t, int i) -> double {       // alpha number i of the piece (i < 0: the walk before it)
        return (t.clen + t.lead <= ACAP) ? s_alpha[wave][i + t.lead] : p.alpha[(size_t)t.prob0 + i];
    };

    // ---- first pieces ----
    if (wave < MCC) {
        Slot t;
        int c;
        if (dynamic && WGPC == 2 && x.n_solo > 0) {
            // first pieces by position in the grid, so that the head of the queue lands in the workgroups that are
            // alone on their CU (the counter starts behind these entries)
            const int b = (int)blockIdx.x < half_grid ? (int)blockIdx.x : (int)blockIdx.x - x.n_solo;
            const int idx = b * MCC + wave;
            c = (idx < x.n_queue) ? x.queue[idx] : -1;
        } else if (dynamic) {
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
            t = Slot{1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0};
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
    const double* __restrict__ V  = p.Vx + (size_t)ds * nwp * NP;       // (columns interleaved for 16-byte loads, see KParams)
    const double* __restrict__ Vt = p.Vt + (size_t)ds * NP * nwp;
    if (wave == 0) { cc[lane] = p.c[ds * NP + lane]; ci[lane] = p.cinv[ds * NP + lane]; }
    __syncthreads();

#ifdef MXE_PROFILE
    // per-wave stamps (diagnostic build only): row = workgroup * 8 + wave
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t = clock64();
    long long prof_rounds = 0;
#if defined(MXE_PROFILE_ACCEPT)   // split the accept step: everything else goes to slot 5
#define MXE_STAMPW(idx) do { const long long t__ = clock64(); prof_acc[5] += t__ - prof_t; prof_t = t__; } while (0)
#define MXE_STAMPH(idx) do {} while (0)
#define MXE_STAMPA(idx) do { const long long t__ = clock64(); prof_acc[idx] += t__ - prof_t; prof_t = t__; } while (0)
#elif defined(MXE_PROFILE_HOME)  // split the home phase instead: everything else goes to slot 5
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
#ifndef MXE_STAMPA
#define MXE_STAMPA(idx) do {} while (0)
#endif

    // ------------------------------------------------------------------
    // home wave: the slot's Newton system  (c W c + a I) z = rhs  on the active block, in registers.
    // All 64 lanes work on the one N x N system (N <= 32): lane (h = lane >> 5, i = lane & 31) holds of
    // row i the columns of parity h -- A[kk] = column 2 kk + h, N / 2 doubles -- and the right-hand side b_i
    // (both halves carry it).  Gauss-Jordan elimination without pivoting (positive definite matrix), two pivots
    // per step (gj2_solve64, mxe_kernel.hip.h): every row subtracts (f0, f1) = (A_ij, A_i,j+1) P^-1 times rows j
    // and j + 1 -- P the 2 x 2 pivot block, inverted in closed form -- from its columns k > j + 1 and from b_i.
    //   * rows j, j + 1, as far as a lane needs them, sit in lanes of ITS OWN half: a broadcast inside the
    //     groups of 32 lanes, ds_swizzle_b32 (the LDS crossbar; no memory, static pattern);
    //   * columns j (even: lower half) and j + 1 (upper half) are the same register A[j / 2]: one pair of
    //     v_permlane32_swap hands both to every lane.
    // Per pair a wave issues N - j - 2 fused multiply-adds and 2 (N - j - 2) swizzles behind ONE dependent chain
    // (swap, three broadcasts, determinant, reciprocal, multipliers); the one-half layout of round 1 (lane = row,
    // v_readlane broadcasts) issued three times the vector instructions.  The pivot rows stay untouched and the
    // matrix ends block diagonal: z of a row pair is P^-1 (b_j, b_j+1), no back substitution.
    // The solve only preconditions the (inexact) Newton step; rows >= n_act are identity rows.
    // ------------------------------------------------------------------
    // Precision: the elimination runs in binary32 (gj2_solve64_f32: one dword per cross-lane move, full-rate multiply-
    // adds) on the system scaled to a diagonal of O(1) -- the Newton matrix is inexact at the 1e-6 level anyway (Gram
    // tiles from binary16 products, decoupling threshold) and the backward error of that solve, ~N eps |A|, is of the
    // same size: cfg4 0.915 -> 0.868 ms with the same iteration counts and audit.  Where it meets a small pivot (an
    // ill-conditioned block: few data points, tiny alpha) 24 bits slow the iteration down; such alphas are handed to the
    // one-chain kernel (binary64 throughout) early, like those the binary16 Gram products stall (mxe_chains_finish).
    // Measured and dropped: a binary64 body beside the binary32 one for those slots -- inlined it cost the kernel 3 %
    // through its register allocation even when it never ran (0.870 -> 0.899 ms), as a function of its own 5 % (0.919).
    auto gj_home = [&](auto NTag, double a, int n_act, double isc2, double& nrm_out, bool& small_pivot) -> bool {
        constexpr int N = decltype(NTag)::value;
        static_assert(N <= 32 && N % 2 == 0, "two half-waves of 32 rows");
        constexpr int NHALF = N / 2;
        const int q = wave;
        int ln = lane;
        // (opaque to the optimiser: the load addresses below depend on the lane only, and hoisted out of the
        //  round loop -- five instantiations of N -- they cost 150 registers for the whole kernel)
        asm volatile("" : "+v"(ln));
        const int i = ln & 31, h = ln >> 5;
        const double* Wq = Wt + (size_t)q * NPAIR * 256;
        const double* rq = rhs + q * NP;
        bool ok = true;
        const bool live = i < n_act;
        // rows and columns times s_k = 2^-e_k ~ 1 / sqrt(c_k^2 wmax + alpha) (W_kk <= wmax: V has orthonormal columns)
        const double* cq = csc + q * NP;
        const double si = ssc[q * NP + min(i, NP - 1)];
        const double ci_ = live ? cq[i] : 0.0;
        const double cis = ci_ * isc2;               // the tiles carry the factor sc2 of their operands
        float A[NHALF], A0[NHALF];                   // A0: c W c alone (scaled), for the norm of the step
        {
            // W is kept as the upper-triangular 16x16 tiles in the accumulator layout of the MFMA: entry
            // (r, c), r <= c, sits in tile pair (r >> 4, c >> 4) at register r & 3, lane 16 ((r & 15) >> 2) +
            // (c & 15).  The lane needs (min(i, k), max(i, k)) for k = 2 kk + h: offset = (lane part) + (static
            // part in kk) on either side of the diagonal; 2 kk is even, so the h of k adds without carry.
            // All loads are issued back to back (clamped row index, selected afterwards)
            const int ic = min(i, N - 1);
            const int imt = ic >> 4, iri = ic & 15;
            const int up_l = (imt * NT - imt * (imt - 1) / 2 - imt) * 256 + (iri & 3) * 64 + (iri >> 2) * 16 + h;   // row ic, column k >= ic
            const int lo_l = imt * 256 + iri + h * 64;                                                             // row k < ic, column ic
            double wr[NHALF], ck[NHALF];
#pragma unroll
            for (int kk = 0; kk < NHALF; ++kk) {
                const int k0 = 2 * kk;
                const int kmt = k0 >> 4, kri = k0 & 15;
                const int up_s = kmt * 256 + kri;
                const int lo_s = (kmt * NT - kmt * (kmt - 1) / 2 - kmt) * 256 + (kri & 3) * 64 + (kri >> 2) * 16;
                wr[kk] = Wq[(k0 + h >= ic) ? up_l + up_s : lo_l + lo_s];
                ck[kk] = cq[k0 + h];
            }
#pragma unroll
            for (int kk = 0; kk < NHALF; ++kk) {
                const int k = 2 * kk + h;
                double xv = (k < n_act) ? cis * wr[kk] * ck[kk] : 0.0;      // 0 in the rows and columns >= n_act
                A0[kk] = (float)xv;
                if (k == i) xv = live ? fma(a * si, si, xv) : 1.0;
                A[kk] = (float)xv;
            }
        }
        // two pivots per step: gj2_solve64 / gj2_solve64_f32 (mxe_kernel.hip.h)
        MXE_STAMPH(1);
        double z;
        small_pivot = false;
        {
            float zf;
            ok = gj2_solve64_f32<N>(A, live ? (float)(rq[i] * si) : 0.0f, i, zf, small_pivot);
            z = (double)zf * si;
            // delta^T W delta = z^T (c W c) z for Bryan's bound, as the quadratic form itself: z (rhs - a z) -- the
            // binary64 shortcut -- cancels to nothing where a dominates the matrix and z carries 24 bits.  Lane (h, i)
            // sums its columns (parity h) of row i, the halves are added, row i weighs in with z_i
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
        MXE_STAMPH(3);
        if (ok && live && h == 0) zz[q * NP + i] = z;
        MXE_STAMPH(4);
#ifdef MXE_PROFILE_HOME
        prof_acc[6] += 1;                        // solves (slot 6 is a count in this build)
#endif
        return ok;
    };

    // Round 5: the same system in the accumulator layout of v_mfma_f32_32x32x2_f32, eliminated by gjm_solve_f32 (mxe_kernel.hip.h):
    // one matrix instruction per pivot pair instead of ~56 swizzles and multiply-adds.  Lane (h, n) holds column n of the rows
    // 8 (v / 4) + 4 h + (v % 4), v = 0 .. 15; one instantiation for every size of the active block (steps beyond n_act are skipped).
#ifndef MXE_X_GJ_MFMA
#define MXE_X_GJ_MFMA 0       // (measured: the launch of the full batch 0.817 -> 0.852 ms with it, profiles/r05_experiments.txt 5.)
#endif
    auto gj_home_m = [&](double a, int n_act, double isc2, double& nrm_out, bool& small_pivot) -> bool {
        const int q = wave;
        int ln = lane;
        asm volatile("" : "+v"(ln));                 // (opaque: what depends on the lane only must not be hoisted out of the round loop)
        const int n = ln & 31, h = ln >> 5;
        const double* Wq = Wt + (size_t)q * NPAIR * 256;
        const double* rq = rhs + q * NP;
        const double* cq = csc + q * NP;
        const bool live = n < n_act;
        const double sn = ssc[q * NP + n];
        const double cn = live ? cq[n] * isc2 : 0.0;  // the tiles carry the factor sc2 of their operands
        floatx16 D, D0;
        {
            // W: upper-triangular 16 x 16 tiles in the accumulator layout of the 16 x 16 MFMA -- entry (r, c), r <= c, in tile pair
            // (r >> 4, c >> 4) at register r & 3, lane 16 ((r & 15) >> 2) + (c & 15).  All loads back to back, selected afterwards.
            double wr[16], cr[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 8 * (v / 4) + (v % 4) + 4 * h;
                const int lo = min(row, n), hi = max(row, n);
                const int mr = lo >> 4, mc = hi >> 4;
                const int off = (mr * NT - mr * (mr - 1) / 2 + (mc - mr)) * 256 + (lo & 3) * 64 + ((lo & 15) >> 2) * 16 + (hi & 15);
                wr[v] = Wq[off];
                cr[v] = cq[row];
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 8 * (v / 4) + (v % 4) + 4 * h;
                double xv = (row < n_act) ? cn * wr[v] * cr[v] : 0.0;      // 0 in the rows and columns >= n_act
                D0[v] = (float)xv;
                if (row == n) xv = live ? fma(a * sn, sn, xv) : 1.0;
                D[v] = (float)xv;
            }
        }
        MXE_STAMPH(1);
        float zf;
        small_pivot = false;
        const bool ok = gjm_solve_f32<32>(D, live ? (float)(rq[n] * sn) : 0.0f, n_act, zf, small_pivot);
        const double z = (double)zf * sn;
        // delta^T W delta = z^T (c W c) z for Bryan's bound, as the quadratic form itself (see gj_home): lane (h, n) sums its rows
        // of column n, the halves are added, column n (= row n: symmetric) weighs in with z_n
        float* zs = &s_zp[q][0][0];
        if (h == 0) zs[n] = live ? zf : 0.0f;
        wave_sync();
        float y = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 zr = *reinterpret_cast<const float4*>(zs + 8 * g + 4 * h);
            y = __builtin_fmaf(D0[4 * g + 0], zr.x, y); y = __builtin_fmaf(D0[4 * g + 1], zr.y, y);
            y = __builtin_fmaf(D0[4 * g + 2], zr.z, y); y = __builtin_fmaf(D0[4 * g + 3], zr.w, y);
        }
        {
            const unsigned yu = __builtin_bit_cast(unsigned, y);
            unsigned ya = yu, yb = yu;
            asm volatile("" : "+v"(ya), "+v"(yb));
            const auto ys = __builtin_amdgcn_permlane32_swap(ya, yb, false, false);
            const float yt = __builtin_bit_cast(float, (unsigned)ys[0]) + __builtin_bit_cast(float, (unsigned)ys[1]);
            nrm_out = wave_sum((h == 0 && live) ? (double)(zf * yt) : 0.0);
        }
        MXE_STAMPH(3);
        if (ok && live && h == 0) zz[q * NP + n] = z;
        MXE_STAMPH(4);
#ifdef MXE_PROFILE_HOME
        prof_acc[6] += 1;
#endif
        return ok;
    };

    // More than 32 coupled directions (NA = 48 / 64 builds): one row per lane, all N columns in registers, pivot rows by
    // v_readlane (gj1_solve_rows_f32).  Same scaling, same quadratic form for the norm of the step.
    auto gj_home_rows = [&](auto NTag, double a, int n_act, double isc2, double& nrm_out, bool& small_pivot) -> bool {
        constexpr int N = decltype(NTag)::value;
        static_assert(N <= 64 && N <= NA, "one row per lane");
        const int q = wave;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int i = ln;
        const double* Wq = Wt + (size_t)q * NPAIR * 256;
        const double* rq = rhs + q * NP;
        const double* cq = csc + q * NP;
        const bool live = i < n_act;
        const double si = ssc[q * NP + i];
        const double cis = (live ? cq[i] : 0.0) * isc2;
        float A[N], A0[N];
        {
            const int ic = min(i, N - 1);
            const int imt = ic >> 4, iri = ic & 15;
            const int up_l = (imt * NT - imt * (imt - 1) / 2 - imt) * 256 + (iri & 3) * 64 + (iri >> 2) * 16;    // row ic, column k >= ic
            const int lo_l = imt * 256 + iri;                                                                  // row k < ic, column ic
#pragma unroll
            for (int k0 = 0; k0 < N; k0 += 8) {
                double wr[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + e, kmt = k >> 4, kri = k & 15;
                    const int up_s = kmt * 256 + kri;
                    const int lo_s = (kmt * NT - kmt * (kmt - 1) / 2 - kmt) * 256 + (kri & 3) * 64 + (kri >> 2) * 16;
                    wr[e] = Wq[(k >= ic) ? up_l + up_s : lo_l + lo_s];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + e;
                    double xv = (k < n_act) ? cis * wr[e] * cq[k] : 0.0;
                    A0[k] = (float)xv;
                    if (k == i) xv = live ? fma(a * si, si, xv) : 1.0;
                    A[k] = (float)xv;
                }
            }
        }
        MXE_STAMPH(1);
        float zf;
        small_pivot = false;
        const bool ok = gj1_solve_rows_f32<N>(A, live ? (float)(rq[i] * si) : 0.0f, i, n_act, zf, small_pivot);
        const double z = (double)zf * si;
        s_zr[q][i & (NA > 32 ? 63 : 3)] = live ? zf : 0.0f;
        wave_sync();
        float y = 0.0f;
#pragma unroll
        for (int k0 = 0; k0 < N; k0 += 4) {
            const float4 z4 = *reinterpret_cast<const float4*>(&s_zr[q][(NA > 32) ? k0 : 0]);
            y = __builtin_fmaf(A0[k0], z4.x, y); y = __builtin_fmaf(A0[k0 + 1], z4.y, y);
            y = __builtin_fmaf(A0[k0 + 2], z4.z, y); y = __builtin_fmaf(A0[k0 + 3], z4.w, y);
        }
        nrm_out = wave_sum(live ? (double)(zf * y) : 0.0);
        MXE_STAMPH(3);
        if (ok && live) zz[q * NP + i] = z;
        MXE_STAMPH(4);
        return ok;
    };

    // WGPC = 2: u of the eight (row, slot) elements this lane updates in the row pass (n_omega_pad <= 512:
    // one batch of eight tiles per wave covers every row)
    double ureg[8 - UL];
#pragma unroll
    for (int tt = 0; tt < 8 - UL; ++tt) ureg[tt] = 0.0;
    // Two workgroups that share a CU run the same program from the same start: left alone they sit in the
    // same phase at the same time (solve beside solve, stream beside stream) and overlap nothing.  The
    // second half of the grid -- dispatched onto the CUs the first half already occupies -- starts half a
    // round late.
    if (WGPC == 2 && (int)blockIdx.x >= (int)(gridDim.x + 1) / 2)
        for (int sl = 0; sl < x.stagger; ++sl) __builtin_amdgcn_s_sleep(64);
    long long guard = 0;
    const long long guard_max = (long long)(dynamic ? x.n_queue : 1) * p.n_alpha * (p.maxiter + 64) + 64;

    bool first_round = true;
    while (guard++ < guard_max) {
#if MXE_X_KERNARG_RELOAD
        kargs.refresh();
#endif
        // (a slot that finishes its piece takes the next one from the queue right away, in step 4)

        // ---- 1. home wave: right-hand side, active block, factorise, solve, step ----
        if (wave < MCC) {
#if MXE_X_HOME_PRIO > 0
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(MXE_X_HOME_PRIO);     // the serial section: its dependent chains go first on a SIMD that a streaming wave shares
#endif
            const int q = wave, k = lane;
            Slot t;
            // ---- 4 (of the round before). rho, sums, accept / converge / advance, results ----
            // Accept and the solve that follows it are ONE serial section of the home wave, in one block: the slot state
            // goes from one to the other in registers (loaded from LDS behind the passes, stored at the end of the solve;
            // r02 stored it at the end of the accept step and loaded it again two lines later: ~1 k cycles per round)
            if (!first_round) {
                MXE_STAMPA(5);
                double h = 0.0;
    #pragma unroll
                for (int wv = 0; wv < HPW; ++wv) h += hpart[(wv * MCC + q) * NP + k];
                const double r = (k < ns) ? cc[k] * h - gh[q * NP + k] : 0.0;
                const double r2 = wave_sum(r * r);
                double sS = 0.0, sdH = 0.0, sHn = 0.0, swm = 0.0, sdu = 0.0;
    #pragma unroll
                for (int wv = 0; wv < NWV; ++wv) {
                    sS += red[wv * 32 + q * 8 + 0]; sdH += red[wv * 32 + q * 8 + 1];
                    sHn += red[wv * 32 + q * 8 + 2]; swm = fmax(swm, red[wv * 32 + q * 8 + 3]);
                    sdu = fmax(sdu, red[wv * 32 + q * 8 + 4]);
                }
                MXE_STAMPA(0);
                load_slot(t);
                MXE_STAMPA(1);
                if (t.active) {
                    rho[q * NP + k] = r;
                    const double chi2t = r2 + t.cperp, St = sS;
                    const double Qt = 0.5 * chi2t - t.alpha * St;
                    const bool finite = fabs(Qt) <= 1.7e308;
                    bool finish_alpha = false, failed = false; int conv = 0;
                    bool on_path = false;      // (a landing of the walk that ended well inside its tolerance: the next step may be long)
                    // (fresh: the piece's start state came from the table and this round evaluated its first Newton step
                    //  v0 - delta from scratch: a trial point like any other, except that the row pass has no old state to
                    //  measure the step against -- no convergence test on it)
                    const bool fresh = t.scratch == 2;
                    if (t.scratch == 1 || (fresh && !t.okprev)) {
                        // state restored from v (or first evaluation of the piece); damping kept
                        ++t.nevals;
                        if (finite) { t.scratch = 0; t.chi2 = chi2t; t.S = St; t.Hn2 = sHn; t.wmax = swm; t.Q = Qt; }
                        else { finish_alpha = true; failed = true; }   // cannot even evaluate: give up on this alpha
                    } else if (!t.okprev) {
                        finish_alpha = true; failed = true;         // the damping loop ran out of range
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
                            // (out of range -- or out of evaluations: the limit of 32 counts ACCEPTED steps, and an alpha whose every
                            //  step takes a dozen dampings held its workgroup, and with it the launch, for 470 rounds: 5.2 ms for a
                            //  batch of 0.5 ms; profiles/r05_experiments.txt 13.)
                            if (!(t.mu <= p.mu_max * t.alpha) || t.nevals >= p.mc_maxevals) { finish_alpha = true; failed = true; }
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
                        const bool estimated = p.stop_estimate && t.mu == 0.0 && t.okprev == 1 && sdu <= 1.0;
                        if (estimated) {
                            const double em1 = sdu * fma(sdu, fma(sdu, fma(sdu, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0);
                            const double fac = em1 + p.theta + MC_GRAM_ERR;
                            fac2 = fmin(1.0, fac * fac);
                        }
                        // The estimate assumes a Newton matrix that is exact up to theta.  The matrix this kernel solves with
                        // is not (Gram tiles from binary16 products, binary32 elimination): harmless where the system is well
                        // conditioned, but an ill-conditioned one contracts slower than predicted.  The estimate therefore
                        // checks itself: what it predicted at the previous step of this alpha for the correction just taken
                        // (t.pred) against that correction (sdH); an optimistic prediction inflates the present one by the
                        // same factor
                        double relH2_min = fac2 * sdH;                  // min(relH, relH_next)^2 * Hn2
                        if (t.mu > 0.0) {                               // (a damped step: the bound on the undamped one, see chain_kernel; rare: one division)
                            const double ud = 1.0 + t.mu / t.alpha;
                            relH2_min *= ud * ud;
                        }
                        const double pred_here = estimated ? relH2_min : 0.0;
    #ifndef MXE_X_NO_PRED_CHECK
                        if (estimated && t.pred > 0.0 && sdH > t.pred) relH2_min = fmin(sdH, relH2_min * (sdH / t.pred));     // (rare: one division)
    #endif
                        // (a leading alpha is only a starting point for the piece's first alpha: 1e-3 is enough)
                        const double tol_here = (LEAD && t.ia < 0) ? fmax(p.tol_h, (t.ia == -t.lead) ? MXE_X_LEAD_TOL : MXE_X_WALK_TOL) : p.tol_h;
                        const double tol2Hn = tol_here * tol_here * t.Hn2;
                        vv[q * NP + k] -= dlc[q * NP + k];
                        if (t.niter == 0) t.capp = (t.okprev == 5) ? 2 : (t.okprev == 1 && t.mu == 0.0) ? 1 : 0;
                        else eacc[q * NP + k] -= dlc[q * NP + k];     // what the later iterations add to the first iterate
                        t.chi2 = chi2t; t.S = St; t.Hn2 = sHn; t.wmax = swm;
                        t.Qprev = t.Q; t.Q = Qt; t.muh = t.mu; t.mu = 0.0;
                        t.pred = pred_here;
                        ++t.niter;
                        const bool newton_step = t.okprev != 4 && !fresh;       // a halved step says nothing about convergence
                        // Contraction: a full Newton step that does not even halve the correction of the full step before it.
                        // Three of them in a row, eight iterations into the alpha: the Newton matrix of this layout (binary16
                        // Gram products, binary32 elimination, at most NA coupled directions) does not fit this system and
                        // the iteration would crawl to its limit of 32 -- the alpha is handed over now (MXE_X_SLOW_STEPS)
                        {
                            const bool full = newton_step && t.okprev == 1 && t.muh == 0.0;
                            t.slow = (full && t.dHp > 0.0 && sdH > 0.25 * t.dHp) ? t.slow + 1 : 0;
                            t.dHp = full ? sdH : 0.0;
                        }
                        t.bt = 0;
                        on_path = newton_step && t.niter == 1 && relH2_min < (MXE_X_WALK_SKIP_TOL * MXE_X_WALK_SKIP_TOL) * t.Hn2;
                        if (newton_step && p.tol_h > 0.0 && relH2_min < tol2Hn && t.niter > p.miniter) { conv = 1; finish_alpha = true; }
                        else if (p.tol_relq > 0.0 && fabs(fabs(t.Qprev - t.Q) / t.Q) < p.tol_relq && t.niter > p.miniter) { conv = 1; finish_alpha = true; }
                        else if (t.niter >= ((LEAD && t.ia < 0 && t.ia > -t.lead) ? MXE_X_WALK_ITERS : p.mc_maxiter)) finish_alpha = true;   // (an alpha of a walk is a starting point: a few rounds, then on)
                        else if (t.wide > 0 && t.niter - t.wide >= MXE_X_ILL_ITERS) finish_alpha = true;    // (ill conditioned for the binary32 solve: handed over)
                        else if (MXE_X_SLOW_STEPS > 0 && t.slow >= MXE_X_SLOW_STEPS && t.niter >= 8 && (!LEAD || t.ia >= 0)) finish_alpha = true;    // (crawling: handed over)
                    }
                    MXE_STAMPA(2);
                    if (fresh && t.scratch == 2) t.scratch = 0;     // accepted or to be halved: the state in LDS is that trial point
                    if (finish_alpha) {
                        // (the leading alpha of a piece, number -1, writes its record where the piece's first alpha will
                        //  write its own over it -- no branch; where it fails, the first alpha starts from the state it
                        //  ended in and reports what becomes of it)
                        const size_t prob = (size_t)t.prob0 + max(t.ia, 0);
                        const bool own = !LEAD || t.ia >= 0;          // (an alpha of the walk leaves no record)
                        if (p.out_H && own) {
                            // H of the point just evaluated = the accepted one.  An alpha that FAILED (damping out of
                            // range, nothing finite to evaluate) ends on a rejected trial point: its H does not belong
                            // to the v, chi2, S, Q of the record (the last accepted state) and is written as NaN
                            double* Ho = p.out_H + prob * nw;
                            for (int i = lane; i < nw; i += 64) Ho[i] = failed ? __builtin_nan("") : Hi[i * MCC + q];
                        }
                        if (p.out_v && own) p.out_v[prob * NP + lane] = vv[q * NP + lane];
                        if (lane == 0 && own) {
                            p.out_chi2[prob] = t.chi2; p.out_S[prob] = t.S; p.out_Q[prob] = t.Q;
                            p.out_niter[prob] = t.niter; p.out_conv[prob] = conv;
                            p.out_nevals[prob] = t.nevals; p.out_nact[prob] = t.nact;
                        }
                        MXE_STAMPA(3);
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
                        if constexpr (LEAD) {
                            // the walk lands on every alpha that is a factor MXE_X_WALK_RATIO below the last landing (and on the
                            // last one before the piece's own): on the BASELINE mesh (ratio 0.87) every third alpha
                            if (t.ia < -1 && on_path) {
                                const double lo = t.alpha * (1.0 / MXE_X_WALK_RATIO), hi = t.alpha * MXE_X_WALK_RATIO;
                                while (t.ia < -1) {
                                    const double an = alpha_at(t, t.ia + 1);
                                    if (!(an >= lo && an <= hi)) break;
                                    ++t.ia;
                                }
                            }
                        }
                        if (!conv && p.mc_abandon) {
                            // An alpha this layout gave up on: the rest of the piece would start from a point that is not a
                            // solution and go the same way.  Its alphas are marked open -- mxe_chains_finish solves every run of
                            // open alphas as one warm-started chain from the converged alpha before it -- and the slot moves on
                            for (int i = max(t.ia, 0) + lane; i < t.clen; i += 64) {
                                const size_t pr = (size_t)t.prob0 + i;
                                p.out_conv[pr] = 0; p.out_niter[pr] = 0; p.out_nevals[pr] = 0; p.out_nact[pr] = 0;
                            }
                            t.ia = t.clen;
                        }
                        t.niter = 0; t.nevals = 0; t.mu = 0.0; t.bt = 0; t.capp = 0; t.pred = 0.0; t.wide = 0; t.slow = 0; t.dHp = 0.0;
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
                    MXE_STAMPA(4);
                    MXE_STAMPA(6);            // (no store: the state stays in registers for the solve below)
                }
                MXE_STAMPW(5);
            } else {
                load_slot(t);
            }
            int okflag = 0;
            double dk = 0.0;
            double dtot = 0.0;                       // total step from v of the trial point (dk: operand of the row pass)
            // binary16 range of the Gram operands: the row pass of this round writes sw = sqrt(w sc2) with
            // sc2 = 2^(8 - exponent of the largest w of the accepted point), so that sw <= 16 while w stays
            // below that maximum and a w that grows 2^24-fold still converts to a finite binary16 (the
            // conversion saturates beyond; such a trial point does not survive the test on Q)
            const double isc2 = ldexp(1.0, -ilogb(t.sc2));      // sc2 is a power of two: no division in this serial section
            {
                const double wm = (t.wmax > 1e-290 && t.wmax < 1e290) ? t.wmax : 1.0;
                const double sc2n = ldexp(1.0, 8 - ilogb(wm));
                if (lane == 0) { s_scw[q][0] = isc2; s_scw[q][1] = sc2n; }
                t.sc2 = sc2n;
            }
            if (t.active && t.scratch == 1) {
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
                MXE_STAMPH(0);
                // damping loop: raise mu until the factorisation succeeds and Bryan's bound holds
                while (true) {
                    const double a = t.alpha + t.mu;
                    double ia = __builtin_amdgcn_rcp(a);             // (the decoupled directions: z = rhs / a)
                    ia = fma(fma(-a, ia, 1.0), ia, ia);
                    ia = fma(fma(-a, ia, 1.0), ia, ia);
                    bool ok, small = false;
                    double nrm_gj = -1.0;
                    if constexpr (NA > 32) {
                        // (the build for more than 32 coupled directions: two sizes of the one-row-per-lane solve beside the
                        //  32-row two-half one)
                        if (na <= 32) ok = MXE_X_GJ_MFMA ? gj_home_m(a, na, isc2, nrm_gj, small) : gj_home(std::integral_constant<int, 32>{}, a, na, isc2, nrm_gj, small);
                        else if (na <= 48) ok = gj_home_rows(std::integral_constant<int, 48>{}, a, na, isc2, nrm_gj, small);
                        else ok = gj_home_rows(std::integral_constant<int, (NA > 48 ? 64 : 48)>{}, a, na, isc2, nrm_gj, small);
                    } else
#if MXE_X_GJ_MFMA
                    ok = gj_home_m(a, na, isc2, nrm_gj, small);
#else
                    if (na <= 16) ok = gj_home(std::integral_constant<int, 16>{}, a, na, isc2, nrm_gj, small);
                    else if (na <= 20) ok = gj_home(std::integral_constant<int, 20>{}, a, na, isc2, nrm_gj, small);
                    else if (na <= 24) ok = gj_home(std::integral_constant<int, 24>{}, a, na, isc2, nrm_gj, small);
                    else if (na <= 28) ok = gj_home(std::integral_constant<int, 28>{}, a, na, isc2, nrm_gj, small);
                    else ok = gj_home(std::integral_constant<int, 32>{}, a, na, isc2, nrm_gj, small);
#endif
                    // (a small pivot: the block is ill conditioned for 24 bits.  The step is still a descent direction --
                    //  an inexact Newton step -- but the iteration may crawl: the alpha is given up early, see step 4)
                    if (t.wide == 0 && __builtin_amdgcn_readfirstlane(__any(small ? 1 : 0))) t.wide = t.niter + 1;       // (pivots are wave-uniform)
                    if (ok) {
                        double z = 0.0, nrm = 0.0;
                        if (k < na) { z = zz[q * NP + k]; nrm = z * (rhs[q * NP + k] - a * z); }
                        else if (k < ns) z = rhs[q * NP + k] * ia;
                        nrm = nrm_gj;
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
            if (t.active && t.scratch == 2) {
                // the first step of a piece whose start state came from the table: the row pass evaluates
                // v - delta from scratch (u is not in the other waves' registers yet)
                dtot = okflag ? dk : 0.0;
                dk = vv[q * NP + k] - dtot;
            } else
            if (okflag != 4) dtot = okflag ? dk : 0.0;
            dlc[q * NP + k] = dtot;
            vecI[k * MCC + q] = dk;
            t.okprev = okflag;
            store_slot(t);
#if MXE_X_HOME_PRIO > 0
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
#endif
            MXE_STAMPH(0);
        }
        __syncthreads();
        MXE_STAMPW(2);
        if (!(s_act[0] | s_act[1] | s_act[2] | s_act[3])) break;      // uniform: LDS flags behind a barrier

#ifdef MXE_X_ROW_PRIO
        __builtin_amdgcn_s_setprio(MXE_X_ROW_PRIO);
#endif
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
                double2* Wz = reinterpret_cast<double2*>(Wt);
#pragma unroll
                for (int idx = 0; idx < (MCC * NPAIR * 128 + T - 1) / T; ++idx)
                    if (tid + idx * T < MCC * NPAIR * 128) Wz[tid + idx * T] = double2{0.0, 0.0};
                if (UREG) { hpart[tid] = 0.0; hpart[T + tid] = 0.0; }   // [2][MCC][NP] = 2 x 256 sums of h, added to in step 3
            }
            const int j = lane & 3;                              // slot of this lane's results
            const int drow = 4 * ((lane >> 2) & 3) + (lane >> 4);      // result row inside the tile
            const int ak = lane >> 4;                            // operand k inside the chunk
            const bool scr_j = s_scr[j] != 0;
            const bool pm_j = s_kind[j] != 0;
            const double isc_old = s_scw[j][0], sc_new = s_scw[j][1];
            const double* Dj = p.D + (size_t)((s_elem[j] >= 0) ? s_elem[j] : any_elem) * nwp;
            double pS = 0.0, pdH = 0.0, pHn = 0.0, pwm = 0.0, pdu = 0.0;
            const int nblk = nwp >> 5;                           // blocks of 32 omega rows = two 16-row tiles
            const int nchunk = (ns + 3) >> 2;                    // chunks of four singular directions
            constexpr int TB = (NWV == 8) ? 4 : 8;               // tiles per batch (accumulators)
            constexpr int RD = (WGPC == 1) ? MXE_X_RD1 : 4;           // ring depth in chunks (one workgroup per CU: registers to spare)
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
                    uo[tt] = !UREG ? ui[row * MCC + j] : (tt < 8 - UL) ? ureg[tt < 8 - UL ? tt : 0] : ui[(tt - (8 - UL)) * T + tid];
                    { const float so = swF[row * MCC + j]; wo[tt] = (double)(so * so) * isc_old; }
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
                        if (!UREG) ui[row * MCC + j] = uq;
                        else if (tt < 8 - UL) ureg[tt < 8 - UL ? tt : 0] = uq;
                        else ui[(tt - (8 - UL)) * T + tid] = uq;
                        Hi[row * MCC + j] = Hq;
                        swF[row * MCC + j] = __builtin_sqrtf(fminf((float)(wq * sc_new), 3.0e38f));
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
#ifdef MXE_X_ROW_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        MXE_STAMPW(3);
        __syncthreads();                         // Hi, wi, ui and the partial sums complete
        MXE_STAMPW(7);

#if MXE_X_FUSED_PRIO > 0
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(MXE_X_FUSED_PRIO);
#endif
        // ---- 3. fused pass (V once): h = V^T H (binary64) and W = V_a^T diag(w) V_a (split binary16) ----
        {
            // h_q = V^T H_q of the four slots is v_mfma_f64_4x4x4 (four blocks per instruction: block b =
            // columns 16 t + 4 b .. + 3 of V, the four slots as columns, K = the 4 omega rows of the
            // group): A[b][i][k] (lane 16 k + 4 b + i) = V[i0 + k][16 t + 4 b + i] is the register f[t] the
            // lane loaded; B[b][k][j] (lane 16 k + 4 b + j) = H[i0 + k][slot j] is one 8-byte LDS read;
            // D[b][i][j] lands on lane 16 i + 4 b + j: one accumulator per tile and no cross-lane sum.
            //
            // The Gram matrix only preconditions the Newton step (the residual rho that defines the answer
            // comes from h, binary64).  On gfx950 the binary32 and binary64 MFMAs run at the rate of -- and
            // instead of -- the SIMD's vector instructions; the binary16 / bfloat16 ones have a pipe of their
            // own, 16 times faster.  So the tiles are W = X^T X with X = diag(sw) V_a, sw = sqrt(w sc2) (the
            // row pass wrote it), every element of X split into two binary16 numbers, x = hi + lo (hi: x
            // rounded towards zero, lo: the remainder; 21 bits together), and
            //     X^T X ~ hi^T hi + hi^T lo + lo^T hi                 (three v_mfma_f32_16x16x32_f16,
            // binary32 accumulation; the dropped lo^T lo is 2^-20 of it).  Measured against binary64 on the
            // weights of the BASELINE spectra: 5e-7 of sqrt(W_ii W_jj) (binary32 MFMA: 1e-7; theta = 1e-5).
            // Operand layout of 16x16x32: lane (g = l >> 4, m = l & 15) holds A[m][8 g + e] / B[8 g + e][m],
            // e = 0..7.  A and B are the same matrix here, so ANY assignment of omega rows to (g, e) sums
            // over the right products as long as both operands use it: element e of lane (g, m) is row
            // 4 (row group e of the trip) + g, column 16 t + m -- exactly the registers the h product uses.
            typedef float g4 __attribute__((ext_vector_type(4)));
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            g4 acc[MCC][NPAIR];
            double hp[4];
#pragma unroll
            for (int c = 0; c < MCC; ++c)
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) acc[c][pr] = g4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) hp[t] = 0.0;
            const int kq = lane >> 4, cn = lane & 15;
            const int n_groups = nwp >> 2;           // 4 omega rows per group; multiple of 32
            struct HW { double h; float4 w; };       // H of slot (lane & 3), sw of the four slots, row i0 + kq
            auto loadHW = [&](HW& hw, const double* hsrc, const float* wsrc) {
                hw.h = hsrc[0];
                hw.w = *reinterpret_cast<const float4*>(wsrc);
            };
            // The waves take groups wave, wave + 4, ...; a trip = 8 row groups = one K = 32 step of the Gram
            // tiles.  V comes through a ring of DEPTH register sets (DEPTH - 1 row groups in flight), H / sw
            // from LDS two groups ahead.  n_groups is a multiple of 8 * 4 (n_omega_pad is a multiple of 128)
            // and every load is unconditional: the look-ahead past the end reads the zero rows behind V and
            // the padding behind H / sw in LDS and is not used.
            constexpr int ST = NWV;
            constexpr int TRIP = 8;
            constexpr int DEPTH = (WGPC == 1 && NWV == 4) ? MXE_X_DEPTH1 : MXE_X_DEPTH2;   // (256 registers: WGPC = 2 and the eight-wave build)
            static_assert(TRIP % DEPTH == 0 && DEPTH >= 2, "ring indices are static across trips");
            int g = wave;
            {
                double fr[DEPTH][4];
                HW hr[4];
                u4v xh[MCC][NT], xl[MCC][NT];        // packed binary16 hi / lo parts, element e = row group e of the trip
                float xe[MCC][NT];                   // the even row group of a pair waits for the odd one
                // V through buffer loads: one resource descriptor for the data set's V (wave-uniform), one
                // 32-bit lane offset, the position of the row group as a SCALAR offset -- the loop carries
                // no 64-bit vector address arithmetic
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, 0x7fffffff, 0x00020000);
                // 16 bytes per lane: columns 16 (2 t') + cn and 16 (2 t' + 1) + cn of the row sit side by side in Vx
                const int loff = (kq * NP + 2 * cn) * 8;              // bytes
                auto loadV = [&](double (&f)[4], int soff_bytes) {
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp) {
                        const u4v v4 = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, loff + 256 * tp, soff_bytes, 0);
                        f[2 * tp] = __hiloint2double((int)v4.y, (int)v4.x);
                        f[2 * tp + 1] = __hiloint2double((int)v4.w, (int)v4.z);
                    }
                };
                constexpr int VSTEPB = 4 * ST * NP * 8;              // bytes per group step (V)
                constexpr int HSTEP = 4 * ST * MCC;                  // entries per group step (H, sw in LDS)
                int vp = 4 * __builtin_amdgcn_readfirstlane(g) * NP * 8;          // byte offset of the wave's first row group (uniform)
                const double* hb = Hi + (size_t)(4 * g + kq) * MCC + (lane & 3);
                const float* wb = swF + (size_t)(4 * g + kq) * MCC;
#pragma unroll
                for (int j = 0; j < DEPTH - 1; ++j) loadV(fr[j], vp + j * VSTEPB);
                loadHW(hr[0], hb, wb);
                loadHW(hr[1], hb + HSTEP, wb + HSTEP);
                for (; g < n_groups; g += TRIP * ST, vp += TRIP * VSTEPB, hb += TRIP * HSTEP, wb += TRIP * HSTEP) {
#pragma unroll
                    for (int j = 0; j < TRIP; ++j) {
                        loadV(fr[(j + DEPTH - 1) % DEPTH], vp + (j + DEPTH - 1) * VSTEPB);
                        loadHW(hr[(j + 2) & 3], hb + (j + 2) * HSTEP, wb + (j + 2) * HSTEP);
                        const double (&f)[4] = fr[j % DEPTH];
                        const HW& hw = hr[j & 3];
#pragma unroll
                        for (int t = 0; t < 4; ++t) hp[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[t], hw.h, hp[t], 0, 0, 0);
                        // (the four products of a column tile as two packed multiplications, v_pk_mul_f32, and the remainders by
                        //  v_fma_mix_f32 written out: 212 -> 176 vector instructions per trip, cfg4 0.870 -> 0.861 ms on one box;
                        //  either alone: nothing, profiles/r03_c_experiments.txt)
                        typedef float f2v __attribute__((ext_vector_type(2)));
                        const f2v sw01 = {hw.w.x, hw.w.y}, sw23 = {hw.w.z, hw.w.w};
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float ff = (float)f[t];
#ifndef MXE_X_NO_PKMUL
                            const f2v ff2 = {ff, ff};
                            const f2v x01 = ff2 * sw01, x23 = ff2 * sw23;
                            const float xs[MCC] = {x01.x, x01.y, x23.x, x23.y};
#else
                            const float xs[MCC] = {ff * hw.w.x, ff * hw.w.y, ff * hw.w.z, ff * hw.w.w};
#endif
#pragma unroll
                            for (int c = 0; c < MCC; ++c) {
                                const float xv = xs[c];
                                if (j & 1) {
                                    const float x0 = xe[c][t];
                                    const auto hh = __builtin_amdgcn_cvt_pkrtz(x0, xv);
                                    const unsigned hu = __builtin_bit_cast(unsigned, hh);
#ifdef MXE_X_NO_MIX    // (the compiler fuses this conversion and subtraction into v_fma_mix_f32 itself when the products are single multiplications; behind the packed ones it does not: 0.898 against 0.870 ms)
                                    const auto ll = __builtin_amdgcn_cvt_pkrtz(x0 - (float)hh[0], xv - (float)hh[1]);
#else
                                    // remainder x - hi in ONE instruction per element: v_fma_mix_f32 reads the binary16
                                    // half of hu as an operand (hi * -1 + x, exact), no v_cvt_f32_f16 + v_sub_f32 pair
                                    float l0, l1;
                                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hu), "v"(x0));
                                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hu), "v"(xv));
                                    const auto ll = __builtin_amdgcn_cvt_pkrtz(l0, l1);
#endif
                                    xh[c][t][j >> 1] = hu;
                                    xl[c][t][j >> 1] = __builtin_bit_cast(unsigned, ll);
                                } else {
                                    xe[c][t] = xv;
                                }
                            }
                        }
                    }
                    // the K = 32 step of every tile: hi^T hi, hi^T lo, lo^T hi (independent accumulators between
                    // the dependent ones)
#pragma unroll
                    for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                        for (int c = 0; c < MCC; ++c) {
                            int pr = 0;
#pragma unroll
                            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                                for (int nt = mt; nt < NT; ++nt) {
                                    const h8 a = __builtin_bit_cast(h8, prod == 2 ? xl[c][mt] : xh[c][mt]);
                                    const h8 b = __builtin_bit_cast(h8, prod == 1 ? xl[c][nt] : xh[c][nt]);
                                    acc[c][pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c][pr], 0, 0, 0);
                                    ++pr;
                                }
                        }
                }
            }
#if MXE_X_FUSED_PRIO > 0
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
#endif
            MXE_STAMPW(1);
            // h: lane 16 i + 4 b + j holds column 16 t + 4 b + i of slot j; the waves are summed in step 4
            {
                const int hj = lane & 3, hcol = 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (UREG) __hip_atomic_fetch_add(hpart + ((wave >> 1) * MCC + hj) * NP + 16 * t + hcol, hp[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else hpart[(wave * MCC + hj) * NP + 16 * t + hcol] = hp[t];
                }
            }
            MXE_STAMPW(6);
            // Gram tiles: every wave adds its partial tiles into the slots' tiles with LDS atomics
            // (ds_add_f64, one lane-contiguous instruction per accumulator register; the tiles were zeroed
            // at the start of the row pass)
            {
#pragma unroll
                for (int c = 0; c < MCC; ++c)
#pragma unroll
                    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            __hip_atomic_fetch_add(Wt + ((c * NPAIR + pr) * 4 + r) * 64 + lane, (double)acc[c][pr][r],
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
        }
        MXE_STAMPW(4);

        first_round = false;
#ifdef MXE_PROFILE
        ++prof_rounds;
#endif
    }
    // (not in the build of the batch that fills the GPU, <32, 2> without led pieces: its depth is not what bounds it, and the
    //  store cost that kernel 0.6 % -- 0.8173 against 0.8127 ms on one box, register allocation: profiles/r04_experiments.txt)
    if constexpr (LEAD || WGPC == 1)
        if (dynamic && x.wg_chains && tid == 0) const_cast<int*>(x.wg_chains)[blockIdx.x] = (int)guard - 1;      // (the last entry of the loop left it before its passes)
#ifdef MXE_PROFILE
    if (lane == 0 && p.prof && blockIdx.x < 1024) {
        long long* pr = p.prof + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int r = 0; r < 7; ++r) pr[r] = prof_acc[r];
        pr[7] = (wave == 0) ? prof_rounds : prof_acc[7];     // wave 0: rounds; others: wait at the row-pass barrier
    }
#endif
}

#if MXE_X_KERNARG_RELOAD
#undef p
#endif

} // namespace mxe
