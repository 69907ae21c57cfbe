// maxent_hip.hip -- C-ABI of libmaxent_hip.so (see include/maxent_hip.h)
//
// Host side of the drop-in boundary: owns the device, whitens each data set
// (one-sided Jacobi SVD of diag(1/err) U S, done once per (U, err) pair on
// the host), stages the singular basis in HBM in the two layouts the kernel
// streams (V row-major for h = V^T H and the Gram matrix, V^T for u = V v),
// launches the chain kernel on its own stream and times it with HIP events.
#include "../../include/maxent_hip.h"
#include "mxe_kernel.hip.h"
#include "mxe_kernel_mc.hip.h"
#include "mxe_kernel_lv.hip.h"
#include "mxe_eval.hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <map>
#include <vector>
#include <mutex>
#include <thread>
#include <array>
#include <tuple>

using mxe::KParams;

namespace {

struct DataSet {
    int n_rows = 0;
    bool identity_q = true;
    std::vector<double> Q;      // n_s x n_s (row-major), caller basis <- whitened basis
    std::vector<double> c;      // n_s
    std::vector<double> Uhat;   // n_rows x n_s
    std::vector<double> err;    // n_rows
};

// Wait for the ctx stream by polling: a blocking hipStreamSynchronize was seen to return tens of
// milliseconds after the work had finished in a quarter of the processes on the MI355X boxes (the
// device time of 20 launches 33 ms, the host-side wait 67 ms); the launches last 1-20 ms, so the host
// thread spins on hipStreamQuery for up to two seconds and only then blocks.
static hipError_t stream_wait(hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return hipStreamSynchronize(s);
    }
}

// Device allocations are kept when a context lets go of them and handed to the next one that asks (per device, best
// fit): a fresh TauMaxEnt / ElementwiseMaxEnt object -- a new kernel, so a new context with its 100+ MB of result
// buffers -- otherwise pays hipMalloc + hipFree of that size, 50-150 ms per object on the MI355X boxes against 25 ms for
// everything else it does.  At most 4 GiB and 512 blocks are kept; what comes out of the pool is NOT zeroed (nothing
// here relies on zeroed memory: hipMalloc does not promise it either).
struct DevPool {
    struct Block { void* p; size_t bytes; int device; };
    std::mutex mu;
    std::vector<Block> blocks;
    size_t held = 0;
    static constexpr size_t MAX_HELD = (size_t)4 << 30, MAX_BLOCKS = 512;
    void* take(size_t bytes, size_t* got) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].device == dev && blocks[i].bytes >= bytes && blocks[i].bytes <= 2 * bytes + 65536 &&
                (best < 0 || blocks[i].bytes < blocks[best].bytes)) best = (int)i;
        if (best < 0) return nullptr;
        void* p = blocks[best].p; *got = blocks[best].bytes;
        held -= blocks[best].bytes;
        blocks.erase(blocks.begin() + best);
        return p;
    }
    bool give(void* p, size_t bytes) {
        int dev = 0;
        if (getenv("MXE_NO_POOL") || hipGetDevice(&dev) != hipSuccess) return false;
        std::lock_guard<std::mutex> lk(mu);
        if (held + bytes > MAX_HELD || blocks.size() >= MAX_BLOCKS) return false;
        blocks.push_back(Block{p, bytes, dev});
        held += bytes;
        return true;
    }
};
DevPool g_pool;

// Large device-to-host copies into pageable memory (all H of a launch: 102 MB for the BASELINE batch) run at 2.5-5 GB/s
// through hipMemcpy on these boxes.  d2h_pipelined copies in chunks through two pinned staging buffers (allocated once
// per process): the DMA of chunk k + 1 runs while the host copies chunk k out of the other buffer.
struct PinnedPair {
    std::mutex mu;
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    static constexpr size_t CHUNK = (size_t)8 << 20;
    bool ready() {
        if (buf[0]) return true;
        for (int k = 0; k < 2; ++k) {
            if (hipHostMalloc(&buf[k], CHUNK, hipHostMallocDefault) != hipSuccess) { buf[0] = nullptr; return false; }
            if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { buf[0] = nullptr; return false; }
        }
        return true;
    }
};
// one pair per device: an event belongs to the device it was created on, and d2h_pipelined records it on the stream of
// whichever context is fetching (with one process-wide pair, created on the device of the first context, a fetch
// on another device failed with hipErrorInvalidHandle -- BatchSolver(device_ids = (0, 1, ...)) from ~22 scans of 500
// frequencies on)
constexpr int MXE_MAX_DEVICES = 64;
PinnedPair g_pinned_dev[MXE_MAX_DEVICES];

// Pinned host memory for the callers' result arrays (mxe_host_alloc / mxe_host_free): a copy into it runs at the rate of the
// link (102 MB: 2-3 ms) where one into pageable memory takes 11-22 ms even through the staging pair above -- page faults of
// the fresh destination and the second pass of the host copy.  Blocks are kept (a few, best fit) because pinning costs
// about as much as the copy it saves.
struct HostPool {
    std::mutex mu;
    std::map<void*, size_t> live;                // handed out: base -> bytes
    std::vector<std::pair<void*, size_t>> idle;
    static constexpr size_t KEEP = 24;     // (the scalars and rows of jobs in flight are ~1 MB blocks, several per job)
    static constexpr size_t KEEP_BIG = 3, BIG = (size_t)16 << 20;
    void* take(size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].second >= bytes && idle[i].second <= 2 * bytes + (1u << 20) && (best < 0 || idle[i].second < idle[best].second)) best = (int)i;
        void* p = nullptr; size_t got = bytes;
        if (best >= 0) { p = idle[best].first; got = idle[best].second; idle.erase(idle.begin() + best); }
        else if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        live[p] = got;
        return p;
    }
    void give(void* p) {
        std::unique_lock<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) return;
        const size_t bytes = it->second;
        live.erase(it);
        idle.emplace_back(p, bytes);
        void* drop = nullptr;
        // (of the large blocks -- all H of a launch -- three are kept, as before; the oldest goes first)
        size_t big = 0;
        for (auto& b : idle) big += b.second >= BIG;
        if (big > KEEP_BIG) {
            for (size_t i = 0; i < idle.size(); ++i) if (idle[i].second >= BIG) { drop = idle[i].first; idle.erase(idle.begin() + i); break; }
        } else if (idle.size() - big > KEEP) {
            for (size_t i = 0; i < idle.size(); ++i) if (idle[i].second < BIG) { drop = idle[i].first; idle.erase(idle.begin() + i); break; }
        }
        lk.unlock();
        if (drop) (void)hipHostFree(drop);
    }
    bool holds(const void* q, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.upper_bound(const_cast<void*>(q));
        if (it == live.begin()) return false;
        --it;
        const char* b = (const char*)it->first;
        return (const char*)q >= b && (const char*)q + bytes <= b + it->second;
    }
};
HostPool g_host;

hipError_t d2h_pipelined(void* dst, const void* src, size_t bytes, hipStream_t s)
{
    if (g_host.holds(dst, bytes)) {              // a destination of mxe_host_alloc: one DMA, no staging
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
        return e != hipSuccess ? e : hipStreamSynchronize(s);
    }
    // (also the analyzers' rows of a launch, 3 MB: hipMemcpy into pageable memory took 0.7-1.1 ms for them, this way 0.3)
    if (bytes < ((size_t)256 << 10)) { hipError_t e0 = hipStreamSynchronize(s); return e0 != hipSuccess ? e0 : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MXE_MAX_DEVICES) dev = -1;
    PinnedPair* pp = dev >= 0 ? &g_pinned_dev[dev] : nullptr;
    std::unique_lock<std::mutex> lk;
    if (pp) lk = std::unique_lock<std::mutex>(pp->mu, std::try_to_lock);
    if (!pp || !lk.owns_lock() || !pp->ready()) {
        // (another thread's copy through this device's pair is in flight, or no pinned memory: a plain copy -- which runs on the
        //  null stream and does NOT wait for work queued on the non-blocking stream s: wait for it first)
        hipError_t e0 = hipStreamSynchronize(s);
        return e0 != hipSuccess ? e0 : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    }
    PinnedPair& g_pinned = *pp;
    const size_t C = PinnedPair::CHUNK, n = (bytes + C - 1) / C;
    hipError_t e = hipSuccess;
    for (size_t k = 0; k <= n && e == hipSuccess; ++k) {
        if (k < n) {
            const size_t len = std::min(C, bytes - k * C);
            e = hipMemcpyAsync(g_pinned.buf[k & 1], (const char*)src + k * C, len, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipEventRecord(g_pinned.ev[k & 1], s);
        }
        if (k > 0 && e == hipSuccess) {
            const size_t j = k - 1, len = std::min(C, bytes - j * C);
            while ((e = hipEventQuery(g_pinned.ev[j & 1])) == hipErrorNotReady) {}
            if (e == hipSuccess) std::memcpy((char*)dst + j * C, g_pinned.buf[j & 1], len);
        }
    }
    return e;
}

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    size_t bytes = 0;                            // of the allocation (>= n * sizeof(T) when it came from the pool)
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        const size_t need = std::max<size_t>(count, 1) * sizeof(T);
        if (p && need <= bytes) { n = count; return hipSuccess; }
        if (p) { hipDeviceSynchronize(); release(); }      // (growing: whoever still reads the old block finishes first, as hipFree made sure)
        size_t got = 0;
        void* q = g_pool.take(need, &got);
        static const bool poison = getenv("MXE_POISON_ALLOC") != nullptr;      // (debugging aid: nothing may rely on what a fresh block holds)
        // (the fill runs on the null stream and the contexts' streams do not wait for that one: it has to be through before
        //  anybody writes the block -- without the wait a staged array now and then came out with the pattern in it)
        if (q) { p = (T*)q; bytes = got; n = count; if (poison) { hipMemset(p, 0xA5, bytes); hipDeviceSynchronize(); } return hipSuccess; }
        hipError_t e = hipMalloc((void**)&p, need);
        if (e == hipSuccess) { n = count; bytes = need; if (poison) { hipMemset(p, 0xA5, bytes); hipDeviceSynchronize(); } }
        return e;
    }
    // (callers make sure nothing in flight uses the block: mxe_ctx_destroy waits for its stream first)
    void release() { if (p && !g_pool.give(p, bytes)) hipFree(p); p = nullptr; n = 0; bytes = 0; }
};

} // namespace

struct mxe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mark = nullptr;
    int n_tau = 0, n_omega = 0, n_s = 0, NP = 64, nwp = 0;
    std::vector<double> U, S, V;
    std::vector<DataSet> ds;
    bool ds_dirty = true;
    // elements (host)
    int n_elem = 0;
    std::vector<int> elem_ds, elem_kind;
    std::vector<double> h_sumD, h_D;           // h_D: [n_elem][nwp], for the classes of the start-state table
    // chains
    int n_chain = 0, n_alpha = 0;
    std::vector<int> chain_elem;      // per parent chain
    std::vector<int> sub_elem, sub_prob0, sub_len, sub_v0, wg_chains;   // per launched (sub-)chain
    int n_sub = 0, n_wg = 0, mc_na = 0, mc_wgpc = 1, wgpc_auto = 2, n_queue = 0, n_solo = 0;
    std::vector<int> excluded;          // problems no piece of the lock-step launch covers (more than 32 coupled directions): mxe_chains_finish
    DevBuf<int> dexcluded;
    bool mc_gst = false;              // lock-step layout with u, H, sw in device memory (frequency meshes beyond the LDS)
    // chain_kernel_lv (V^T resident in LDS as binary32, mxe_kernel_lv.hip.h): 0 = not used, 1 = it IS the launch
    // (mxe_opts.precision = F32), 2 = first pass of a binary64 launch that does not fill the GPU -- chain_kernel_mc then
    // takes every alpha as a piece of its own from the v it left (the arrays of that second pass: d2_*)
    int lv_mode = 0;
    int mc_wgpc_hint = 1;               // workgroups per CU the auto rule of mxe_chains_upload expects (piece_taper)
    int n_wg2 = 0, wgpc2 = 1;
    DevBuf<int> d2_elem, d2_prob0, d2_len, d2_v0, d2_queue, dcnt1_niter, dcnt1_nevals;
    DevBuf<int> drounds;                // rounds per workgroup of the last lock-step launch: [n_wg] | [n_wg2] (mxe_launch_depth)
    int rounds_n[2] = {0, 0};
    int placement_checked = 0;          // the last upload that wanted solo workgroups: 0 none did, 1 the placement rule holds, 2 it does not
    DevBuf<double> dgstate_mc;
    std::vector<int> queue;
    std::vector<int> sub_pre;                                           // leading alpha of a piece: entries before its first alpha (0: none)
    std::vector<int> sub_walk0;                                         // led piece: its ladder starts at walk_alpha[sub_walk0] (-1: it walks the scan's mesh)
    std::vector<double> walk_alpha;                                     // the ladders of the led pieces of a coarse mesh (mxe_chains_upload)
    bool has_walk = false;
    bool has_pre = false;
    mxe_opts opts;
    bool chains_ready = false, launched = false;
    int last_nw = 0, last_lds = 0;
    std::string last_kernel;
    // device
    DevBuf<float> dVf, dVtf;          // binary32 copies of dV / dVt (mxe_opts.precision = F32)
    DevBuf<double> dsel3;                        // mxe_select3_launch: [3][n_chain] indices | [3][n_chain][n_omega] rows
    std::vector<double> h_sel3;
    // copies queued behind the launch (mxe_chains_prefetch / mxe_select3_prefetch_rows): where to, until a launch, an upload or a
    // finishing pass that ran makes them stale.  The fetch calls given the same destinations then only wait.
    double* pre_d = nullptr; int32_t* pre_i = nullptr; bool pre_scaled = false;
    double* pre_rows = nullptr; int pre_first = 0, pre_count = 0; bool pre_index = false;
    double* h_sel3_pinned = nullptr; size_t h_sel3_pinned_n = 0;
    // mxe_elements_update_data projects on the device: Uhat | err of every data set, built at its first call
    DevBuf<double> dproj_U, dproj_err, dproj_G;
    DevBuf<long long> dproj_off;                 // [n_ds][3]: offset of Uhat, offset of err, rows
    bool proj_ready = false;
    DevBuf<double> dV, dVx, dVt, dc, dcinv, dghat, dcperp, dD, dsumD, dalpha, dv0;
    DevBuf<int> delem_ds, delem_kind, dchain_elem, dsub_prob0, dsub_len, dsub_v0, dwg_chains;
    // H, chi2, S, Q live back to back in ONE allocation (dout_pack) so that a
    // multi-GPU driver can move all per-alpha results with a single collective
    DevBuf<double> dout_v, dout_pack, dout_pack2, dB, dA, dlogdet;
    DevBuf<int> dparent_elem;
    int result_buffer = 0;        // which of the two result allocations launches write to
    struct View { double* p = nullptr; } dout_H, dout_chi2, dout_S, dout_Q;
    DevBuf<int> dout_niter, dout_nact;       // dout_niter: niter [P] | converged [P] | nevals [P] in ONE block (mxe_chains_fetch: one copy)
    struct IView { int* p = nullptr; } dout_conv, dout_nevals;
    DevBuf<long long> dprof;
    DevBuf<int> dqueue, dcounter;
    DevBuf<int> dsub_pre, dsub_init, dsub_walk0;
    DevBuf<double> dwalk_alpha;
    DevBuf<double> dinit_tab;           // start states per class of pieces (KParams::init_tab)
    DevBuf<double> dgstate;             // omega-space state of the chains when it does not fit LDS (KParams::gstate)
    DevBuf<int> dfin_budget, dfin_out;  // mxe_chains_finish: iterations every entry may still spend; where it writes its record (-1: a rung)
    DevBuf<double> dfin_alpha;          //   the alphas of the entries (mesh alphas and rungs)
    DevBuf<int> dfin_elem, dfin_prob0, dfin_len, dfin_v0;       // mxe_chains_finish: one piece per alpha that is solved again
    DevBuf<double> dfin_start;          //   and its start vector (the state the lock-step kernel left)
    int sel3_nc = 0;                    // scans of the launch the last mxe_select3_launch chose for (0: none since the chains were uploaded)
    int last_finished = 0;              // alphas the last mxe_chains_finish solved again
    int last_rungs = 0;                 //   rungs it laid between them (coarse meshes)
    bool has_init = false;
    // mxe_eval_batch / mxe_audit scratch
    DevBuf<double> ev_x, ev_alpha, ev_scal, ev_vecw, ev_vecs, ev_mat;
    DevBuf<int> ev_elem;
    DevBuf<double> rows_out;        // mxe_fetch_rows: the selected rows, gathered on the device
    DevBuf<int> rows_idx;
    double chi2_factor = 1.0;     // of the staged chains (mxe_opts.chi2_factor)
    struct mxe_comm_state* comm = nullptr;      // gather between GPUs (mxe_comm_*)
    std::string hip_err;
};

// nothing may propagate through the C boundary: function-try-blocks of the entry points
#define MXE_CATCH_ALL \
    catch (const std::bad_alloc&) { return MXE_ERR_NOMEM; } \
    catch (...) { return MXE_ERR_ARG; }

#define HIPCHK(ctx, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { \
    (ctx)->hip_err = std::string(#call) + ": " + hipGetErrorString(e__); return MXE_ERR_HIP; } } while (0)

namespace {

// One-sided Jacobi (Hestenes) SVD of C (m x n, row-major, m >= 1):
// C = Uhat diag(c) Q^T.  On exit C's columns hold Uhat*c, Q the rotations.
// High relative accuracy for columns of wildly different scale, which is
// exactly the situation of diag(1/err) U S (S spans 1e1 .. 1e-14).
void jacobi_svd(std::vector<double>& C, int m, int n, std::vector<double>& Q)
{
    Q.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) Q[(size_t)i * n + i] = 1.0;
    const double eps = 2.220446049250313e-16;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double a = 0.0, b = 0.0, g = 0.0;
                for (int i = 0; i < m; ++i) {
                    const double x = C[(size_t)i * n + p], y = C[(size_t)i * n + q];
                    a += x * x; b += y * y; g += x * y;
                }
                if (a == 0.0 || b == 0.0) continue;
                if (std::fabs(g) <= eps * std::sqrt(a * b)) continue;
                rotated = true;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) /
                                 (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < m; ++i) {
                    const double x = C[(size_t)i * n + p], y = C[(size_t)i * n + q];
                    C[(size_t)i * n + p] = cs * x - sn * y;
                    C[(size_t)i * n + q] = sn * x + cs * y;
                }
                for (int i = 0; i < n; ++i) {
                    const double x = Q[(size_t)i * n + p], y = Q[(size_t)i * n + q];
                    Q[(size_t)i * n + p] = cs * x - sn * y;
                    Q[(size_t)i * n + q] = sn * x + cs * y;
                }
            }
        }
        if (!rotated) break;
    }
}

int build_dataset(mxe_ctx* ctx, int n_rows, const double* U_rot, const double* err, DataSet& d)
{
    const int ns = ctx->n_s;
    for (int i = 0; i < n_rows; ++i)
        if (!(err[i] > 0.0) || !std::isfinite(err[i])) return MXE_ERR_NUMERIC;
    d.n_rows = n_rows;
    d.err.assign(err, err + n_rows);
    const double* U = U_rot ? U_rot : ctx->U.data();
    bool scalar_err = (U_rot == nullptr);
    for (int i = 1; i < n_rows && scalar_err; ++i) scalar_err = (err[i] == err[0]);
    d.c.resize(ns);
    d.Uhat.assign((size_t)n_rows * ns, 0.0);
    if (scalar_err) {
        // the ctx's U has orthonormal columns: C = U diag(S/err) already
        d.identity_q = true;
        d.Q.clear();
        for (int k = 0; k < ns; ++k) d.c[k] = ctx->S[k] / err[0];
        std::copy(U, U + (size_t)n_rows * ns, d.Uhat.begin());
    } else {
        d.identity_q = false;
        std::vector<double> C((size_t)n_rows * ns);
        for (int i = 0; i < n_rows; ++i)
            for (int k = 0; k < ns; ++k)
                C[(size_t)i * ns + k] = U[(size_t)i * ns + k] * ctx->S[k] / err[i];
        std::vector<double> Q;
        jacobi_svd(C, n_rows, ns, Q);
        std::vector<double> nrm(ns);
        for (int k = 0; k < ns; ++k) {
            double s = 0.0;
            for (int i = 0; i < n_rows; ++i) s += C[(size_t)i * ns + k] * C[(size_t)i * ns + k];
            nrm[k] = std::sqrt(s);
        }
        std::vector<int> order(ns);
        for (int k = 0; k < ns; ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nrm[a] > nrm[b]; });
        d.Q.assign((size_t)ns * ns, 0.0);
        for (int kk = 0; kk < ns; ++kk) {
            const int k = order[kk];
            d.c[kk] = nrm[k];
            for (int i = 0; i < ns; ++i) d.Q[(size_t)i * ns + kk] = Q[(size_t)i * ns + k];
            if (nrm[k] > 0.0)
                for (int i = 0; i < n_rows; ++i)
                    d.Uhat[(size_t)i * ns + kk] = C[(size_t)i * ns + k] / nrm[k];
        }
    }
    double cmax = 0.0;
    for (int k = 0; k < ns; ++k) cmax = std::max(cmax, d.c[k]);
    if (!(cmax > 0.0)) return MXE_ERR_NUMERIC;
    for (int k = 0; k < ns; ++k) d.c[k] = std::max(d.c[k], 1e-150 * cmax);
    return MXE_OK;
}

int upload_bases(mxe_ctx* ctx)
{
    const int nds = (int)ctx->ds.size(), ns = ctx->n_s, NP = ctx->NP, nw = ctx->n_omega, nwp = ctx->nwp;
    // V carries zero rows behind the last data set for the look-ahead of the fused pass
    std::vector<double> hV(((size_t)nds * nwp + mxe::MC_LOOKAHEAD_ROWS) * NP, 0.0), hVt((size_t)nds * NP * nwp + 2 * mxe::MC_LOOKAHEAD_ROWS, 0.0);
    std::vector<double> hc((size_t)nds * NP, 1.0), hci((size_t)nds * NP, 1.0);
    for (int d = 0; d < nds; ++d) {
        const DataSet& D = ctx->ds[d];
        for (int i = 0; i < nw; ++i) {
            for (int k = 0; k < ns; ++k) {
                double val;
                if (D.identity_q) val = ctx->V[(size_t)i * ns + k];
                else {
                    val = 0.0;
                    for (int j = 0; j < ns; ++j) val += ctx->V[(size_t)i * ns + j] * D.Q[(size_t)j * ns + k];
                }
                hV[((size_t)d * nwp + i) * NP + k] = val;
                hVt[((size_t)d * NP + k) * nwp + i] = val;
            }
        }
        for (int k = 0; k < ns; ++k) { hc[(size_t)d * NP + k] = D.c[k]; hci[(size_t)d * NP + k] = 1.0 / D.c[k]; }
    }
    HIPCHK(ctx, ctx->dV.ensure(hV.size()));
    if (NP == 64) {
        // the lock-step fused pass reads V with 16-byte loads (the address unit takes a wave-instruction in 16 cycles
        // whatever its width: 8 B per lane stream at 32 B per cycle and CU, 16 B at 64 -- tools/l2_stream_rate.hip):
        // a lane's two doubles are the same lane position of two neighbouring 16-column tiles
        std::vector<double> hVx(hV.size(), 0.0);
        for (size_t r = 0; r < hV.size() / NP; ++r)
            for (int tp = 0; tp < 2; ++tp)
                for (int m = 0; m < 16; ++m)
                    for (int sdx = 0; sdx < 2; ++sdx)
                        hVx[r * NP + 32 * tp + 2 * m + sdx] = hV[r * NP + 16 * (2 * tp + sdx) + m];
        HIPCHK(ctx, ctx->dVx.ensure(hVx.size()));
        HIPCHK(ctx, hipMemcpyAsync(ctx->dVx.p, hVx.data(), hVx.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, stream_wait(ctx->stream));
    }
    HIPCHK(ctx, ctx->dVt.ensure(hVt.size()));
    HIPCHK(ctx, ctx->dc.ensure(hc.size()));
    HIPCHK(ctx, ctx->dcinv.ensure(hci.size()));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dV.p, hV.data(), hV.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dVt.p, hVt.data(), hVt.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dc.p, hc.data(), hc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dcinv.p, hci.data(), hci.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<float> hVf(hV.begin(), hV.end()), hVtf(hVt.begin(), hVt.end());
    HIPCHK(ctx, ctx->dVf.ensure(hVf.size()));
    HIPCHK(ctx, ctx->dVtf.ensure(hVtf.size()));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dVf.p, hVf.data(), hVf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dVtf.p, hVtf.data(), hVtf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    ctx->ds_dirty = false;
    return MXE_OK;
}

// dynamic LDS of chain_kernel_mc<NA, WGPC> in bytes (the carve at the top of the kernel)
size_t mc_lds_bytes(int NA, int nwp, int wgpc, int nwv = 4, bool gst = false)
{
    const int NT = NA / 16, NPAIR = NT * (NT + 1) / 2;
    if (gst) nwp = 0;                        // (u, H, sw in device memory: MCExtra::gstate)
    const size_t doubles = 8 * 4 * 64 + 2 * 64 + 64 * 4 + (wgpc == 2 ? 2 : nwv) * 4 * 64 + nwv * 32 + 2 * 4 * 64 +   // vectors, c, 1/c, step, h, sums, solve scales
                           (wgpc == 2 ? (size_t)MXE_X_UL * 256 + (size_t)nwp * 4 : (size_t)2 * nwp * 4) +
                           (size_t)4 * NPAIR * 256;                                                  // u, H, Gram tiles
    const size_t floats = (size_t)nwp * 4;                                              // sw
    return doubles * 8 + floats * 4;
}

// LDS bytes of chain_kernel<NW, NAB, TS>: stream arrays (u, ut, w, wt, Hs, vecs) in the stream
// type, the Gram staging area only in the binary64 build
// (gst: the five omega arrays live in device memory, KParams::gstate)
size_t lds_bytes(int NP, int nwp, int NW, bool f32, bool gst = false)
{
    const int SROW = (NP / 4) * mxe::GBLK;
    const size_t fixed = (size_t)NP * (NP + 1) + (NP == 64 ? 13 : 11) * (size_t)NP + (size_t)NW * NP + (size_t)NW * 8;
    const size_t stream = (gst ? 0 : 5 * (size_t)nwp) + NP;
    const size_t stage = f32 ? 0 : (size_t)NW * 2 * mxe::GRAM_R * SROW;
    return (fixed + stage) * 8 + stream * (f32 ? 4 : 8);
}

template <int NW, int NAB, typename TS = double, bool GST = false>
hipError_t launch_t(const KParams& kp, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute((const void*)mxe::chain_kernel<NW, NAB, TS, GST>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mxe::chain_kernel<NW, NAB, TS, GST>), dim3(kp.n_chain), dim3(64 * NW), lds, s, kp);
    return hipGetLastError();
}

template <int NW>
hipError_t launch_nab(int NP, const KParams& kp, size_t lds, hipStream_t s)
{
    return (NP == 64) ? launch_t<NW, 2>(kp, lds, s) : launch_t<NW, 4>(kp, lds, s);
}

} // namespace

extern "C" {

const char* mxe_version(void) { return "maxent_hip 0.1 (gfx950)"; }

#ifndef MXE_SRC_HASH
#define MXE_SRC_HASH "unknown"
#endif
const char* mxe_source_hash(void) { return MXE_SRC_HASH; }
void* mxe_host_alloc(size_t bytes) { return bytes ? g_host.take(bytes) : nullptr; }
void mxe_host_free(void* p) { if (p) g_host.give(p); }

const char* mxe_strerror(int code)
{
    switch (code) {
        case MXE_OK: return "ok";
        case MXE_ERR_ARG: return "invalid argument";
        case MXE_ERR_HIP: return "HIP runtime error (see mxe_last_hip_error)";
        case MXE_ERR_NODEVICE: return "no usable HIP device";
        case MXE_ERR_STATE: return "call order violated";
        case MXE_ERR_LIMIT: return "problem exceeds kernel limits (n_s <= 128, 160 KB LDS per chain)";
        case MXE_ERR_NUMERIC: return "numerical failure (whitening: error bars must be finite and > 0; device SVD: Jacobi sweeps exhausted)";
        case MXE_ERR_NOMEM: return "out of host memory";
        default: return "unknown error";
    }
}

int mxe_device_count(int* n)
{
    if (!n) return MXE_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return MXE_ERR_NODEVICE; }
    *n = c;
    return MXE_OK;
}

void mxe_opts_default(mxe_opts* o)
{
    if (!o) return;
    o->maxiter = 1000; o->miniter = 0;
    o->tol_h = 1e-9; o->tol_d = 0.0; o->tol_relq = 0.0;
    o->step_max = 0.2; o->mu_first = 1e-3; o->mu_grow = 4.0; o->mu_max = 1e20;
    o->decouple_tol = 1e-5;
    o->waves_per_chain = 0; o->chains_per_wg = 0; o->alpha_split = 0; o->stop_estimate = 1;
    o->precision = MXE_PRECISION_F64; o->wg_per_cu = 0;
    o->chi2_factor = 1.0;
    o->lds_basis = 0; o->in_flight = 0;
}

int mxe_ctx_create(int device, int n_tau, int n_omega, int n_s,
                   const double* U, const double* S, const double* V, mxe_ctx** out)
try {
    if (!out || !S || !V || n_tau < 1 || n_omega < 1 || n_s < 1) return MXE_ERR_ARG;
    if (n_s > 128) return MXE_ERR_LIMIT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MXE_ERR_NODEVICE;
    if (device < 0 || device >= ndev) return MXE_ERR_ARG;
    mxe_ctx* ctx = new (std::nothrow) mxe_ctx();
    if (!ctx) return MXE_ERR_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return MXE_ERR_NODEVICE; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreate(&ctx->ev_mark) != hipSuccess) {
        delete ctx; return MXE_ERR_HIP;
    }
    ctx->n_tau = n_tau; ctx->n_omega = n_omega; ctx->n_s = n_s;
    ctx->NP = (n_s <= 64) ? 64 : 128;
    ctx->nwp = ((n_omega + 127) / 128) * 128;   // the lock-step kernel's fused pass runs whole trips of 128 rows
    if (ctx->nwp > 1536) ctx->nwp = ((n_omega + 255) / 256) * 256;     // (long meshes run its eight-wave build: trips of 256)
    if (U) ctx->U.assign(U, U + (size_t)n_tau * n_s);
    ctx->S.assign(S, S + n_s);
    ctx->V.assign(V, V + (size_t)n_omega * n_s);
    mxe_opts_default(&ctx->opts);
    if (device >= 0 && device < MXE_MAX_DEVICES) {      // (the staging buffers of d2h_pipelined: once per device -- the current one is this context's)
        std::lock_guard<std::mutex> lk(g_pinned_dev[device].mu); g_pinned_dev[device].ready();
    }
    *out = ctx;
    return MXE_OK;
}
MXE_CATCH_ALL

static void comm_release(mxe_ctx* ctx);

void mxe_ctx_destroy(mxe_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);      // (the blocks go back to the pool: nothing in flight may use them)
    comm_release(ctx);
    if (ctx->h_sel3_pinned) { (void)hipHostFree(ctx->h_sel3_pinned); ctx->h_sel3_pinned = nullptr; ctx->h_sel3_pinned_n = 0; }
    ctx->dproj_U.release(); ctx->dproj_err.release(); ctx->dproj_G.release(); ctx->dproj_off.release();
    ctx->dVx.release(); ctx->dsel3.release(); ctx->dgstate.release(); ctx->dgstate_mc.release(); ctx->dfin_elem.release(); ctx->dfin_prob0.release();
    ctx->dfin_len.release(); ctx->dfin_v0.release(); ctx->dfin_start.release();
    ctx->dfin_budget.release(); ctx->dfin_out.release(); ctx->dfin_alpha.release();
    ctx->dlogdet.release(); ctx->dparent_elem.release(); ctx->dV.release(); ctx->dVt.release(); ctx->dVf.release(); ctx->dVtf.release(); ctx->dc.release(); ctx->dcinv.release();
    ctx->dghat.release(); ctx->dcperp.release(); ctx->dD.release(); ctx->dsumD.release();
    ctx->dalpha.release(); ctx->dv0.release(); ctx->delem_ds.release(); ctx->delem_kind.release();
    ctx->dchain_elem.release(); ctx->dsub_prob0.release(); ctx->dsub_len.release(); ctx->dsub_v0.release(); ctx->dwg_chains.release(); ctx->dqueue.release(); ctx->dcounter.release(); ctx->dsub_pre.release(); ctx->dsub_init.release(); ctx->dsub_walk0.release(); ctx->dwalk_alpha.release(); ctx->dinit_tab.release(); ctx->dout_v.release(); ctx->dout_pack.release(); ctx->dout_pack2.release();
    ctx->dout_niter.release(); ctx->dout_conv.p = ctx->dout_nevals.p = nullptr; ctx->dout_nact.release(); ctx->dexcluded.release();
    ctx->dB.release(); ctx->dA.release(); ctx->dprof.release();
    ctx->rows_out.release(); ctx->rows_idx.release();
    ctx->ev_x.release(); ctx->ev_alpha.release(); ctx->ev_scal.release(); ctx->ev_vecw.release(); ctx->ev_vecs.release();
    ctx->ev_mat.release(); ctx->ev_elem.release();
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->ev_mark) hipEventDestroy(ctx->ev_mark);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* mxe_last_hip_error(mxe_ctx* ctx) { return ctx ? ctx->hip_err.c_str() : ""; }

int mxe_dataset_add(mxe_ctx* ctx, int n_rows, const double* U_rot, const double* err, int* id)
try {
    if (!ctx || !err || n_rows < 1) return MXE_ERR_ARG;
    if (!U_rot && (ctx->U.empty() || n_rows != ctx->n_tau)) return MXE_ERR_ARG;
    DataSet d;
    int rc = build_dataset(ctx, n_rows, U_rot, err, d);
    if (rc != MXE_OK) return rc;
    ctx->ds.push_back(std::move(d));
    ctx->ds_dirty = true; ctx->proj_ready = false;
    ctx->chains_ready = false;
    if (id) *id = (int)ctx->ds.size() - 1;
    return MXE_OK;
}
MXE_CATCH_ALL

int mxe_dataset_clear(mxe_ctx* ctx)
{
    if (!ctx) return MXE_ERR_ARG;
    ctx->ds.clear(); ctx->ds_dirty = true; ctx->proj_ready = false; ctx->n_elem = 0; ctx->chains_ready = false;
    return MXE_OK;
}

namespace {
// ghat = Uhat^T (G / err) and c_perp = |G / err - Uhat ghat|^2 of elements [e0, e1): the data side of the whitened problem
// (DESIGN.md section 2).  Both along the rows of Uhat (unit stride); the residual is formed term by term (a sum of squares: no
// cancellation).  A batch of 256 elements is 5.7 M multiply-adds: on one host core 0.5-0.9 ms -- more than the kernel that solves
// them takes --, so batches are cut over a handful of threads (the arithmetic of an element does not depend on the cut).
void project_elements(const mxe_ctx* ctx, int e0, int e1, const int32_t* dataset_of_elem, const double* G, const int64_t* G_offset,
                      double* hghat, double* hcperp)
{
    const int ns = ctx->n_s, NP = ctx->NP;
    std::vector<double> Gt;
    for (int e = e0; e < e1; ++e) {
        const DataSet& DS = ctx->ds[dataset_of_elem[e]];
        const double* Ge = G + G_offset[e];
        Gt.assign(DS.n_rows, 0.0);
        for (int i = 0; i < DS.n_rows; ++i) Gt[i] = Ge[i] / DS.err[i];
        double* gh = hghat + (size_t)e * NP;
        for (int i = 0; i < DS.n_rows; ++i) {
            const double* ui = DS.Uhat.data() + (size_t)i * ns;
            const double gi = Gt[i];
            for (int k = 0; k < ns; ++k) gh[k] += ui[k] * gi;
        }
        double cp = 0.0;
        for (int i = 0; i < DS.n_rows; ++i) {
            const double* ui = DS.Uhat.data() + (size_t)i * ns;
            double r = Gt[i];
            for (int k = 0; k < ns; ++k) r -= ui[k] * gh[k];
            cp += r * r;
        }
        hcperp[e] = cp;
    }
}

void project_all(const mxe_ctx* ctx, int n_elem, const int32_t* dataset_of_elem, const double* G, const int64_t* G_offset,
                 double* hghat, double* hcperp)
{
    int nt = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::min(nt, n_elem / 32);
    if (nt <= 1) { project_elements(ctx, 0, n_elem, dataset_of_elem, G, G_offset, hghat, hcperp); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) {
        const int e0 = (int)((long long)n_elem * t / nt), e1 = (int)((long long)n_elem * (t + 1) / nt);
        pool.emplace_back(project_elements, ctx, e0, e1, dataset_of_elem, G, G_offset, hghat, hcperp);
    }
    for (auto& th : pool) th.join();
}
}

int mxe_elements_set(mxe_ctx* ctx, int n_elem, const int32_t* dataset_of_elem,
                     const double* G, const int64_t* G_offset,
                     const double* D, const int32_t* entropy)
try {
    if (!ctx || n_elem < 1 || !dataset_of_elem || !G || !G_offset || !D || !entropy) return MXE_ERR_ARG;
    if (ctx->ds.empty()) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int ns = ctx->n_s, NP = ctx->NP, nw = ctx->n_omega, nwp = ctx->nwp;
    std::vector<double> hghat((size_t)n_elem * NP, 0.0), hcperp(n_elem), hD((size_t)n_elem * nwp, 0.0), hsumD(n_elem);
    ctx->elem_ds.assign(n_elem, 0); ctx->elem_kind.assign(n_elem, 0);
    for (int e = 0; e < n_elem; ++e) {
        const int d = dataset_of_elem[e];
        if (d < 0 || d >= (int)ctx->ds.size()) return MXE_ERR_ARG;
        if (entropy[e] != MXE_ENTROPY_NORMAL && entropy[e] != MXE_ENTROPY_PLUSMINUS) return MXE_ERR_ARG;
        double sd = 0.0;
        for (int i = 0; i < nw; ++i) { hD[(size_t)e * nwp + i] = D[(size_t)e * nw + i]; sd += D[(size_t)e * nw + i]; }
        hsumD[e] = (entropy[e] == MXE_ENTROPY_PLUSMINUS) ? 2.0 * sd : sd;
        ctx->elem_ds[e] = d; ctx->elem_kind[e] = entropy[e];
    }
    project_all(ctx, n_elem, dataset_of_elem, G, G_offset, hghat.data(), hcperp.data());
    ctx->n_elem = n_elem;
    ctx->h_sumD = hsumD;
    ctx->h_D = hD;
    HIPCHK(ctx, ctx->dghat.ensure(hghat.size()));
    HIPCHK(ctx, ctx->dcperp.ensure(n_elem));
    HIPCHK(ctx, ctx->dD.ensure(hD.size()));
    HIPCHK(ctx, ctx->dsumD.ensure(n_elem));
    HIPCHK(ctx, ctx->delem_ds.ensure(n_elem));
    HIPCHK(ctx, ctx->delem_kind.ensure(n_elem));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dghat.p, hghat.data(), hghat.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dcperp.p, hcperp.data(), (size_t)n_elem * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dD.p, hD.data(), hD.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dsumD.p, hsumD.data(), (size_t)n_elem * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->delem_ds.p, ctx->elem_ds.data(), (size_t)n_elem * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->delem_kind.p, ctx->elem_kind.data(), (size_t)n_elem * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    if (ctx->ds_dirty) { int rc = upload_bases(ctx); if (rc != MXE_OK) return rc; }
    ctx->chains_ready = false;
    return MXE_OK;
}
MXE_CATCH_ALL

namespace {
// project_elements for one element per workgroup, the same operations in the same order with every product and sum rounded on
// its own (no contraction into fused multiply-adds: the host code has none either), so that new data staged this way give
// the bits mxe_elements_set gives: ghat_k = sum_i Uhat_ik G_i / err_i summed over i in turn, the residual row by row, its
// squares summed in turn.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256)
void project_kernel(const double* __restrict__ G, const int* __restrict__ elem_ds, const long long* __restrict__ ds_off,
                    const double* __restrict__ U, const double* __restrict__ err, int ns, int NP,
                    double* __restrict__ ghat, double* __restrict__ cperp, int rows_max, const long long* __restrict__ G_off)
{
    extern __shared__ double psh[];
    double* Gt = psh; double* gh = psh + rows_max; double* r2 = gh + NP;
    const int e = blockIdx.x, tid = threadIdx.x;
    const long long* o = ds_off + 3 * (size_t)elem_ds[e];
    const double* Ud = U + o[0]; const double* ed = err + o[1]; const int rows = (int)o[2];
    const double* Ge = G + G_off[e];
    for (int i = tid; i < rows; i += 256) Gt[i] = Ge[i] / ed[i];
    __syncthreads();
    for (int k = tid; k < NP; k += 256) {
        double acc = 0.0;
        if (k < ns) for (int i = 0; i < rows; ++i) { const double pr = Ud[(size_t)i * ns + k] * Gt[i]; acc = acc + pr; }
        gh[k] = acc;
        ghat[(size_t)e * NP + k] = acc;
    }
    __syncthreads();
    for (int i = tid; i < rows; i += 256) {
        double r = Gt[i];
        for (int k = 0; k < ns; ++k) { const double pr = Ud[(size_t)i * ns + k] * gh[k]; r = r - pr; }
        r2[i] = r * r;
    }
    __syncthreads();
    if (tid == 0) {
        double cp = 0.0;
        for (int i = 0; i < rows; ++i) cp = cp + r2[i];
        cperp[e] = cp;
    }
}
#pragma clang fp contract(fast)
}

int mxe_elements_update_data(mxe_ctx* ctx, int n_elem, const double* G, const int64_t* G_offset)
try {
    if (!ctx || !G || !G_offset) return MXE_ERR_ARG;
    if (ctx->n_elem < 1 || ctx->ds.empty() || ctx->ds_dirty) return MXE_ERR_STATE;
    if (n_elem != ctx->n_elem) return MXE_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int ns = ctx->n_s, NP = ctx->NP;
    // the data of the elements as ONE span of the caller's array (what set_elements of maxent_amd.device builds; anything else
    // goes through the host projection)
    long long lo = 0, hi = 0; int rows_max = 0; bool span = true;
    for (int e = 0; e < n_elem && span; ++e) {
        const int rows = ctx->ds[ctx->elem_ds[e]].n_rows;
        rows_max = std::max(rows_max, rows);
        if (G_offset[e] < 0) span = false;
        if (e == 0) { lo = G_offset[e]; hi = lo + rows; }
        else { lo = std::min<long long>(lo, G_offset[e]); hi = std::max<long long>(hi, G_offset[e] + rows); }
    }
    size_t total_rows = 0;
    for (int e = 0; e < n_elem; ++e) total_rows += ctx->ds[ctx->elem_ds[e]].n_rows;
    const size_t lds = ((size_t)2 * rows_max + NP) * 8;
    if (span && (size_t)(hi - lo) <= 2 * total_rows && lds <= 64 * 1024 && !getenv("MXE_HOST_PROJECTION")) {
        if (!ctx->proj_ready) {
            std::vector<double> hU, herr; std::vector<long long> off;
            for (const DataSet& D : ctx->ds) {
                off.push_back((long long)hU.size()); off.push_back((long long)herr.size()); off.push_back(D.n_rows);
                hU.insert(hU.end(), D.Uhat.begin(), D.Uhat.end());
                herr.insert(herr.end(), D.err.begin(), D.err.end());
            }
            HIPCHK(ctx, ctx->dproj_U.ensure(hU.size()));
            HIPCHK(ctx, ctx->dproj_err.ensure(herr.size()));
            HIPCHK(ctx, ctx->dproj_off.ensure(off.size()));
            HIPCHK(ctx, hipMemcpyAsync(ctx->dproj_U.p, hU.data(), hU.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->dproj_err.p, herr.data(), herr.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->dproj_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, stream_wait(ctx->stream));
            ctx->proj_ready = true;
        }
        std::vector<long long> goff(n_elem);
        for (int e = 0; e < n_elem; ++e) goff[e] = G_offset[e] - lo;
        HIPCHK(ctx, ctx->dproj_G.ensure((size_t)(hi - lo) + (size_t)n_elem));
        // (offsets behind the data in the same allocation: two copies, one wait)
        long long* dgoff = (long long*)(ctx->dproj_G.p + (hi - lo));
        HIPCHK(ctx, hipMemcpyAsync(ctx->dproj_G.p, G + lo, (size_t)(hi - lo) * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(dgoff, goff.data(), (size_t)n_elem * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(project_kernel, dim3(n_elem), dim3(256), lds, ctx->stream, ctx->dproj_G.p, ctx->delem_ds.p, ctx->dproj_off.p,
                           ctx->dproj_U.p, ctx->dproj_err.p, ns, NP, ctx->dghat.p, ctx->dcperp.p, rows_max, dgoff);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, stream_wait(ctx->stream));       // (goff is the caller-side vector of this call; G the caller's array)
        return MXE_OK;
    }
    std::vector<double> hghat((size_t)n_elem * NP, 0.0), hcperp(n_elem);
    project_all(ctx, n_elem, ctx->elem_ds.data(), G, G_offset, hghat.data(), hcperp.data());    // (as mxe_elements_set: the same bits)
    // (behind whatever the stream still runs with the old data)
    HIPCHK(ctx, hipMemcpyAsync(ctx->dghat.p, hghat.data(), hghat.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dcperp.p, hcperp.data(), (size_t)n_elem * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    return MXE_OK;
}
MXE_CATCH_ALL

namespace {
int launch_eval(mxe_ctx* ctx, const mxe::EvalParams& ep, size_t P);
void eval_params_base(mxe_ctx* ctx, mxe::EvalParams& ep);

// Start states of the lock-step kernel (KParams::init_tab): every piece of a scan starts from the scan's v0, and
// scans of one class -- same data set, entropy, default model, v0 -- share what the evaluation of that vector
// gives: H, S, h = V^T H, the Gram matrix.  They are evaluated ONCE per class with the device's own evaluation
// kernel (eval_kernel, the one behind mxe_eval_batch) when the chains are uploaded, and a piece begins with a
// Newton step instead of the evaluation of its start vector (one round of the kernel per piece).  At most
// eight classes; scans beyond that start the old way.
int build_init_table(mxe_ctx* ctx, int n_chain, const int32_t* elem_of_chain, const std::vector<double>& hv0)
{
    ctx->has_init = false;
    const int NP = ctx->NP, nw = ctx->n_omega, nwp = ctx->nwp;
    if (NP != 64 || ctx->mc_na != 32) return MXE_OK;
    std::vector<int> rep, class_of(n_chain, -1);
    for (int c = 0; c < n_chain; ++c) {
        const int e = elem_of_chain[c];
        int k = -1;
        for (size_t q = 0; q < rep.size() && k < 0; ++q) {
            const int c0 = rep[q], e0 = elem_of_chain[c0];
            if (ctx->elem_kind[e0] == ctx->elem_kind[e] && ctx->elem_ds[e0] == ctx->elem_ds[e] &&
                std::memcmp(&hv0[(size_t)c0 * NP], &hv0[(size_t)c * NP], (size_t)NP * 8) == 0 &&
                (e0 == e || std::memcmp(&ctx->h_D[(size_t)e0 * nwp], &ctx->h_D[(size_t)e * nwp], (size_t)nwp * 8) == 0)) k = (int)q;
        }
        if (k < 0 && rep.size() < 8) { k = (int)rep.size(); rep.push_back(c); }
        class_of[c] = k;
    }
    const int P = (int)rep.size();
    if (P == 0) return MXE_OK;
    std::vector<double> hx((size_t)P * NP), halpha(P, 1.0);
    std::vector<int> helem(P);
    for (int q = 0; q < P; ++q) {
        std::copy(hv0.begin() + (size_t)rep[q] * NP, hv0.begin() + (size_t)(rep[q] + 1) * NP, hx.begin() + (size_t)q * NP);
        helem[q] = elem_of_chain[rep[q]];
    }
    HIPCHK(ctx, ctx->ev_x.ensure(hx.size()));
    HIPCHK(ctx, ctx->ev_alpha.ensure(P));
    HIPCHK(ctx, ctx->ev_elem.ensure(P));
    HIPCHK(ctx, ctx->ev_scal.ensure((size_t)5 * P));
    HIPCHK(ctx, ctx->ev_vecw.ensure((size_t)4 * P * nw));
    HIPCHK(ctx, ctx->ev_vecs.ensure((size_t)2 * P * NP));
    HIPCHK(ctx, ctx->ev_mat.ensure((size_t)P * NP * NP));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ev_x.p, hx.data(), hx.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ev_alpha.p, halpha.data(), (size_t)P * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ev_elem.p, helem.data(), (size_t)P * 4, hipMemcpyHostToDevice, ctx->stream));
    mxe::EvalParams ep;
    eval_params_base(ctx, ep);
    ep.elem = ctx->ev_elem.p; ep.alpha = ctx->ev_alpha.p; ep.x = ctx->ev_x.p; ep.x_stride = NP;
    ep.input_is_H = 0; ep.eta = 1.0; ep.want_gram = 1;
    ep.Q = ctx->ev_scal.p; ep.chi2 = ep.Q + P; ep.S = ep.chi2 + P;
    ep.H = ctx->ev_vecw.p; ep.u = ep.H + (size_t)P * nw; ep.w = ep.u + (size_t)P * nw;
    ep.h = ctx->ev_vecs.p; ep.g = ep.h + (size_t)P * NP;
    ep.W = ctx->ev_mat.p;
    int rc = launch_eval(ctx, ep, (size_t)P);
    if (rc != MXE_OK) return rc;
    std::vector<double> hS(P), hH((size_t)P * nw), hw((size_t)P * nw), hh((size_t)P * NP), hW((size_t)P * NP * NP);
    HIPCHK(ctx, hipMemcpyAsync(hS.data(), ep.S, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hH.data(), ep.H, hH.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hw.data(), ep.w, hw.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hh.data(), ep.h, hh.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hW.data(), ep.W, hW.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    const int STR = mxe::MC_INIT_STRIDE;
    std::vector<double> tab((size_t)P * STR, 0.0);
    for (int q = 0; q < P; ++q) {
        double Hn2 = 0.0, wmax = 0.0;
        for (int i = 0; i < nw; ++i) { Hn2 += hH[(size_t)q * nw + i] * hH[(size_t)q * nw + i]; wmax = std::max(wmax, hw[(size_t)q * nw + i]); }
        if (!(wmax > 1e-290 && wmax < 1e290) || !std::isfinite(Hn2) || !std::isfinite(hS[q])) {       // nothing to tabulate: these scans start the old way
            for (int c = 0; c < n_chain; ++c) if (class_of[c] == q) class_of[c] = -1;
            continue;
        }
        const double sc2 = std::ldexp(1.0, 8 - std::ilogb(wmax));       // as the kernel scales the operands of its Gram tiles
        double* T = &tab[(size_t)q * STR];
        const double* W = &hW[(size_t)q * NP * NP];
        int pr = 0;
        for (int mt = 0; mt < 2; ++mt)
            for (int nt = mt; nt < 2; ++nt, ++pr)
                for (int r = 0; r < 4; ++r)
                    for (int l = 0; l < 64; ++l) {
                        const int a = 16 * mt + 4 * (l >> 4) + r, b = 16 * nt + (l & 15);      // accumulator layout of the 16 x 16 tile
                        T[(pr * 4 + r) * 64 + l] = sc2 * W[(size_t)a * NP + b];
                    }
        for (int k = 0; k < NP; ++k) T[3 * 256 + k] = hh[(size_t)q * NP + k];
        T[3 * 256 + NP + 0] = hS[q]; T[3 * 256 + NP + 1] = Hn2; T[3 * 256 + NP + 2] = wmax; T[3 * 256 + NP + 3] = sc2;
    }
    std::vector<int> sub_init(std::max(ctx->n_sub, 1), -1);
    bool any = false;
    for (int sc = 0; sc < ctx->n_sub; ++sc) { sub_init[sc] = class_of[ctx->sub_v0[sc]]; any = any || sub_init[sc] >= 0; }
    if (!any) return MXE_OK;
    HIPCHK(ctx, ctx->dinit_tab.ensure(tab.size()));
    HIPCHK(ctx, ctx->dsub_init.ensure(sub_init.size()));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dinit_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dsub_init.p, sub_init.data(), sub_init.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    ctx->has_init = true;
    return MXE_OK;
}
// Which workgroups of a launch of two per CU share a CU?  The solo schedule of the lock-step kernel (the longest pieces get a
// CU to themselves: workgroup b's partner b + gridDim / 2 leaves at once) rests on an OBSERVED placement -- the dispatcher fills
// every CU with one workgroup before any gets its second, in order -- that HIP does not promise.  Probed once per device with
// the launch shape of that kernel (256 threads, LDS for exactly two workgroups per CU, every workgroup resident at once):
// every workgroup records XCC_ID / HW_ID; the rule holds when b and b + n_cu report the same CU for every b.  Where it does
// not, n_solo stays 0: the schedule is a speed-up only, results never depend on it.
__global__ __launch_bounds__(256)
void placement_probe_kernel(unsigned* out)
{
    extern __shared__ char probe_lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; probe_lds[0] = 1; }
    const long long t0 = clock64();
    while (clock64() - t0 < 60000) { }           // ~25 us: every workgroup of the grid is resident before the first one leaves
}
int g_placement[MXE_MAX_DEVICES];                // 0: not probed, 1: the rule holds, 2: it does not
std::mutex g_placement_mu;

int placement_rule_holds(mxe_ctx* ctx, int n_cu, bool* holds)
{
    *holds = false;
    if (getenv("MXE_FORCE_NO_SOLO_RULE")) return MXE_OK;              // (tests: the branch a different dispatcher would take)
    const int dev = ctx->device;
    if (dev < 0 || dev >= MXE_MAX_DEVICES) return MXE_OK;
    std::lock_guard<std::mutex> lk(g_placement_mu);
    if (g_placement[dev] == 0) {
        const int n = 2 * n_cu;
        DevBuf<unsigned> d;
        HIPCHK(ctx, d.ensure((size_t)2 * n));
        HIPCHK(ctx, hipFuncSetAttribute((const void*)placement_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
        std::vector<unsigned> h((size_t)2 * n);
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; ++rep) {          // (twice: the rule has to hold on every launch, not on the first)
            hipLaunchKernelGGL(placement_probe_kernel, dim3(n), dim3(256), 73728, ctx->stream, d.p);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(h.data(), d.p, h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, stream_wait(ctx->stream));
            for (int b = 0; b < n_cu && ok; ++b)
                ok = ((h[2 * b] >> 8) & 0xff) == ((h[2 * (b + n_cu)] >> 8) & 0xff) && (h[2 * b + 1] & 0xf) == (h[2 * (b + n_cu) + 1] & 0xf);      // (HW_ID: CU 11:8, SH 12, SE 15:13; XCC_ID 3:0)
        }
        d.release();
        g_placement[dev] = ok ? 1 : 2;
    }
    *holds = g_placement[dev] == 1;
    return MXE_OK;
}
}  // namespace

constexpr double MC_COUPLING_MAX = 1e-3;     // relative coupling of the first direction the 32-row active block leaves out (see below)

namespace {
// pieces of unequal length for launches that FILL the GPU (see mxe_chains_upload); MXE_TAPER overrides (A/B runs)
double piece_taper(const mxe_opts& o, int wgpc)
{
    if (const char* e = getenv("MXE_TAPER")) return std::max(0.2, atof(e));
    (void)o; (void)wgpc;
    return 1.0;
}
}

int mxe_chains_upload(mxe_ctx* ctx, int n_chain, int n_alpha,
                      const int32_t* elem_of_chain, const double* alpha_scaled,
                      const double* v0, const mxe_opts* opts)
try {
    if (!ctx || n_chain < 1 || n_alpha < 1 || !elem_of_chain || !alpha_scaled || !v0) return MXE_ERR_ARG;
    if (ctx->n_elem < 1) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->ds_dirty) { int rc = upload_bases(ctx); if (rc != MXE_OK) return rc; }
    const int ns = ctx->n_s, NP = ctx->NP, nw = ctx->n_omega;
    if (opts) ctx->opts = *opts; else mxe_opts_default(&ctx->opts);
    const mxe_opts& o = ctx->opts;
    if (o.maxiter < 1 || o.step_max <= 0 || o.mu_first <= 0 || o.mu_grow <= 1.0 || o.decouple_tol < 0) return MXE_ERR_ARG;
    if (o.waves_per_chain != 0 && o.waves_per_chain != 1 && o.waves_per_chain != 2 &&
        o.waves_per_chain != 4 && o.waves_per_chain != 8) return MXE_ERR_ARG;
    if (o.chains_per_wg != 0 && o.chains_per_wg != 1 && o.chains_per_wg != 4) return MXE_ERR_ARG;
    if (o.alpha_split < 0) return MXE_ERR_ARG;
    if (o.wg_per_cu < 0 || o.wg_per_cu > 2) return MXE_ERR_ARG;
    if (o.in_flight < 0 || o.in_flight > 64) return MXE_ERR_ARG;
    if (o.precision != MXE_PRECISION_F64 && o.precision != MXE_PRECISION_F32) return MXE_ERR_ARG;
    if (o.precision == MXE_PRECISION_F32 && NP != 64) return MXE_ERR_LIMIT;
    if (!(o.chi2_factor > 0.0) || !std::isfinite(o.chi2_factor)) return MXE_ERR_ARG;
    // Q = eta chi2 / 2 - alpha S has the minimiser of chi2 / 2 - (alpha / eta) S: the device iterates on
    // alpha / eta and the fetch multiplies Q by eta (cost_function.py:60, bryan_cost_function.py:71)
    ctx->chi2_factor = o.chi2_factor;
    std::vector<double> alpha_dev(alpha_scaled, alpha_scaled + (size_t)n_chain * n_alpha);
    if (o.chi2_factor != 1.0) for (double& a : alpha_dev) a /= o.chi2_factor;
    ctx->chain_elem.assign(elem_of_chain, elem_of_chain + n_chain);
    std::vector<double> hv0((size_t)n_chain * NP, 0.0);
    for (int c = 0; c < n_chain; ++c) {
        const int e = elem_of_chain[c];
        if (e < 0 || e >= ctx->n_elem) return MXE_ERR_ARG;
        const DataSet& DS = ctx->ds[ctx->elem_ds[e]];
        for (int k = 0; k < ns; ++k) {
            double s;
            if (DS.identity_q) s = v0[(size_t)c * ns + k];
            else {   // v' = Q^T v
                s = 0.0;
                for (int j = 0; j < ns; ++j) s += DS.Q[(size_t)j * ns + k] * v0[(size_t)c * ns + j];
            }
            hv0[(size_t)c * NP + k] = s;
        }
    }
    for (size_t i = 0; i < (size_t)n_chain * n_alpha; ++i)
        if (!(alpha_scaled[i] > 0.0) || !std::isfinite(alpha_scaled[i])) return MXE_ERR_ARG;
    const size_t P = (size_t)n_chain * n_alpha;
    // chain_kernel_lv (V^T resident in LDS as binary32): possible where the basis fits beside the state of four slots
    const bool lv_fits = NP == 64 && ctx->nwp <= 512 && o.lds_basis != 2 && !getenv("MXE_NO_LDS_BASIS") &&
                         o.chains_per_wg != 1 && o.tol_d <= 0.0 && o.decouple_tol > 0.0 &&
                         mxe::lv_lds_bytes(ns, ctx->nwp) + mxe::LV_STATIC_LDS <= (size_t)160 * 1024;
    // (binary32 launches in that kernel are scheduled like binary64 ones: lock-step pieces, one workgroup per CU)
    bool f32_lv = o.precision == MXE_PRECISION_F32 && lv_fits;
    if (f32_lv && ns > 32) {
        // (chain_kernel_lv has the plain 32-row build only: a job with an alpha that couples more than 32 directions -- the criterion
        //  of the layout decision below, per scan -- keeps the one-chain binary32 kernel AND its pieces of six alphas)
        for (int c = 0; c < n_chain && f32_lv; ++c) {
            const int e = elem_of_chain[c];
            if (e < 0 || e >= ctx->n_elem) return MXE_ERR_ARG;
            const DataSet& DS = ctx->ds[ctx->elem_ds[e]];
            double amin = 1e300;
            for (int i = 0; i < n_alpha; ++i) amin = std::min(amin, alpha_dev[(size_t)c * n_alpha + i]);
            if (!(DS.c[32] * DS.c[32] * std::max(1.0, ctx->h_sumD[e]) / amin <= MC_COUPLING_MAX)) f32_lv = false;
        }
        if (!f32_lv) {
            // Binary32 is asked for as the cheaper arithmetic; for such a job the cheaper arithmetic is the binary64 lock-step
            // build with the 64-row block (and the hand-over of what it leaves): the one-chain binary32 kernel took 0.7-1.5 s
            // where that takes 4-5 ms, and stops at its rounding floor besides (STRESS_F32=1 tools/stress.py, cases 24 / 25:
            // profiles/r04_e_stress_f32.txt).  The launch is promoted; mxe_last_launch_info names the kernel that ran.
            ctx->opts.precision = MXE_PRECISION_F64;
        }
    }
    if (o.precision == MXE_PRECISION_F32 && !lv_fits && NP == 64 && o.lds_basis != 2 && !getenv("MXE_NO_LDS_BASIS") &&
        o.chains_per_wg != 1 && o.tol_d <= 0.0 && o.decouple_tol > 0.0) {
        // A frequency mesh whose basis does not fit the LDS as binary32 (n_omega > 512, or n_s x (n_omega_pad + 4) floats beyond what
        // the slots leave): the binary32 request would run one chain per workgroup with V streamed from the L2 by every chain --
        // 3.6-7.8 ms where the binary64 lock-step kernel takes 0.5-1.7 (8 x 8 and 16 x 16 elements x 100 alpha at n_omega = 640 ...
        // 1500), and stops at its rounding floor besides (audit 8e-4 against 1e-8).  Binary32 is asked for as the cheaper
        // arithmetic: the launch is promoted like the two cases above.  lds_basis = 2 or chains_per_wg = 1 keep the one-chain
        // binary32 kernel (BASELINE config 5's tolerance sweep on such a mesh asks for it that way).
        ctx->opts.precision = MXE_PRECISION_F64;
    }
    ctx->lv_mode = 0;
    // ---- (sub-)chains: an alpha scan may be cut into pieces that are cold-started
    //      from the same v0 (the minimiser of each alpha does not depend on the path)
    int split = o.alpha_split;
    int split_pm = 0;                 // pieces per plus-minus scan where that differs from the normal-entropy scans' (0: the same)
    bool cut_by_cost = false;         // launches that do not fill the GPU: pieces of equal COST, one per slot (see below)
    bool cost_needs_short_pm = false; //   at two workgroups per CU: only when the plus-minus pieces stay within the depth target too
    long long slots_by_cost = 0;
    if (split <= 0) {
        // about two pieces per chain slot of the GPU (CUs x workgroups per CU x 4 slots): the persistent grid
        // then balances (pieces have unequal costs and are handed out most expensive first), and a batch that
        // is small for the GPU -- one rank's shard of a job that is spread over several -- is cut into many
        // short cold-started pieces rather than left on a fraction of the CUs.  None shorter than two alphas
        // (a cold start costs 4-10 iterations, a warm alpha 2-3).  Measured, kernel time of 256 / 128 / 64 / 32
        // scans of 100 alphas (profiles/r02_d_shard_sweep.txt): 16 pieces per scan 1.28 / 1.09 / 0.93 / 1.54 ms,
        // 34: 2.95 / 0.84 / 0.73 / 0.62, 50: 3.27 / 0.87 / 0.67 / 0.57.
        hipDeviceProp_t prop;
        HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
        // (two workgroups per CU where the lock-step kernel has a build for it: n_omega_pad <= 512)
        const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (f32_lv && o.wg_per_cu == 0 && ctx->nwp <= 512) {
            // A binary32 launch runs in chain_kernel_lv at ONE workgroup per CU.  A batch that fills the GPU at two per CU (the test
            // of the loop below) is faster in the binary64 kernel that runs that way: the 25 600-problem batch 0.81 ms against
            // 1.24 ms -- binary32 is asked for as the cheaper arithmetic, and there it is not.  Such a launch is promoted like the
            // one that couples more than 32 directions; wg_per_cu = 1 keeps it in chain_kernel_lv.
            long long weight = 0;
            for (int c = 0; c < n_chain; ++c) weight += (ctx->elem_kind[elem_of_chain[c]] == MXE_ENTROPY_NORMAL) ? 2 : 1;
            const int want2 = (int)std::max(1LL, (2LL * 8 * n_cu) / std::max(1LL, weight));
            if (std::max(1, std::min(want2, n_alpha / 2)) >= want2) { f32_lv = false; ctx->opts.precision = MXE_PRECISION_F64; }
        }
        int wgpc_guess = (o.wg_per_cu != 1 && ctx->nwp <= 512 && NP == 64 && o.chains_per_wg != 1 && !f32_lv) ? 2 : 1;
        for (;;) {
            const int n_slots = 4 * wgpc_guess * n_cu;
            // A piece of a normal-entropy scan costs about twice one of a plus-minus scan of the same length (10-18
            // against 5 evaluations for the cold start, 3 against 2 per alpha): it counts twice, so that a slot gets
            // two plus-minus pieces or one normal piece, not three (cfg4: 15 pieces per scan instead of 16 -- of
            // 4096 pieces 14 % of the slots took a third --, 1.228 -> 1.171 ms; 14: 1.34 ms)
            long long weight = 0;
            for (int c = 0; c < n_chain; ++c) weight += (ctx->elem_kind[elem_of_chain[c]] == MXE_ENTROPY_NORMAL) ? 2 : 1;
            // (mxe_opts.in_flight = n: the caller keeps n such batches in flight -- each fills 1 / n of the slots, with 1 / n of
            //  the cold starts: 25 600 alpha-solves in 4 x 256 pieces of 25 alphas cost 0.65 ms side by side with three other
            //  batches, in 15 x 256 pieces 0.83 ms alone and 0.74 ms next to one other; profiles/r04_experiments.txt 10.)
            const long long nfl = std::max(1, o.in_flight);
            weight *= nfl;
            // (rounded UP when batches share the GPU: 4 pieces per scan x 4 batches 0.647 ms per batch, 3 x 4: 0.690)
            const int want = (int)std::max(1LL, (2LL * n_slots + (nfl > 1 ? std::max(1LL, weight) - 1 : 0)) / std::max(1LL, weight));
            split = std::max(1, std::min(want, n_alpha / 2));
            // A batch with more scans than that rule has pieces for (two pieces per slot would be fewer than six per scan: 32 x 32
            // elements and beyond) was left with one to four long pieces per scan, 1.1-2.3 per slot -- and a slot that takes one piece
            // more than its neighbours then runs half a launch longer: 48 x 48 x 100 alphas 9.6 ms, 23 M alpha-solves/s, where six
            // pieces per scan take 5.6 ms, 40.8 M (tools/batch_size_sweep.sh, profiles/r05_experiments.txt 11.).  There the count
            // is chosen by what it costs: the cold start of a piece, 4 evaluations against 2 per alpha of its length, and the
            // imbalance of a queue of pieces of one size, half a piece per slot.  (Batches in flight fill each other's gaps: fewest
            // pieces, as above.)
            if (nfl == 1 && want < 6 && n_alpha >= 16) {
                double best = 1e300; int bs = split;
                for (int sp = std::max(1, want); sp <= std::min(16, n_alpha / 4); ++sp) {
                    const double len = (double)n_alpha / sp, per_slot = (double)sp * (double)weight / n_slots;
                    const double loss = 4.0 / (4.0 + 2.0 * len) + 0.5 / per_slot;
                    if (loss < best) { best = loss; bs = sp; }
                }
                split = bs;
            }
            // (the binary32 streaming variant stops an alpha at its rounding floor, which a cold start reaches
            //  from further away: it keeps pieces of at least six alphas, at most 16 per scan)
            if (o.precision == MXE_PRECISION_F32 && !f32_lv) split = std::max(1, std::min(std::min(want, 16), n_alpha / 6));
            // two workgroups per CU pay when there is work for two rounds of them; a batch that cannot be cut
            // into that many pieces runs at one per CU, where a round of a workgroup takes 45 k instead of 73 k
            // cycles (the 3 200-problem shard of cfg4 / 8: 0.48 against 0.59 ms)
            if (wgpc_guess == 2 && o.wg_per_cu == 0 && split < want) { wgpc_guess = 1; continue; }
            break;
        }
        ctx->wgpc_auto = wgpc_guess;
        ctx->mc_wgpc_hint = wgpc_guess;
        // A launch that does not fill the GPU (one workgroup per CU: pieces at the cap of two alphas) is as long as its deepest slot,
        // and a slot that takes a second piece pays a second cold start.  Where the uniform cut gives more pieces than slots, the
        // plus-minus scans -- cold start 5 rounds against 12-16, 2 rounds per alpha against 2.75: a piece of twice the alphas costs
        // what a normal-entropy piece does -- are cut into fewer, longer pieces, so that every slot gets ONE piece (the 3 200-problem
        // shard of cfg4 / 8: 1 600 pieces on 1 024 slots -> 1 012; profiles/r04_experiments.txt).  MXE_NO_SPLIT_BY_KIND: the old cut
        // (one workgroup per CU only.  At two per CU -- the two-GPU shard of cfg4, whose plus-minus pieces would stay short enough --
        //  the cut by cost LOST: 0.565 -> 0.617 ms; two workgroups per CU are the throughput regime, profiles/r04_experiments.txt)
        if (wgpc_guess == 1 && o.wg_per_cu == 0 && !getenv("MXE_NO_SPLIT_BY_KIND") && n_alpha >= 4) {
            // (one workgroup per CU is not always a launch that does not fill the GPU -- a binary32 launch runs that way whatever its
            //  size --: the cut is only taken when the plus-minus pieces it leaves are short, see min_pm below.  Without that test
            //  the 25 600-problem batch in binary32 got ONE piece per plus-minus scan: 1.24 -> 3.18 ms)
            cut_by_cost = true; slots_by_cost = 4LL * n_cu; cost_needs_short_pm = true;
        }
        // a small batch that cannot fill the lock-step layout (>= 768 pieces) with pieces of six alphas,
        // but can with shorter ones, takes those: the lock-step kernel serves four pieces with the loads
        // and the time the one-chain kernel spends on one (cfg3, 16 scans: 1.9 ms with 256 pieces in the
        // one-chain layout, 0.7 ms with 768 pieces of two alphas in the lock-step layout)
        if ((long long)n_chain * split < 768 && n_alpha >= 4 && (long long)n_chain * (n_alpha / 2) >= 768 &&
            NP == 64 && o.chains_per_wg != 1 && o.tol_d <= 0.0 && o.decouple_tol > 0.0 && (o.precision == MXE_PRECISION_F64 || f32_lv))
            split = (768 + n_chain - 1) / n_chain;
    }
    if (split > n_alpha) split = n_alpha;
    ctx->sub_elem.clear(); ctx->sub_prob0.clear(); ctx->sub_len.clear(); ctx->sub_v0.clear();
    // Normal entropy: from the default model the smallest alphas of a scan are far away.  Measured on the BASELINE
    // batch (profiles/r02_f_cold_start_profile.txt), a cold start in the last 6 % of the logarithmic alpha range takes
    // 20-30 evaluations on average and 50-390 for single scans (above that range: 10-18, at most 21) -- and a launch
    // ends with its slowest piece.  A piece that starts there is led by the last alpha ABOVE the range: cold start
    // where it is cheap and safe, then one warm step down to the piece's first alpha (lock-step kernel: chain_pre);
    // in the other layouts, and where that step would be long, the piece is joined to the one before it.
    ctx->sub_pre.clear(); ctx->has_pre = false;
    ctx->sub_walk0.clear(); ctx->walk_alpha.clear(); ctx->has_walk = false;
    // Launches that do not fill the GPU (cut_by_cost): such a launch is as long as its deepest slot, so the pieces are cut to equal
    // COST and every slot gets one.  The cost of a normal-entropy piece is its cold start -- 9-10 evaluations in the upper third of
    // the logarithmic alpha range, rising to 17-19 just above the guarded tail (profiles/r04_b_depth_by_piece.txt; the same numbers
    // as the cold-start profile of r02) -- plus ~3 per further alpha: with the uniform pieces of two alphas the deepest slot of the
    // 8-GPU shards was a piece at alpha index 88-92 (19 + 5 evaluations), not the led tail pieces (~22 rounds with their walk).
    // Pieces of a normal-entropy scan therefore get as many alphas as fit MC_DEPTH_TARGET evaluations (4 at the top of the mesh, 1
    // next to the tail) and end where the guarded range begins; the plus-minus scans (cold start ~4.5, ~2.2 per alpha) share the
    // slots the normal-entropy pieces leave.
    // (the same cut for the normal-entropy scans of the batch that FILLS the GPU -- targets of 34 / 38 / 42 evaluations instead of 15
    //  uniform pieces -- was 10-13 % slower, 0.814 -> 0.894 / 0.893 / 0.917 ms: there the queue balances, profiles/r04_experiments.txt)
    constexpr double MC_DEPTH_TARGET = 20.0;
    auto scan_range = [&](const double* ac, double& lmax, double& lmin) {
        lmax = -1e300; lmin = 1e300;
        for (int i = 0; i < n_alpha; ++i) { const double l = std::log(ac[i]); lmax = std::max(lmax, l); lmin = std::min(lmin, l); }
    };
    auto cost_cuts = [&](const double* ac, std::vector<int>& cuts) {          // piece starts of one normal-entropy scan (+ n_alpha)
        double lmax, lmin;
        scan_range(ac, lmax, lmin);
        const double lguard = lmax - 0.94 * (lmax - lmin);
        int lead = -1;                               // the smallest alpha above the guarded range: what leads the tail pieces
        { double best = 1e300; for (int i = 0; i < n_alpha; ++i) if (std::log(ac[i]) >= lguard && ac[i] < best) { best = ac[i]; lead = i; } }
        cuts.clear();
        int a0 = 0;
        while (a0 < n_alpha) {
            cuts.push_back(a0);
            const double la = std::log(ac[a0]);
            if (lmax > lmin && la < lguard && lead >= 0 && lead < a0) break;     // (the guarded tail: one range, cut into led single alphas below)
            const double xpos = (lmax > lmin) ? (lmax - la) / (lmax - lmin) : 0.0;
            const double cold = 9.5 + 13.0 * std::max(0.0, xpos - 0.3);
            int L = 1 + (int)std::floor(std::max(0.0, (MC_DEPTH_TARGET - cold) / 3.0));
            L = std::max(1, std::min(L, 6));
            int a1 = std::min(n_alpha, a0 + L);
            for (int j = a0 + 1; j < a1; ++j) if (lmax > lmin && std::log(ac[j]) < lguard && lead >= 0 && lead < j) { a1 = j; break; }
            a0 = a1;
        }
        cuts.push_back(n_alpha);
    };
    // plus-minus scans: cold start ~4.5 evaluations, then 2 per alpha at the top of the mesh and 3 at its bottom -- with uniform pieces of
    // seven alphas the deepest slot of the four-GPU shards was a plus-minus piece at the smallest alphas (4.6 + 7 x 3.0 = 26 rounds).
    // At most `pieces` pieces of equal cost: the smallest cost per piece that needs no more (bisection)
    auto pm_cuts = [&](const double* ac, int pieces, std::vector<int>& cuts) {
        double lmax, lmin;
        scan_range(ac, lmax, lmin);
        auto w = [&](int a) { return 2.0 + ((lmax > lmin) ? (lmax - std::log(ac[a])) / (lmax - lmin) : 0.0); };
        auto cut = [&](double T, std::vector<int>* out) {
            int n = 0, a0 = 0;
            while (a0 < n_alpha) {
                if (out) out->push_back(a0);
                double cost = 4.5 + w(a0);
                int a1 = a0 + 1;
                while (a1 < n_alpha && cost + w(a1) <= T) { cost += w(a1); ++a1; }
                a0 = a1; ++n;
            }
            return n;
        };
        double lo = 6.0, hi = 4.5 + 3.0 * n_alpha + 1.0;
        for (int it = 0; it < 40 && hi - lo > 0.05; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (cut(mid, nullptr) <= pieces) hi = mid; else lo = mid;
        }
        cuts.clear();
        cut(hi, &cuts);
        cuts.push_back(n_alpha);
    };
    if (cut_by_cost) {
        long long n_normal = 0, pieces_normal = 0;
        std::vector<int> cuts;
        for (int c = 0; c < n_chain; ++c)
            if (ctx->elem_kind[elem_of_chain[c]] == MXE_ENTROPY_NORMAL) {
                ++n_normal;
                cost_cuts(alpha_dev.data() + (size_t)c * n_alpha, cuts);
                pieces_normal += (long long)cuts.size() - 2 + (n_alpha - cuts[cuts.size() - 2]);      // (the last range: one piece per alpha if it is the guarded tail -- an upper bound otherwise)
            }
        const long long n_pm = n_chain - n_normal;
        // (a plus-minus piece of eight or nine alphas costs ~4.5 + 8 x 2.5 = 24 evaluations: the depth of the led tail pieces.  The
        //  cut is for launches whose plus-minus pieces stay that short: the four-GPU shard of cfg4 -- 14 pieces per scan -- does,
        //  cfg4 itself does not)
        const long long min_pm = cost_needs_short_pm ? (n_alpha + 8) / 9 : 1;       // (at most nine alphas per plus-minus piece)
        if (pieces_normal + n_pm * min_pm <= slots_by_cost) {
            if (n_pm > 0) split_pm = (int)std::max(1LL, std::min<long long>(n_alpha / 2, (slots_by_cost - pieces_normal) / n_pm));
        } else cut_by_cost = false;                  // (more scans than slots can take one piece of each: the uniform cut)
    }
    constexpr double MC_LADDER_COARSE = 2.0;      // a step between neighbouring alphas beyond this factor is not taken in one go (measured: up to a factor ~1.5 a warm step is safe, over a factor 2 single scans took 100-300 evaluations; 1.6 here cost stress case 51 -- ratio 1.66, sigma 1e-5 -- two converged flags and 3 x the time)
    const double MC_LADDER_RATIO = getenv("MXE_LADDER_RATIO") ? std::max(1.05, atof(getenv("MXE_LADDER_RATIO"))) : 1.56;      // ratio of the rungs (just above MXE_X_WALK_RATIO: the walk lands on every rung)
    constexpr int MC_LADDER_MAX = 28;             // rungs per piece (the slot's alpha table holds 32 entries)
    const bool ladder_ok = !getenv("MXE_NO_LADDER") && NP == 64 && o.chains_per_wg != 1 && o.tol_d <= 0.0 && o.decouple_tol > 0.0 &&
                           (o.precision == MXE_PRECISION_F64 || f32_lv);
    for (int c = 0; c < n_chain; ++c) {
        const double* ac = alpha_dev.data() + (size_t)c * n_alpha;
        double pre_alpha = 0.0, lguard = 0.0;
        int pre_index = -1;
        const bool normal_c = ctx->elem_kind[elem_of_chain[c]] == MXE_ENTROPY_NORMAL;
        const double hard_below = 0.25 * ctx->ds[ctx->elem_ds[elem_of_chain[c]]].n_rows;      // alpha~ below N_data / 4
        if (normal_c && split > 1) {
            double lmax, lmin;
            scan_range(ac, lmax, lmin);
            lguard = lmax - 0.94 * (lmax - lmin);
            double best = 1e300;                     // the smallest alpha of the scan that is still above the guarded range
            for (int i = 0; i < n_alpha; ++i) if (std::log(ac[i]) >= lguard && ac[i] < best) { best = ac[i]; pre_index = i; }
            if (lmax > lmin && best < 1e300) pre_alpha = best;
        }
        std::vector<int> cuts;
        if (cut_by_cost && normal_c && split > 1) cost_cuts(ac, cuts);
        else if (cut_by_cost && !normal_c && split_pm > 1 && split_pm < n_alpha / 2) pm_cuts(ac, split_pm, cuts);      // (at the cap of two alphas there is nothing to balance)
        else {
            const int split_c = (split_pm > 0 && !normal_c) ? std::min(split_pm, n_alpha) : split;
            // (taper: the pieces of a scan grow from its first alpha to its last -- piece i of n has 1 + (taper - 1) i / (n - 1) parts --
            //  so that what the queue hands out LAST, the cheap short pieces at the top of the mesh, evens the workgroups out;
            //  1 = pieces of equal length)
            const double taper = piece_taper(o, ctx->mc_wgpc_hint);
            if (taper != 1.0 && split_c > 1) {
                std::vector<double> wsum(split_c + 1, 0.0);
                for (int i = 0; i < split_c; ++i) wsum[i + 1] = wsum[i] + 1.0 + (taper - 1.0) * i / (split_c - 1);
                for (int sidx = 0; sidx <= split_c; ++sidx) {
                    int a = (int)std::llround(n_alpha * wsum[sidx] / wsum[split_c]);
                    a = std::max(a, sidx == 0 ? 0 : cuts.back() + 1);          // (no empty piece)
                    a = std::min(a, n_alpha - (split_c - sidx));
                    if (sidx == split_c) a = n_alpha;
                    if (cuts.empty() || a > cuts.back()) cuts.push_back(a);
                }
            } else
            for (int sidx = 0; sidx <= split_c; ++sidx) {
                const int a = (int)((long long)n_alpha * sidx / split_c);
                if (cuts.empty() || a > cuts.back()) cuts.push_back(a);
            }
        }
        for (size_t sidx = 0; sidx + 1 < cuts.size(); ++sidx) {
            const int a0 = cuts[sidx], a1 = cuts[sidx + 1];
            if (a1 <= a0) continue;
            const bool guarded = pre_alpha > 0.0 && sidx > 0 && std::log(ac[a0]) < lguard && pre_index < a0;
            // (the warm step from the leading alpha is safe over a factor of 1.5 in alpha -- measured: at most 11
            //  evaluations; over a factor of 2 single scans took 100-300 --: deeper into the range, the piece
            //  before runs on instead)
            // (lock-step kernel: the piece WALKS from the leading alpha down the mesh to its first alpha with a loose
            //  tolerance -- every step as safe as the scan itself --, so every piece of the tail stands alone and the tail
            //  of a scan is as many short chains side by side as it has pieces.  A jump over more than a factor 1.5 in
            //  alpha -- single scans took 100-300 evaluations over a factor 2 -- was what joined pieces until r02_k)
            // (a led piece is cut into single alphas: each walks down from the leading alpha on its own, and the tail of
            //  the scan -- the longest chain of every launch that does not fill the GPU -- is as deep as ONE walk)
            // (a piece that starts above the range and runs into it stays whole: cutting it where it enters cost the
            //  batch that fills the GPU 6 % -- cfg4 on one GPU 0.947 -> 1.01 ms)
            const int g0 = guarded ? a0 : a1;
            auto emit = [&](int first, int len, int pre, int walk0) {
                ctx->sub_elem.push_back(elem_of_chain[c]);
                ctx->sub_prob0.push_back(c * n_alpha + first);
                ctx->sub_len.push_back(len);
                ctx->sub_v0.push_back(c);
                ctx->sub_pre.push_back(pre);
                ctx->sub_walk0.push_back(walk0);
                if (pre > 0) ctx->has_pre = true;
            };
            // A mesh too coarse to walk on (round 5).  The warm step into an alpha is safe over a factor ~1.5 in alpha; the reference's
            // own tests and defaults use 3-20 alphas over 4-6 decades (alpha_meshes.py:81, test/python/tau_maxent.py:44), and
            // where the entropy term no longer holds the solution (alpha~ below about N_data / 4) a step over a factor 2 ... 600 took
            // 250-2 300 evaluations (profiles/r05_b_coarse_mesh.txt: smoke()'s last alpha 913 of the launch's 308 rounds).  Such an
            // alpha is a piece of its own that starts cold where that is cheap -- at max(N_data / 4, its own alpha) -- and walks down a
            // LADDER of alphas of its own (ratio MC_LADDER_RATIO, a few loose rounds per rung, no records) to its alpha: all hard
            // alphas of a scan side by side, each as deep as one cold start + one walk.
            auto hard = [&](int i) {
                if (!ladder_ok || i < 0 || i >= n_alpha) return false;
                const double a = ac[i];
                if (!(a < hard_below)) return false;
                if (i == 0) return false;                  // (the head of a scan starts from the default model as ever)
                const double r = ac[i - 1] / a;
                return r > MC_LADDER_COARSE || r < 1.0 / MC_LADDER_COARSE;
            };
            auto emit_ladder = [&](int i, int len = 1) {
                const double a = ac[i];
                const double top = std::max(hard_below, a * MC_LADDER_RATIO);
                int rungs = (int)std::ceil(std::log(top / a) / std::log(MC_LADDER_RATIO) - 1e-9);
                rungs = std::max(1, std::min(rungs, std::min(MC_LADDER_MAX, 30 - len)));      // (rungs + alphas of the piece: the slot's table of 32)
                const double ratio = std::pow(top / a, 1.0 / rungs);          // (equal rungs; more than MC_LADDER_MAX would not fit the slot's table)
                const int w0 = (int)ctx->walk_alpha.size();
                for (int k = 0; k < rungs; ++k) ctx->walk_alpha.push_back(a * std::pow(ratio, rungs - k));
                emit(i, len, rungs, w0);
                ctx->has_walk = true;
            };
            // A piece whose FIRST alpha is its hardest: the head of a scan that begins deep in the hard region, and every piece of an
            // ASCENDING scan there (each starts from the default model at its smallest alpha: on 150 alphas rising from alpha~ = 0.5 at
            // sigma = 4e-5 the head took 2 989 evaluations and did not converge, tools/stress.py case 17).  It is led down a ladder from
            // N_data / 4 to its first alpha and goes on up its own mesh from there.
            auto emit_plain = [&](int first, int len) {
                const bool deep = ladder_ok && len <= 24 && ac[first] * (MC_LADDER_RATIO * MC_LADDER_RATIO) < hard_below &&
                                  (first == 0 || ac[first - 1] < ac[first]);
                if (deep) emit_ladder(first, len);
                else emit(first, len, 0, -1);
            };
            {
                // the part of the piece that is not led: cut at every hard alpha
                int b = a0;
                for (int i = a0; i < g0; ++i)
                    if (hard(i)) {
                        if (i > b) emit_plain(b, i - b);
                        emit_ladder(i);
                        b = i + 1;
                    }
                if (g0 > b) emit_plain(b, g0 - b);
            }
            for (int b0 = g0; b0 < a1; ++b0) {
                if (hard(b0)) emit_ladder(b0);
                else emit(b0, 1, b0 - pre_index, -1);
            }
        }
    }
    ctx->n_sub = (int)ctx->sub_elem.size();
    // ---- layout: four chains of one data set per workgroup wherever the lock-step kernel has a build for the
    //      problem -- also for a handful of pieces: its round (four chains) takes no longer than an iteration of the
    //      one-chain kernel (one), and a single scan of 100 alphas in 50 pieces runs in 0.45 ms against 0.94 ms
    //      (profiles/r02_k_small_batches.txt; until r02_j: only from 768 pieces on)
    int layout = o.chains_per_wg;
    ctx->mc_na = 0;
    if (layout == 0) layout = 4;
    if (layout == 4 && (NP != 64 || o.tol_d > 0.0 || o.decouple_tol <= 0.0 || (o.precision != MXE_PRECISION_F64 && !f32_lv))) layout = 1;
    if (layout == 4) {
        // capacity of the active block: the kernel clamps n_act to NA, and the
        // first neglected direction couples with relative strength
        // c_NA^2 wmax / alpha (wmax <= sum w ~ max(1, sum D)); accept NA when that
        // is below MC_COUPLING_MAX for every chain (inexact Newton: the contraction is that number, and the stopping
        // estimate of the kernel does not know about it -- an alpha stops when (e^{|du|} - 1 + theta) relH < tol_h, so its
        // last correction relH may be as large as tol_h / theta = 1e-4 and what the neglected direction leaves behind is
        // coupling x relH.  With 1e-2, the value until r03, converged alphas of launches AT that limit were 1.1e-6 ... 1.6e-6
        // from their fixed points (profiles/r03_i_small_sigma.txt); 1e-3 keeps a factor ten to the 1e-6 of the audit).
        double worst32 = 0.0, worst48 = 0.0;
        for (int sc = 0; sc < ctx->n_sub; ++sc) {
            const int e = ctx->sub_elem[sc];
            const DataSet& DS = ctx->ds[ctx->elem_ds[e]];
            double amin = 1e300;
            for (int i = 0; i < ctx->sub_len[sc]; ++i) amin = std::min(amin, alpha_dev[ctx->sub_prob0[sc] + i]);
            const double wbound = std::max(1.0, ctx->h_sumD[e]);
            if (ns > 32) worst32 = std::max(worst32, DS.c[32] * DS.c[32] * wbound / amin);
            if (ns > 48) worst48 = std::max(worst48, DS.c[48] * DS.c[48] * wbound / amin);
        }
        // (an active block of 48 in the lock-step kernel spilled registers in every tiling that was tried: problems
        //  that couple more than 32 directions run in the one-chain layout, whose solve lives in LDS)
        (void)worst48;
        ctx->excluded.clear();
        if (worst32 <= MC_COUPLING_MAX) ctx->mc_na = 32;
        else if (f32_lv) layout = 1;      // (chain_kernel_lv has the plain 32-row build only: the one-chain binary32 kernel, BEFORE any piece is cut or dropped below)
        else {
            // Some alphas couple more than 32 directions (very small error bars: sigma = 1e-6 on the BASELINE grids does at
            // the 27 smallest of 100 alphas).  Until r03 the whole launch then went to the one-chain layout (7 x slower).
            // (a) Plus-minus scans: the build with a 64-row active block -- ten Gram tiles per slot (80 KB of the LDS: one
            // workgroup per CU, n_omega_pad <= 512) and the one-row-per-lane elimination (gj1_solve_rows_f32).  240
            // off-diagonal scans x 100 alphas at sigma = 4e-6 ... 5e-7: 2.2 / 3.5 / 3.8 / 4.4 ms, nothing left over, audit
            // 5e-10 (one-chain layout: 12.8 ms).  (b) Normal-entropy scans: their systems at those alphas are ill conditioned
            // beyond what the binary16 Gram products of either lock-step build resolve (the iteration crawls to its limit
            // where the one-chain kernel, binary64 throughout, takes 3-28 steps): their pieces are cut where the criterion
            // fails -- coupling grows as alpha falls, so that is the tail of a scan -- and the alphas behind the cut are left
            // open for mxe_chains_finish: one warm chain per scan from the last alpha before the cut (records of such alphas
            // are NaN / not converged / 0 iterations until then: clear_excluded_kernel).  Without the 64-row build (a larger
            // frequency mesh) the plus-minus scans are cut as well.  Measured on the BASELINE batch (16 diagonal + 240
            // off-diagonal scans) with sigma = 4e-6 / 2e-6 / 1e-6 (maxiter 100): 15.4 / 17.1 / 146 ms in the one-chain
            // layout, 8-12 / 11-16 / 55-63 ms in every variant of this -- the serial depth of the 16 finishing chains (13-30
            // alphas x 3-28 iterations x 150-190 us in the one-chain kernel with 64 coupled directions) is the floor.
            // Leaving the cut alphas to the lock-step kernel's own give-up costs accuracy in the 32-row build (exact Newton
            // correction up to 9e-7, p99 1e-7, against 4e-8 / 2e-9) and time in the 64-row build (pieces of 10 alphas x 32
            // iterations: launch 8-9 ms).  Not when more than a third of the alphas would be left to the finishing pass
            // (sigma = 5e-7 without the 64-row build: 197 against 150 ms).  profiles/r03_c_cut_pieces.txt, r03_e_na64.txt
            const bool have64 = !getenv("MXE_NO_NA64") && ns > 32 && o.wg_per_cu != 2 && mc_lds_bytes(64, ctx->nwp, 1) <= 160 * 1024 - 6144;
            bool need64 = false;
            std::vector<char> bad(P, 0);
            size_t n_bad = 0;
            for (int c = 0; c < n_chain; ++c) {
                const int e = elem_of_chain[c];
                const DataSet& DS = ctx->ds[ctx->elem_ds[e]];
                const double lim = MC_COUPLING_MAX / (DS.c[32] * DS.c[32] * std::max(1.0, ctx->h_sumD[e]));    // alpha >= 1 / lim passes
                const bool to64 = have64 && ctx->elem_kind[e] != MXE_ENTROPY_NORMAL;
                for (int i = 0; i < n_alpha; ++i)
                    if (!(alpha_dev[(size_t)c * n_alpha + i] * lim >= 1.0)) {
                        if (to64) need64 = true;
                        else { bad[(size_t)c * n_alpha + i] = 1; ++n_bad; }
                    }
            }
            if (3 * n_bad > P) layout = 1;
            else {
                ctx->mc_na = need64 ? 64 : 32;
                std::vector<char> covered(P, 0);
                size_t w = 0;
                for (size_t sc = 0; sc < ctx->sub_elem.size(); ++sc) {
                    const int p0 = ctx->sub_prob0[sc];
                    int len = 0;
                    while (len < ctx->sub_len[sc] && !bad[(size_t)p0 + len]) ++len;
                    // (a led piece starts from an alpha above its own: larger, so it passes when the piece's does)
                    if (len == 0) continue;
                    for (int i = 0; i < len; ++i) covered[(size_t)p0 + i] = 1;
                    ctx->sub_elem[w] = ctx->sub_elem[sc]; ctx->sub_prob0[w] = p0; ctx->sub_len[w] = len;
                    ctx->sub_v0[w] = ctx->sub_v0[sc]; ctx->sub_pre[w] = ctx->sub_pre[sc]; ctx->sub_walk0[w] = ctx->sub_walk0[sc]; ++w;
                }
                if (w == 0) { ctx->mc_na = 0; layout = 1; }
                else {
                    ctx->sub_elem.resize(w); ctx->sub_prob0.resize(w); ctx->sub_len.resize(w); ctx->sub_v0.resize(w); ctx->sub_pre.resize(w); ctx->sub_walk0.resize(w);
                    ctx->n_sub = (int)w;
                    ctx->has_pre = false;
                    for (size_t sc = 0; sc < w; ++sc) ctx->has_pre = ctx->has_pre || ctx->sub_pre[sc] > 0;
                    for (size_t i = 0; i < P; ++i) if (!covered[i]) ctx->excluded.push_back((int)i);
                }
            }
        }
        if (layout == 4) {
            ctx->mc_wgpc = (o.wg_per_cu != 1 && (o.wg_per_cu == 2 || ctx->wgpc_auto == 2) && ctx->mc_na == 32 && ctx->nwp <= 512 &&
                            mc_lds_bytes(32, ctx->nwp, 2) <= 80 * 1024 - 2048) ? 2 : 1;
            ctx->mc_gst = false;
            if (mc_lds_bytes(ctx->mc_na, ctx->nwp, ctx->mc_wgpc) > 160 * 1024 - 6144) {
                // a frequency mesh whose state (u, H, sw of four slots: 80 B per omega) does not fit the LDS beside the
                // rest: the state goes to device memory (chain_kernel_mc<.., GSTATE>, one workgroup per CU)
                ctx->mc_wgpc = 1; ctx->mc_gst = true;
            }
        }
    }
    if (layout == 4 && f32_lv) {
        // the binary32 launch: only the plain 32-row layout with nothing cut has a build in chain_kernel_lv
        // (anything else was sent to the one-chain layout above, before the pieces were touched: r04's first form of this fell back
        //  HERE, after pieces of a 64-row launch had been cut -- the alphas behind the cuts were never solved, their records garbage:
        //  STRESS_F32=1 tools/stress.py, cases 24 / 25 / 42)
        if (ctx->mc_na == 32 && ctx->excluded.empty() && !ctx->mc_gst) { ctx->lv_mode = 1; ctx->mc_wgpc = 1; }
        else return MXE_ERR_STATE;
    }
    if (layout != 4 && ctx->has_pre) {
        size_t w = 0;
        for (size_t sc = 0; sc < ctx->sub_elem.size(); ++sc) {
            if (ctx->sub_pre[sc] > 0 && w > 0 && ctx->sub_v0[w - 1] == ctx->sub_v0[sc]) { ctx->sub_len[w - 1] += ctx->sub_len[sc]; continue; }
            ctx->sub_elem[w] = ctx->sub_elem[sc]; ctx->sub_prob0[w] = ctx->sub_prob0[sc]; ctx->sub_len[w] = ctx->sub_len[sc];
            ctx->sub_v0[w] = ctx->sub_v0[sc]; ++w;
        }
        ctx->sub_elem.resize(w); ctx->sub_prob0.resize(w); ctx->sub_len.resize(w); ctx->sub_v0.resize(w);
        ctx->sub_pre.assign(w, 0); ctx->has_pre = false;
        ctx->sub_walk0.assign(w, -1); ctx->has_walk = false;
        ctx->n_sub = (int)w;
    }
    ctx->wg_chains.clear(); ctx->queue.clear(); ctx->n_queue = 0; ctx->n_solo = 0; ctx->placement_checked = 0;
    if (layout == 4) {
        bool one_ds = true;
        for (int sc = 1; sc < ctx->n_sub; ++sc)
            if (ctx->elem_ds[ctx->sub_elem[sc]] != ctx->elem_ds[ctx->sub_elem[0]]) { one_ds = false; break; }
        if (one_ds) {
            // dynamic layout: a persistent grid takes pieces from a queue, most
            // expensive first (normal entropy and small alpha cost more iterations)
            std::vector<double> cost(ctx->n_sub);
            for (int sc = 0; sc < ctx->n_sub; ++sc) {
                const int e = ctx->sub_elem[sc];
                double amin = 1e300;
                for (int i = 0; i < ctx->sub_len[sc]; ++i) amin = std::min(amin, alpha_dev[ctx->sub_prob0[sc] + i]);
                cost[sc] = ctx->sub_len[sc] * (ctx->elem_kind[e] == MXE_ENTROPY_NORMAL ? 4.0 : 3.0) +
                           (ctx->elem_kind[e] == MXE_ENTROPY_NORMAL ? 16.0 : 6.0) - 1e-3 * std::log10(amin) +
                           (ctx->sub_walk0[sc] >= 0 ? 2.0 : 0.7) * ctx->sub_pre[sc];     // (the walk of a led piece: on the scan's mesh a landing every ~third alpha, MXE_X_WALK_RATIO; on a ladder every rung)
            }
            ctx->queue.resize(ctx->n_sub);
            for (int sc = 0; sc < ctx->n_sub; ++sc) ctx->queue[sc] = sc;
            std::stable_sort(ctx->queue.begin(), ctx->queue.end(), [&](int a, int b) { return cost[a] > cost[b]; });
            ctx->n_queue = ctx->n_sub;
            hipDeviceProp_t prop;
            HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
            const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            ctx->n_wg = std::min((ctx->n_sub + 3) / 4, n_cu * ctx->mc_wgpc);
            // The launch ends with its longest pieces: the last pieces of the normal-entropy scans (the most expensive
            // cold start, the most evaluations per alpha).  They are at the head of the queue; with two workgroups per
            // CU, the workgroups that take them get a CU to themselves -- workgroups b and b + n_wg / 2 share one
            // (tools/wg_placement.hip), the partners leave at once --, where a round takes 47 k instead of 69 k cycles
            // (cfg4, 16 such pieces in four workgroups: kernel 1.150 -> 1.123 ms; 16 workgroups 1.130, 64: 1.21).
            // The library's own schedule only.
            if (o.alpha_split == 0 && ctx->mc_wgpc == 2 && ctx->n_wg == 2 * n_cu) {
                int n_tail = 0;
                for (int sc = 0; sc < ctx->n_sub; ++sc)
                    if (ctx->elem_kind[ctx->sub_elem[sc]] == MXE_ENTROPY_NORMAL &&
                        ctx->sub_prob0[sc] + ctx->sub_len[sc] == (ctx->sub_v0[sc] + 1) * n_alpha) ++n_tail;
                ctx->n_solo = std::min((n_tail + 3) / 4, n_cu / 32);
                // (only where workgroups b and b + n_wg / 2 do share a CU on this device: probed once, see placement_rule_holds)
                bool holds = false;
                if (ctx->n_solo > 0) { const int rp = placement_rule_holds(ctx, n_cu, &holds); if (rp != MXE_OK) return rp; }
                ctx->placement_checked = holds ? 1 : 2;
                if (!holds) ctx->n_solo = 0;
            }
            // A binary64 launch that does not fill the GPU (one workgroup per CU by the rule above) is as long as its
            // deepest chain of rounds: its first pass runs in chain_kernel_lv -- binary32, V^T in LDS, a round in a
            // fraction of the time --, every alpha to LV_TOL1, and the binary64 kernel then takes every alpha as a piece
            // of its own from that v: P pieces of one alpha, start vector = the record of the first pass
            if (lv_fits && o.precision == MXE_PRECISION_F64 && ctx->mc_na == 32 && !ctx->mc_gst && ctx->excluded.empty() &&
                o.lds_basis == 1) {
                ctx->lv_mode = 2;
                ctx->mc_wgpc = 1; ctx->n_solo = 0;
                ctx->n_wg = std::min((ctx->n_sub + 3) / 4, n_cu);
                const int P2 = (int)P;
                std::vector<int> e2(P2), p2(P2), l2(P2, 1);
                for (int i = 0; i < P2; ++i) { e2[i] = elem_of_chain[i / n_alpha]; p2[i] = i; }
                ctx->wgpc2 = (o.wg_per_cu != 1 && (P2 + 3) / 4 >= 2 * n_cu && mc_lds_bytes(32, ctx->nwp, 2) <= 80 * 1024 - 2048) ? 2 : 1;
                ctx->n_wg2 = std::min((P2 + 3) / 4, n_cu * ctx->wgpc2);
                HIPCHK(ctx, ctx->d2_elem.ensure(P2)); HIPCHK(ctx, ctx->d2_prob0.ensure(P2));
                HIPCHK(ctx, ctx->d2_len.ensure(P2)); HIPCHK(ctx, ctx->d2_v0.ensure(P2)); HIPCHK(ctx, ctx->d2_queue.ensure(P2));
                HIPCHK(ctx, ctx->dcnt1_niter.ensure(P2)); HIPCHK(ctx, ctx->dcnt1_nevals.ensure(P2));
                HIPCHK(ctx, hipMemcpyAsync(ctx->d2_elem.p, e2.data(), (size_t)P2 * 4, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(ctx->d2_prob0.p, p2.data(), (size_t)P2 * 4, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(ctx->d2_v0.p, p2.data(), (size_t)P2 * 4, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(ctx->d2_queue.p, p2.data(), (size_t)P2 * 4, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(ctx->d2_len.p, l2.data(), (size_t)P2 * 4, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, stream_wait(ctx->stream));          // (the host vectors go out of scope)
            }
        } else {
            // static layout (several data sets: a workgroup streams ONE basis, its four pieces come from one data set and it takes no
            // others): group by data set, four per workgroup, -1 pads.  The four pieces of a workgroup run in lock-step until the
            // longest is through, so pieces of like cost go together (the estimate the queue of the one-data-set launch is ordered
            // by), and the workgroups with the longest pieces are dispatched first: with a data set per element the BASELINE batch
            // 2.13 -> 1.29 ms, with two data sets 1.42 -> 0.95 (tools/many_datasets.py, profiles/r05_experiments.txt 12.)
            std::vector<double> cost(ctx->n_sub);
            for (int sc = 0; sc < ctx->n_sub; ++sc) {
                const int e = ctx->sub_elem[sc];
                double amin = 1e300;
                for (int i = 0; i < ctx->sub_len[sc]; ++i) amin = std::min(amin, alpha_dev[ctx->sub_prob0[sc] + i]);
                cost[sc] = ctx->sub_len[sc] * (ctx->elem_kind[e] == MXE_ENTROPY_NORMAL ? 4.0 : 3.0) +
                           (ctx->elem_kind[e] == MXE_ENTROPY_NORMAL ? 16.0 : 6.0) - 1e-3 * std::log10(amin) +
                           (ctx->sub_walk0[sc] >= 0 ? 2.0 : 0.7) * ctx->sub_pre[sc];
            }
            std::vector<std::vector<int>> by_ds(ctx->ds.size());
            for (int sc = 0; sc < ctx->n_sub; ++sc) by_ds[ctx->elem_ds[ctx->sub_elem[sc]]].push_back(sc);
            std::vector<std::pair<double, std::array<int, 4>>> wgs;
            const bool sorted = !getenv("MXE_NO_SORTED_STATIC");
            for (auto& g : by_ds) {
                if (sorted) std::stable_sort(g.begin(), g.end(), [&](int a, int b) { return cost[a] > cost[b]; });
                for (size_t i0 = 0; i0 < g.size(); i0 += 4) {
                    std::array<int, 4> w4;
                    for (int q = 0; q < 4; ++q) w4[q] = i0 + q < g.size() ? g[i0 + q] : -1;
                    wgs.emplace_back(cost[g[i0]], w4);
                }
            }
            if (sorted) std::stable_sort(wgs.begin(), wgs.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            for (auto& w : wgs) for (int q = 0; q < 4; ++q) ctx->wg_chains.push_back(w.second[q]);
            ctx->n_wg = (int)ctx->wg_chains.size() / 4;
        }
    } else {
        ctx->n_wg = ctx->n_sub;
    }
    HIPCHK(ctx, ctx->dqueue.ensure(std::max<size_t>(ctx->queue.size(), 1)));
    HIPCHK(ctx, ctx->dcounter.ensure(2));
    if (!ctx->queue.empty())
        HIPCHK(ctx, hipMemcpyAsync(ctx->dqueue.p, ctx->queue.data(), ctx->queue.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, ctx->dchain_elem.ensure(ctx->n_sub));
    HIPCHK(ctx, ctx->dsub_prob0.ensure(ctx->n_sub));
    HIPCHK(ctx, ctx->dsub_len.ensure(ctx->n_sub));
    HIPCHK(ctx, ctx->dsub_v0.ensure(ctx->n_sub));
    HIPCHK(ctx, ctx->dsub_pre.ensure(std::max(ctx->n_sub, 1)));
    HIPCHK(ctx, ctx->dwg_chains.ensure(std::max<size_t>(ctx->wg_chains.size(), 1)));
    HIPCHK(ctx, ctx->dalpha.ensure(P));
    HIPCHK(ctx, ctx->dv0.ensure(hv0.size()));
    HIPCHK(ctx, ctx->dout_v.ensure(P * NP));
    // one allocation: H [P][nw] | chi2 S Q [3P] | H of the analyzer's alpha [n_chain][nw] | its index [n_chain]
    // (everything behind H is the COMPACT result pack that a gather between GPUs moves)
    HIPCHK(ctx, ctx->dout_pack.ensure(P * nw + 3 * P + (size_t)n_chain * (nw + 1)));
    ctx->result_buffer = 0;
    ctx->dout_H.p = ctx->dout_pack.p;
    ctx->dout_chi2.p = ctx->dout_pack.p + P * nw;
    ctx->dout_S.p = ctx->dout_chi2.p + P;
    ctx->dout_Q.p = ctx->dout_S.p + P;
    HIPCHK(ctx, ctx->dout_niter.ensure(3 * P));
    ctx->dout_conv.p = ctx->dout_niter.p + P;
    ctx->dout_nevals.p = ctx->dout_conv.p + P;
    HIPCHK(ctx, ctx->dout_nact.ensure(P));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dchain_elem.p, ctx->sub_elem.data(), (size_t)ctx->n_sub * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dsub_prob0.p, ctx->sub_prob0.data(), (size_t)ctx->n_sub * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dsub_len.p, ctx->sub_len.data(), (size_t)ctx->n_sub * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dsub_v0.p, ctx->sub_v0.data(), (size_t)ctx->n_sub * 4, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->has_pre || ctx->mc_gst)     // (the device-memory-state build is the LEAD build: it reads the array)
        HIPCHK(ctx, hipMemcpyAsync(ctx->dsub_pre.p, ctx->sub_pre.data(), (size_t)ctx->n_sub * 4, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->has_walk && ctx->has_pre) {
        HIPCHK(ctx, ctx->dsub_walk0.ensure(std::max(ctx->n_sub, 1)));
        HIPCHK(ctx, ctx->dwalk_alpha.ensure(std::max<size_t>(ctx->walk_alpha.size(), 1)));
        HIPCHK(ctx, hipMemcpyAsync(ctx->dsub_walk0.p, ctx->sub_walk0.data(), (size_t)ctx->n_sub * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->dwalk_alpha.p, ctx->walk_alpha.data(), ctx->walk_alpha.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (!ctx->wg_chains.empty())
        HIPCHK(ctx, hipMemcpyAsync(ctx->dwg_chains.p, ctx->wg_chains.data(), ctx->wg_chains.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (layout != 4) ctx->excluded.clear();
    {
        // every problem belongs to exactly one piece, or to the list the finishing pass takes (checked: a problem that nothing
        // covers would keep whatever the result buffers held before)
        std::vector<char> cov(P, 0);
        size_t twice = 0;
        for (int sc = 0; sc < ctx->n_sub; ++sc)
            for (int i = 0; i < ctx->sub_len[sc]; ++i) { char& c = cov[(size_t)ctx->sub_prob0[sc] + i]; if (c) ++twice; c = 1; }
        for (int x : ctx->excluded) { char& c = cov[(size_t)x]; if (c) ++twice; c = 1; }
        size_t missing = 0;
        for (size_t i = 0; i < P; ++i) if (!cov[i]) ++missing;
        if (missing || twice) {
            fprintf(stderr, "mxe_chains_upload: %zu problems covered by no piece, %zu by two (layout %d, %d pieces, %zu excluded)\n",
                    missing, twice, layout, ctx->n_sub, ctx->excluded.size());
            return MXE_ERR_STATE;
        }
    }
    if (!ctx->excluded.empty()) {
        HIPCHK(ctx, ctx->dexcluded.ensure(ctx->excluded.size()));
        HIPCHK(ctx, hipMemcpyAsync(ctx->dexcluded.p, ctx->excluded.data(), ctx->excluded.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->dalpha.p, alpha_dev.data(), P * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dv0.p, hv0.data(), hv0.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    ctx->n_chain = n_chain; ctx->n_alpha = n_alpha;
    ctx->has_init = false;
    { const int rc_init = build_init_table(ctx, n_chain, elem_of_chain, hv0); if (rc_init != MXE_OK) return rc_init; }
    ctx->chains_ready = true; ctx->launched = false; ctx->sel3_nc = 0;
    return MXE_OK;
}
MXE_CATCH_ALL

// the kernel parameters of the staged chains (mxe_chains_launch; mxe_chains_finish replaces the chain arrays)
constexpr int MC_MAXITER = 32;       // iterations the lock-step kernel spends on one alpha before it gives it up (mxe_chains_finish)
static void fill_kparams(mxe_ctx* ctx, KParams& kp)
{
    const mxe_opts& o = ctx->opts;
    kp.n_omega = ctx->n_omega; kp.n_omega_pad = ctx->nwp; kp.n_s = ctx->n_s; kp.NP = ctx->NP;
    kp.n_alpha = ctx->n_alpha; kp.n_chain = ctx->n_chain;
    kp.Vf = ctx->dVf.p; kp.Vtf = ctx->dVtf.p;
    kp.V = ctx->dV.p; kp.Vx = ctx->dVx.p; kp.Vt = ctx->dVt.p; kp.c = ctx->dc.p; kp.cinv = ctx->dcinv.p;
    kp.elem_ds = ctx->delem_ds.p; kp.elem_kind = ctx->delem_kind.p;
    kp.ghat = ctx->dghat.p; kp.cperp = ctx->dcperp.p; kp.D = ctx->dD.p; kp.sumD = ctx->dsumD.p;
    kp.chain_elem = ctx->dchain_elem.p; kp.alpha = ctx->dalpha.p; kp.v0 = ctx->dv0.p;
    kp.chain_prob0 = ctx->dsub_prob0.p; kp.chain_len = ctx->dsub_len.p; kp.chain_v0 = ctx->dsub_v0.p;
    kp.chain_lead = ((ctx->has_pre || ctx->mc_gst) && ctx->mc_na > 0) ? ctx->dsub_pre.p : nullptr;
    const bool walks = ctx->has_walk && ctx->has_pre && ctx->mc_na > 0;
    kp.chain_walk0 = walks ? ctx->dsub_walk0.p : nullptr;
    kp.walk_alpha = walks ? ctx->dwalk_alpha.p : nullptr;
    kp.init_tab = (ctx->has_init && ctx->mc_na > 0) ? ctx->dinit_tab.p : nullptr;
    kp.chain_init = ctx->dsub_init.p;
    kp.n_chain = ctx->n_sub;
    kp.out_v = ctx->dout_v.p; kp.out_H = ctx->dout_H.p; kp.out_chi2 = ctx->dout_chi2.p;
    kp.out_S = ctx->dout_S.p; kp.out_Q = ctx->dout_Q.p; kp.out_niter = ctx->dout_niter.p;
    kp.out_conv = ctx->dout_conv.p; kp.out_nevals = ctx->dout_nevals.p;
    kp.maxiter = o.maxiter; kp.miniter = o.miniter;
    kp.tol_h = o.tol_h; kp.tol_d = o.tol_d; kp.tol_relq = o.tol_relq; kp.stop_estimate = o.stop_estimate != 0;
    kp.step_max = o.step_max; kp.mu_first = o.mu_first; kp.mu_grow = o.mu_grow; kp.mu_max = o.mu_max;
    kp.theta = o.decouple_tol; kp.out_nact = ctx->dout_nact.p;
    kp.prof = nullptr;
    kp.gstate = nullptr;
    kp.mc_maxiter = std::min(o.maxiter, std::max(MC_MAXITER, o.miniter + 8));     // (a caller's miniter above the limit moves it)
    kp.mc_abandon = 1;
    { const char* e = getenv("MXE_MC_MAXEVALS"); kp.mc_maxevals = e ? atoi(e) : 3 * kp.mc_maxiter; }
    kp.prob_maxiter = nullptr;
    kp.out_index = nullptr;
    kp.dbg_hist = nullptr;
    { const char* e = getenv("MXE_STUCK_SKIP"); kp.stuck_skip = e ? atoi(e) : 1; }
}

namespace {
// one workgroup per problem the lock-step launch leaves out: its records say so until mxe_chains_finish has solved it
// (H, chi2, S, Q = NaN, not converged, no iterations; v = the start vector of its scan, where a chain that begins
// at the head of a scan starts from)
__global__ __launch_bounds__(256)
void clear_excluded_kernel(const int* __restrict__ idx, int n_alpha, int n_omega, int NP, const double* __restrict__ v0,
                           double* __restrict__ H, double* __restrict__ chi2, double* __restrict__ S, double* __restrict__ Q,
                           double* __restrict__ v, int* __restrict__ niter, int* __restrict__ conv, int* __restrict__ nevals, int* __restrict__ nact)
{
    const size_t pidx = (size_t)idx[blockIdx.x];
    const double nan = __builtin_nan("");
    for (int i = threadIdx.x; i < n_omega; i += blockDim.x) H[pidx * n_omega + i] = nan;
    const size_t chain = pidx / n_alpha;
    for (int k = threadIdx.x; k < NP; k += blockDim.x) v[pidx * NP + k] = v0[chain * NP + k];
    if (threadIdx.x == 0) { chi2[pidx] = nan; S[pidx] = nan; Q[pidx] = nan; niter[pidx] = 0; conv[pidx] = 0; nevals[pidx] = 0; nact[pidx] = 0; }
}
// the iteration counts of a two-pass launch (chain_kernel_lv, then chain_kernel_mc per alpha): both passes
__global__ __launch_bounds__(256)
void add_counts_kernel(int* __restrict__ niter, int* __restrict__ nevals, const int* __restrict__ niter1, const int* __restrict__ nevals1, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { niter[i] += niter1[i]; nevals[i] += nevals1[i]; }
}
}  // namespace

namespace {
// Diagnostic (MXE_POISON_LDS = 1: NaN, 2: a pattern of finite values that changes from launch to launch): every CU's LDS is
// written over before a chain kernel starts, so that nothing can lean on what an earlier kernel left there -- the LDS is not
// cleared between kernels, and a read of a cell nobody wrote gives the same answer run after run until the launches before it
// change.  One workgroup of the full 160 KB per CU, and a second wave of them for good measure.
__global__ __launch_bounds__(256)
void scribble_lds_kernel(unsigned long long pattern, int words, unsigned long long* sink)
{
    extern __shared__ unsigned long long sl[];
    for (int i = threadIdx.x; i < words; i += 256) sl[i] = pattern ? (0x3ff0000000000000ull | ((pattern + (unsigned long long)i * 0x9E3779B97F4A7C15ull) & 0x000fffffffffffffull)) : 0x7ff8000000000000ull;
    __syncthreads();
    if (sink && sl[(threadIdx.x * 7) % words] == 1) sink[0] = 1;       // (keeps the stores)
    __builtin_amdgcn_s_sleep(64);
}
hipError_t scribble_lds(hipStream_t s)
{
    static const int mode = getenv("MXE_POISON_LDS") ? atoi(getenv("MXE_POISON_LDS")) : 0;
    if (!mode) return hipSuccess;
    static unsigned long long launches = 0;
    const int bytes = 160 * 1024;
    hipError_t e = hipFuncSetAttribute((const void*)scribble_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    ++launches;
    const unsigned long long pat = mode == 1 ? 0ull : (launches * 0x2545F4914F6CDD1Dull | 1ull);
    hipLaunchKernelGGL(scribble_lds_kernel, dim3(512), dim3(256), bytes, s, pat, bytes / 8, (unsigned long long*)nullptr);
    return hipGetLastError();
}
}  // namespace

// tolerance of the binary32 first pass of a two-pass launch: far enough above the rounding floor of h = V^T H in binary32
// (~2e-7 of |H|) to be reached without crawling, close enough for ONE binary64 Newton step to land below 1e-9
constexpr double LV_TOL1 = 1e-5;

int mxe_chains_launch(mxe_ctx* ctx)
try {
    if (!ctx) return MXE_ERR_ARG;
    if (!ctx->chains_ready) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pre_d = nullptr; ctx->pre_i = nullptr; ctx->pre_rows = nullptr; ctx->pre_index = false;
    const mxe_opts& o = ctx->opts;
    KParams kp;
    fill_kparams(ctx, kp);
    HIPCHK(ctx, scribble_lds(ctx->stream));      // (diagnostic, MXE_POISON_LDS)
    if (getenv("MXE_COUNT_ROUNDS") && ctx->mc_na > 0 && kp.chain_lead == nullptr && ctx->lv_mode == 0) {
        // a diagnostic launch that counts the rounds of its workgroups (mxe_launch_depth) also where the shipped build does not --
        // <32, 2> without led pieces compiles the store out --: the build for led pieces with no piece led, the same schedule
        HIPCHK(ctx, hipMemsetAsync(ctx->dsub_pre.p, 0, (size_t)std::max(ctx->n_sub, 1) * sizeof(int), ctx->stream));
        kp.chain_lead = ctx->dsub_pre.p;
    }
#ifdef MXE_PROFILE
    HIPCHK(ctx, ctx->dprof.ensure(((size_t)ctx->n_sub + 8 * 1024) * 8));   // rows: chain (v2) or workgroup*8 + wave (lock-step)
    HIPCHK(ctx, hipMemsetAsync(ctx->dprof.p, 0, ((size_t)ctx->n_sub + 8 * 1024) * 64, ctx->stream));
    kp.prof = ctx->dprof.p;
#endif
    hipError_t e;
    if (ctx->mc_na > 0) {
        // four chains per workgroup, lock-step (mxe_kernel_mc.hip.h)
        // Eight waves per workgroup (four helpers for the streaming passes) where the launch runs one workgroup per
        // CU, i.e. does not fill the GPU and is as long as its deepest chain of rounds; mxe_opts.waves_per_chain = 4 / 8
        // overrides (the passes of that build want n_omega_pad in units of 256)
        if (ctx->lv_mode != 0) {
            // V^T resident in LDS as binary32 (mxe_kernel_lv.hip.h): the launch itself (precision = F32) or the first pass
            // of a binary64 launch that does not fill the GPU
            const size_t lds1 = mxe::lv_lds_bytes(ctx->n_s, ctx->nwp);
            mxe::MCExtra ex; ex.wg_chains = ctx->dwg_chains.p; ex.n_wg = ctx->n_wg;
            ex.gstate = nullptr; ex.gstate_stride = 0; ex.stagger = 0; ex.n_solo = 0;
            ex.queue = ctx->dqueue.p; ex.n_queue = ctx->n_queue; ex.counter = ctx->dcounter.p;
            HIPCHK(ctx, hipMemsetAsync(ctx->dcounter.p, 0, 2 * sizeof(int), ctx->stream));
            ctx->rounds_n[0] = ctx->n_wg; ctx->rounds_n[1] = (ctx->lv_mode == 2) ? ctx->n_wg2 : 0;
            HIPCHK(ctx, ctx->drounds.ensure((size_t)ctx->rounds_n[0] + ctx->rounds_n[1]));
            HIPCHK(ctx, hipMemsetAsync(ctx->drounds.p, 0, ((size_t)ctx->rounds_n[0] + ctx->rounds_n[1]) * sizeof(int), ctx->stream));
            if (ex.n_queue > 0) ex.wg_chains = ctx->drounds.p;          // (dynamic layout: the rounds of every workgroup come back here)
            else ctx->rounds_n[0] = 0;
            KParams k1 = kp;
            if (ctx->lv_mode == 2) {
                k1.tol_h = std::max(kp.tol_h, LV_TOL1);
                k1.out_H = nullptr;
                k1.out_niter = ctx->dcnt1_niter.p; k1.out_nevals = ctx->dcnt1_nevals.p;
            }
            ctx->last_nw = mxe::LV_NWV; ctx->last_lds = (int)lds1;
            ctx->last_kernel = "mxe::chain_kernel_lv";
            HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            e = hipFuncSetAttribute((const void*)mxe::chain_kernel_lv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
            HIPCHK(ctx, e);
            hipLaunchKernelGGL(mxe::chain_kernel_lv, dim3(ctx->n_wg), dim3(64 * mxe::LV_NWV), lds1, ctx->stream, k1, ex);
            HIPCHK(ctx, hipGetLastError());
            if (ctx->lv_mode == 2) {
                // second pass: every alpha a piece of its own in the binary64 lock-step kernel, from the v of the first
                const size_t Pn = (size_t)ctx->n_chain * ctx->n_alpha;
                const int W2 = ctx->wgpc2;
                const int NWV2 = (W2 == 1 && ctx->nwp % 256 == 0 && mc_lds_bytes(32, ctx->nwp, 1, 8) <= 160 * 1024 - 6144) ? 8 : 4;
                const size_t lds = mc_lds_bytes(32, ctx->nwp, W2, NWV2);
                if (lds > 160 * 1024 - 6144) return MXE_ERR_LIMIT;
                KParams k2 = kp;
                k2.chain_elem = ctx->d2_elem.p; k2.chain_prob0 = ctx->d2_prob0.p; k2.chain_len = ctx->d2_len.p;
                k2.chain_v0 = ctx->d2_v0.p; k2.v0 = ctx->dout_v.p;
                k2.chain_lead = nullptr; k2.init_tab = nullptr; k2.chain_init = nullptr; k2.chain_walk0 = nullptr; k2.walk_alpha = nullptr;
                k2.n_chain = (int)Pn;
                k2.prof = nullptr;                  // (diagnostic build: the stamps of the first pass stay)
                mxe::MCExtra e2 = ex;
                e2.n_wg = ctx->n_wg2;
                e2.queue = ctx->d2_queue.p; e2.n_queue = (int)Pn; e2.counter = ctx->dcounter.p + 1;
                e2.wg_chains = ctx->drounds.p + ctx->rounds_n[0];
#define MXE_LAUNCH_MC2(WG_, NWV_) do { \
                e = hipFuncSetAttribute((const void*)mxe::chain_kernel_mc<32, WG_, false, NWV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (e == hipSuccess) { hipLaunchKernelGGL((mxe::chain_kernel_mc<32, WG_, false, NWV_>), dim3(ctx->n_wg2), dim3(64 * NWV_), lds, ctx->stream, k2, e2); e = hipGetLastError(); } } while (0)
                if (W2 == 2) MXE_LAUNCH_MC2(2, 4); else if (NWV2 == 8) MXE_LAUNCH_MC2(1, 8); else MXE_LAUNCH_MC2(1, 4);
#undef MXE_LAUNCH_MC2
                HIPCHK(ctx, e);
                hipLaunchKernelGGL(add_counts_kernel, dim3((unsigned)((Pn + 255) / 256)), dim3(256), 0, ctx->stream,
                                   ctx->dout_niter.p, ctx->dout_nevals.p, ctx->dcnt1_niter.p, ctx->dcnt1_nevals.p, (int)Pn);
                HIPCHK(ctx, hipGetLastError());
                ctx->last_kernel += " + mxe::chain_kernel_mc<32, " + std::to_string(W2) + (NWV2 == 8 ? ", 8 waves>" : ">") + " per alpha";
            }
            HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
            ctx->launched = true;
            return MXE_OK;
        }
        const int NA = ctx->mc_na, WGPC = ctx->mc_wgpc;
        const bool GST = ctx->mc_gst;
        const int NWV = (NA == 32 && WGPC == 1 && ctx->nwp % 256 == 0 && o.waves_per_chain != 4 &&
                         mc_lds_bytes(NA, ctx->nwp, 1, 8, GST) <= 160 * 1024 - 6144) ? 8 : 4;
        const size_t lds = mc_lds_bytes(NA, ctx->nwp, WGPC, NWV, GST);
        if (lds > 160 * 1024 - 6144) return MXE_ERR_LIMIT;
        mxe::MCExtra ex; ex.wg_chains = ctx->dwg_chains.p; ex.n_wg = ctx->n_wg;
        ex.gstate = nullptr; ex.gstate_stride = 0;
        if (GST) {
            ex.gstate_stride = mxe::mc_gstate_doubles(ctx->nwp);
            HIPCHK(ctx, ctx->dgstate_mc.ensure((size_t)ctx->n_wg * ex.gstate_stride + 4096));
            ex.gstate = ctx->dgstate_mc.p;
        }
        ex.queue = ctx->dqueue.p; ex.n_queue = ctx->n_queue; ex.counter = ctx->dcounter.p;
        // rounds per workgroup (mxe_launch_depth): the builds for launches that do not fill the GPU write them, in the dynamic
        // layout (MCExtra: the pointer is the table of chain ids in the static one).  Not <32, 2> without led pieces -- the batch
        // that fills the GPU: the memset node alone is 1.5 us of its 812 (A/B: profiles/r04_experiments.txt)
        ctx->rounds_n[0] = ctx->rounds_n[1] = 0;
        // (MXE_COUNT_ROUNDS: there as well -- a diagnostic launch, bench.py: launch_depth)
        if (ex.n_queue > 0 && (kp.chain_lead != nullptr || WGPC == 1 || getenv("MXE_COUNT_ROUNDS"))) {
            ctx->rounds_n[0] = ctx->n_wg;
            HIPCHK(ctx, ctx->drounds.ensure((size_t)ctx->n_wg));
            HIPCHK(ctx, hipMemsetAsync(ctx->drounds.p, 0, (size_t)ctx->n_wg * sizeof(int), ctx->stream));
            ex.wg_chains = ctx->drounds.p;
        }
        ex.stagger = 0;              // (a late start of the second half of the grid never paid: 0 ... 14 units measured; 5 cost 0.4 %; with the wave priorities of r03: 0 ... 24 units all within 0.5 %)
        ex.n_solo = 0;
        int counter0 = 0;
        if (ex.n_queue > 0 && WGPC == 2 && ctx->n_solo > 0) {
            ex.n_solo = ctx->n_solo;
            counter0 = (ctx->n_wg - ex.n_solo) * 4;
        }
        HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->dcounter.p, counter0, 1, ctx->stream));
        if (!ctx->excluded.empty()) {
            hipLaunchKernelGGL(clear_excluded_kernel, dim3((unsigned)ctx->excluded.size()), dim3(256), 0, ctx->stream,
                               ctx->dexcluded.p, ctx->n_alpha, ctx->n_omega, ctx->NP, ctx->dv0.p, ctx->dout_H.p, ctx->dout_chi2.p,
                               ctx->dout_S.p, ctx->dout_Q.p, ctx->dout_v.p, ctx->dout_niter.p, ctx->dout_conv.p,
                               ctx->dout_nevals.p, ctx->dout_nact.p);
            HIPCHK(ctx, hipGetLastError());
        }
        ctx->last_nw = NWV; ctx->last_lds = (int)lds;
        const bool lead = kp.chain_lead != nullptr;
        ctx->last_kernel = "mxe::chain_kernel_mc<" + std::to_string(NA) + ", " + std::to_string(WGPC) + (lead ? ", lead" : "") +
                           (NWV == 8 ? ", 8 waves" : "") + (GST ? ", device-memory state>" : ">");
        HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
#define MXE_LAUNCH_MC(NA_, WG_, LD_, ...) do { constexpr int NWV_ = std::get<0>(std::make_tuple(__VA_ARGS__)); \
        e = hipFuncSetAttribute((const void*)mxe::chain_kernel_mc<NA_, WG_, LD_, __VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess && getenv("MXE_DEBUG_OCC")) { int nb__ = 0; \
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb__, (const void*)mxe::chain_kernel_mc<NA_, WG_, LD_, __VA_ARGS__>, 64 * NWV_, lds); \
            fprintf(stderr, "[mxe] lock-step kernel NA=%d NWV=%d lds=%zu: %d workgroup(s) per CU resident\n", NA_, NWV_, (size_t)lds, nb__); } \
        if (e == hipSuccess) { hipLaunchKernelGGL((mxe::chain_kernel_mc<NA_, WG_, LD_, __VA_ARGS__>), dim3(ctx->n_wg), dim3(64 * NWV_), lds, ctx->stream, kp, ex); e = hipGetLastError(); } } while (0)
        if (NA == 64) { if (lead) MXE_LAUNCH_MC(64, 1, true, 4); else MXE_LAUNCH_MC(64, 1, false, 4); }
        else if (GST) { if (NWV == 8) MXE_LAUNCH_MC(32, 1, true, 8, true); else MXE_LAUNCH_MC(32, 1, true, 4, true); }
        else if (NA == 32 && WGPC == 2) { if (lead) MXE_LAUNCH_MC(32, 2, true, 4); else MXE_LAUNCH_MC(32, 2, false, 4); }
        else if (NWV == 8) { if (lead) MXE_LAUNCH_MC(32, 1, true, 8); else MXE_LAUNCH_MC(32, 1, false, 8); }
        else { if (lead) MXE_LAUNCH_MC(32, 1, true, 4); else MXE_LAUNCH_MC(32, 1, false, 4); }
#undef MXE_LAUNCH_MC
        HIPCHK(ctx, e);
    } else {
        ctx->rounds_n[0] = ctx->rounds_n[1] = 0;
        int NW = std::min(o.waves_per_chain, 4);     // (eight waves per chain: two per SIMD at 256 registers each spilled; not built)
        if (NW == 0) {
            // fill the 256 CUs x 4 SIMDs: few chains -> more waves per chain
            const int nc = ctx->n_sub;
            NW = (nc >= 2048) ? 1 : (nc >= 1024) ? 2 : 4;
        }
        const bool f32 = (o.precision == MXE_PRECISION_F32);
        size_t lds = lds_bytes(ctx->NP, ctx->nwp, NW, f32);
        while (lds > 160 * 1024 && NW > 1) { NW /= 2; lds = lds_bytes(ctx->NP, ctx->nwp, NW, f32); }
        // a frequency mesh whose omega-space state (five arrays per chain) does not fit the LDS: the state goes to
        // device memory (four waves per chain; one where the 128 x 128 Newton matrix leaves no room for more)
        const bool gst = lds > 160 * 1024;
        if (gst) {
            NW = (ctx->NP == 64) ? 4 : 1;
            lds = lds_bytes(ctx->NP, ctx->nwp, NW, f32, true);
            if (lds > 160 * 1024) return MXE_ERR_LIMIT;
            HIPCHK(ctx, ctx->dgstate.ensure((size_t)ctx->n_sub * 5 * ctx->nwp));     // (binary32 build: half of it used)
            kp.gstate = ctx->dgstate.p;
        }
        ctx->last_nw = NW; ctx->last_lds = (int)lds;
        ctx->last_kernel = "mxe::chain_kernel<" + std::to_string(NW) + ", " + std::to_string(ctx->NP / 32) + (f32 ? ", float" : ", double") + (gst ? ", device-memory state>" : ">");
        HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        if (gst) {
            if (f32) e = launch_t<4, 2, float, true>(kp, lds, ctx->stream);
            else e = (ctx->NP == 64) ? launch_t<4, 2, double, true>(kp, lds, ctx->stream) : launch_t<1, 4, double, true>(kp, lds, ctx->stream);
        } else
        if (f32) {
            switch (NW) {
                case 1: e = launch_t<1, 2, float>(kp, lds, ctx->stream); break;
                case 2: e = launch_t<2, 2, float>(kp, lds, ctx->stream); break;
                case 4: e = launch_t<4, 2, float>(kp, lds, ctx->stream); break;
                default: e = launch_t<4, 2, float>(kp, lds, ctx->stream); break;
            }
        } else
        switch (NW) {
            case 1: e = launch_nab<1>(ctx->NP, kp, lds, ctx->stream); break;
            case 2: e = launch_nab<2>(ctx->NP, kp, lds, ctx->stream); break;
            case 4: e = launch_nab<4>(ctx->NP, kp, lds, ctx->stream); break;
            default: e = launch_nab<4>(ctx->NP, kp, lds, ctx->stream); break;
        }
        HIPCHK(ctx, e);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->launched = true;
    return MXE_OK;
}
MXE_CATCH_ALL

int mxe_sync(mxe_ctx* ctx)
{
    if (!ctx) return MXE_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, stream_wait(ctx->stream));
    return MXE_OK;
}

// Alphas that the lock-step layout gave up on are solved again in the one-chain layout, from the state they were
// left in.  The lock-step kernel forms its Gram matrices from binary16 products (21 bits): an inexact
// Newton matrix that costs nothing where the system is well conditioned and stalls the iteration where it is not
// (few data points, small alpha: tens to hundreds of iterations per alpha where the binary64 Gram matrix of the
// one-chain kernel takes five); it stops an alpha after MC_MAXITER iterations.  Blocking; a no-op after a launch of
// the one-chain layout and when everything converged.
int mxe_chains_finish(mxe_ctx* ctx, int32_t* n_resolved)
try {
    if (n_resolved) *n_resolved = 0;
    if (!ctx) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    ctx->last_finished = 0;
    if (ctx->mc_na == 0) return MXE_OK;                       // the launch was in the one-chain layout
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, stream_wait(ctx->stream));
    const mxe_opts& o = ctx->opts;
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    std::vector<int> conv(P), nit(P), nev(P);
    HIPCHK(ctx, hipMemcpy(conv.data(), ctx->dout_conv.p, P * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> todo;
    for (size_t i = 0; i < P; ++i) if (!conv[i]) todo.push_back((int)i);
    if (todo.empty()) return MXE_OK;
    HIPCHK(ctx, hipMemcpy(nit.data(), ctx->dout_niter.p, P * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(nev.data(), ctx->dout_nevals.p, P * sizeof(int), hipMemcpyDeviceToHost));
    // (an alpha that used up the caller's own maxiter stays as it is)
    std::vector<char> open(P, 0);
    int n = 0;
    for (int i : todo) if (nit[i] < o.maxiter) { open[i] = 1; ++n; }
    if (n == 0) return MXE_OK;
    ctx->pre_d = nullptr; ctx->pre_i = nullptr; ctx->pre_rows = nullptr; ctx->pre_index = false;   // (what was copied out behind the launch is about to change)
    // The alphas to solve again, as the reference would have reached them: every run of consecutive open alphas of a scan is
    // ONE warm-started chain that begins at the solution of the alpha before it (a converged neighbour: 3-5 iterations
    // per alpha, where the cold-started piece the lock-step kernel gave up on was stuck far from the minimiser -- on inputs
    // whose error bar lies far below their noise a cold start does not get there in thousands of evaluations, the
    // reference's warm scan does: tests/golden/stress_reference.npz); a run at the head of a scan starts from the state its
    // first alpha was left in.
    const int NP = ctx->NP, na = ctx->n_alpha;
    std::vector<int> f_elem, f_prob0, f_len, f_v0;
    std::vector<size_t> f_src;                       // row of dout_v the run starts from
    // The entries of the chains: the open alphas of the mesh and -- where the mesh is too coarse to step on (round 5) -- RUNGS in
    // between.  What is left to this pass lies deep in the region where the entropy term no longer holds the solution (or couples
    // more directions than the lock-step layout takes); there a warm step over the factor 1.9 of the reference's default mesh
    // (alpha_meshes.py:81: 20 alphas from 20 down to 1e-4) took 500-2 300 evaluations where ten steps of 10 % take 35 each
    // (profiles/r05_b_coarse_mesh.txt, tools/stress.py case 17).  A step over more than FIN_COARSE is therefore cut into equal
    // rungs of at most FIN_RATIO; a rung gets FIN_RUNG_ITERS iterations, no record, and its cost is counted with the alpha it leads to.
    constexpr double FIN_COARSE = 1.6;
    const double FIN_RATIO = getenv("MXE_FIN_RATIO") ? std::max(1.05, atof(getenv("MXE_FIN_RATIO"))) : 1.3;
    const int FIN_RUNG_ITERS = getenv("MXE_FIN_RUNG_ITERS") ? std::max(1, atoi(getenv("MXE_FIN_RUNG_ITERS"))) : 10;
    constexpr int FIN_RUNGS_MAX = 40;
    const bool rungs_ok = !getenv("MXE_NO_FINISH_LADDER");
    std::vector<double> halpha(P);
    HIPCHK(ctx, hipMemcpy(halpha.data(), ctx->dalpha.p, P * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> e_alpha;                     // per entry
    std::vector<int> e_out, e_budget;
    int n_rungs = 0;
    for (int c = 0; c < ctx->n_chain; ++c)
        for (int i = 0; i < na; ) {
            if (!open[(size_t)c * na + i]) { ++i; continue; }
            int j = i;
            while (j < na && open[(size_t)c * na + j]) ++j;
            f_elem.push_back(ctx->chain_elem[c]); f_prob0.push_back((int)e_alpha.size());
            f_v0.push_back((int)f_v0.size());
            // (a run at the head of a scan starts from the state its first alpha was left in -- if it was ever evaluated: a led piece
            //  that gives up during its WALK leaves no record of its own alpha, and the vector in the result buffer is then whatever
            //  the block held before: the start vector of the scan instead.  SIZE_MAX marks that)
            const size_t p_head = (size_t)c * na + i;
            f_src.push_back(i > 0 ? p_head - 1 : ((nit[p_head] > 0 || nev[p_head] > 0) ? p_head : (size_t)-1 - (size_t)c));
            if (rungs_ok && i == 0 && f_src.back() > (size_t)-1 - (size_t)ctx->n_chain - 1) {
                // a run that begins at the head of a scan from the default model, deep in the hard region (an ascending mesh, or a
                // mesh that lies there altogether): rungs from N_data / 4 down to its first alpha, like the led pieces of the
                // lock-step kernel (mxe_chains_upload: emit_ladder) -- the cold start at alpha~ = 0.5 of tools/stress.py case 17 ran
                // into maxiter
                const double a_to = halpha[p_head];
                const double a_from = 0.25 * ctx->ds[ctx->elem_ds[ctx->chain_elem[c]]].n_rows;
                // (only where the mesh goes on FINE from there -- the rising mesh of case 17: 148 -> 150 of 150 converged, 61 -> 8 ms;
                //  case 5: 29 -> 4.5 ms.  With three alphas five decades apart, alpha~ = 0.06 at sigma = 1e-5 on 40 data points, the
                //  cold start with its full budget converged 129 of 192 and the rungs 99-108 at any ratio and budget tried: case 93)
                const bool fine_head = na > 1 && std::max(halpha[p_head + 1] / a_to, a_to / halpha[p_head + 1]) <= FIN_COARSE;
                if (fine_head && a_to * FIN_RATIO * FIN_RATIO < a_from) {
                    int m = (int)std::ceil(std::log(a_from / a_to) / std::log(FIN_RATIO) - 1e-9);
                    m = std::max(2, std::min(m, FIN_RUNGS_MAX + 1));
                    const double q = std::pow(a_to / a_from, 1.0 / m);
                    for (int s2 = 0; s2 < m; ++s2) {
                        e_alpha.push_back(a_from * std::pow(q, s2)); e_out.push_back(-1); e_budget.push_back(s2 == 0 ? 3 * FIN_RUNG_ITERS : FIN_RUNG_ITERS);
                        ++n_rungs;
                    }
                }
            }
            for (int k = i; k < j; ++k) {
                const size_t pk = (size_t)c * na + k;
                if (rungs_ok && k > 0) {
                    const double a_from = halpha[pk - 1], a_to = halpha[pk];
                    const double r = a_from > a_to ? a_from / a_to : a_to / a_from;
                    if (r > FIN_COARSE) {
                        int m = (int)std::ceil(std::log(r) / std::log(FIN_RATIO) - 1e-9);          // steps, the last of them the alpha itself
                        m = std::max(2, std::min(m, FIN_RUNGS_MAX + 1));
                        const double q = std::pow(a_to / a_from, 1.0 / m);
                        for (int s2 = 1; s2 < m; ++s2) {
                            e_alpha.push_back(a_from * std::pow(q, s2)); e_out.push_back(-1); e_budget.push_back(FIN_RUNG_ITERS);
                            ++n_rungs;
                        }
                    }
                }
                e_alpha.push_back(halpha[pk]); e_out.push_back((int)pk);
                // the caller's maxiter bounds the iterations of an alpha over BOTH passes, alpha by alpha as the reference caps them
                // (levenberg_minimizer.py:155): every open alpha gets what the lock-step pass left of ITS budget -- an alpha that pass
                // never touched (the rest of an abandoned piece, an excluded alpha: nit = 0) the whole of it.  (Until round 5 the pass had
                // ONE budget, that of the open alpha with the most iterations behind it: ADVICE r04.)
                e_budget.push_back(std::max(1, o.maxiter - nit[pk]));
            }
            f_len.push_back((int)e_alpha.size() - f_prob0.back());
            i = j;
        }
    const int nr = (int)f_elem.size();
    const size_t ne = e_alpha.size();
    HIPCHK(ctx, ctx->dfin_elem.ensure(nr)); HIPCHK(ctx, ctx->dfin_prob0.ensure(nr));
    HIPCHK(ctx, ctx->dfin_len.ensure(nr)); HIPCHK(ctx, ctx->dfin_v0.ensure(nr));
    HIPCHK(ctx, ctx->dfin_start.ensure((size_t)nr * NP));
    HIPCHK(ctx, ctx->dfin_alpha.ensure(ne)); HIPCHK(ctx, ctx->dfin_out.ensure(ne)); HIPCHK(ctx, ctx->dfin_budget.ensure(ne));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_elem.p, f_elem.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_prob0.p, f_prob0.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_len.p, f_len.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_v0.p, f_v0.data(), nr * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_alpha.p, e_alpha.data(), ne * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_out.p, e_out.data(), ne * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_budget.p, e_budget.data(), ne * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    for (int k = 0; k < nr; ++k) {       // start vectors (whitened basis, row stride NP)
        const bool from_v0 = f_src[k] > (size_t)-1 - (size_t)ctx->n_chain - 1;
        const double* from = from_v0 ? ctx->dv0.p + ((size_t)-1 - f_src[k]) * NP : ctx->dout_v.p + f_src[k] * NP;
        HIPCHK(ctx, hipMemcpyAsync(ctx->dfin_start.p + (size_t)k * NP, from, NP * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(ctx, stream_wait(ctx->stream));          // (the host vectors are locals)
    KParams kp;
    fill_kparams(ctx, kp);
    kp.chain_elem = ctx->dfin_elem.p; kp.chain_prob0 = ctx->dfin_prob0.p; kp.chain_len = ctx->dfin_len.p;
    kp.chain_v0 = ctx->dfin_v0.p; kp.v0 = ctx->dfin_start.p;
    kp.chain_lead = nullptr; kp.init_tab = nullptr; kp.chain_init = nullptr; kp.chain_walk0 = nullptr; kp.walk_alpha = nullptr;
    kp.n_chain = nr;
    kp.alpha = ctx->dfin_alpha.p; kp.out_index = ctx->dfin_out.p; kp.prob_maxiter = ctx->dfin_budget.p;
    ctx->last_rungs = n_rungs;
#ifdef MXE_DEBUG_HIST
    DevBuf<double> dhist;
    if (getenv("MXE_DEBUG_HIST_FILE")) {
        HIPCHK(ctx, dhist.ensure(ne * 48));
        HIPCHK(ctx, hipMemsetAsync(dhist.p, 0, ne * 48 * sizeof(double), ctx->stream));
        kp.dbg_hist = dhist.p;
    }
#endif
#ifdef MXE_PROFILE
    // (diagnostic build: the stamps of THIS pass, rows 0 .. nr - 1 -- tools/finish_phases.py; those of the lock-step launch are gone)
    if (ctx->dprof.p && (size_t)nr <= (size_t)ctx->n_sub + 8 * 1024) {
        HIPCHK(ctx, hipMemsetAsync(ctx->dprof.p, 0, ((size_t)ctx->n_sub + 8 * 1024) * 64, ctx->stream));
        kp.prof = ctx->dprof.p;
    }
#endif
    const int NW = 4;
    size_t lds = lds_bytes(NP, ctx->nwp, NW, false);
    hipError_t e;
    HIPCHK(ctx, scribble_lds(ctx->stream));      // (diagnostic, MXE_POISON_LDS)
    if (lds > 160 * 1024) {
        // (a frequency mesh beyond the LDS -- the lock-step launch kept its state in device memory --: so does this one)
        lds = lds_bytes(NP, ctx->nwp, NW, false, true);
        if (lds > 160 * 1024) return MXE_ERR_LIMIT;
        HIPCHK(ctx, ctx->dgstate.ensure((size_t)nr * 5 * ctx->nwp));
        kp.gstate = ctx->dgstate.p;
        e = launch_t<4, 2, double, true>(kp, lds, ctx->stream);
    } else {
        e = launch_t<4, 2>(kp, lds, ctx->stream);
    }
    HIPCHK(ctx, e);
    HIPCHK(ctx, stream_wait(ctx->stream));
#ifdef MXE_DEBUG_HIST
    if (kp.dbg_hist) {
        std::vector<double> hh(ne * 48);
        HIPCHK(ctx, hipMemcpy(hh.data(), dhist.p, ne * 48 * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<int> cv(P);
        HIPCHK(ctx, hipMemcpy(cv.data(), ctx->dout_conv.p, P * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> ni(P);
        HIPCHK(ctx, hipMemcpy(ni.data(), ctx->dout_niter.p, P * sizeof(int), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(getenv("MXE_DEBUG_HIST_FILE"), "a")) {
            for (size_t en = 0; en < ne; ++en) {
                if (e_out[en] < 0) continue;
                fprintf(f, "%d %d %d %.6e", e_out[en], cv[e_out[en]], ni[e_out[en]], e_alpha[en]);
                for (int m = 0; m < 48; ++m) fprintf(f, "%s%.3e", (m % 16 == 0) ? " | " : " ", hh[en * 48 + m]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
        dhist.release();
    }
#endif
    // the counters of the records: both passes (whole arrays: one copy each way)
    std::vector<int> nit2(P), nev2(P);
    HIPCHK(ctx, hipMemcpy(nit2.data(), ctx->dout_niter.p, P * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(nev2.data(), ctx->dout_nevals.p, P * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < P; ++i) if (open[i]) { nit2[i] += nit[i]; nev2[i] += nev[i]; }
    HIPCHK(ctx, hipMemcpy(ctx->dout_niter.p, nit2.data(), P * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->dout_nevals.p, nev2.data(), P * sizeof(int), hipMemcpyHostToDevice));
    ctx->last_finished = n;
    if (n_resolved) *n_resolved = n;
    return MXE_OK;
}
MXE_CATCH_ALL

int mxe_chains_fetch(mxe_ctx* ctx, double* out_v, double* out_H, double* out_chi2,
                     double* out_S, double* out_Q, int32_t* out_niter,
                     int32_t* out_converged, int32_t* out_nevals)
try {
    if (!ctx) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, stream_wait(ctx->stream));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    const int ns = ctx->n_s, NP = ctx->NP, nw = ctx->n_omega;
    if (out_H) HIPCHK(ctx, d2h_pipelined(out_H, ctx->dout_H.p, P * nw * 8, ctx->stream));
    // (chi2 | S | Q and niter | converged | nevals are one block each on the device: a caller whose arrays lie the same way --
    //  maxent_amd.device.DeviceContext.fetch allocates them so -- gets each block in ONE copy; six copies of 100-200 KB
    //  cost 0.15 ms behind a launch of 0.8 ms)
    const bool pre = ctx->pre_d && out_chi2 == ctx->pre_d && out_S == out_chi2 + P && out_Q == out_S + P &&
                     out_niter == ctx->pre_i && out_converged == out_niter + P && out_nevals == out_converged + P;
    if (pre) {}                                  // (mxe_chains_prefetch queued both blocks behind the launch: they are there)
    else if (out_chi2 && out_S == out_chi2 + P && out_Q == out_S + P)
        HIPCHK(ctx, hipMemcpy(out_chi2, ctx->dout_chi2.p, 3 * P * 8, hipMemcpyDeviceToHost));
    else {
        if (out_chi2) HIPCHK(ctx, hipMemcpy(out_chi2, ctx->dout_chi2.p, P * 8, hipMemcpyDeviceToHost));
        if (out_S) HIPCHK(ctx, hipMemcpy(out_S, ctx->dout_S.p, P * 8, hipMemcpyDeviceToHost));
        if (out_Q) HIPCHK(ctx, hipMemcpy(out_Q, ctx->dout_Q.p, P * 8, hipMemcpyDeviceToHost));
    }
    if (out_Q && ctx->chi2_factor != 1.0 && !(pre && ctx->pre_scaled)) for (size_t i = 0; i < P; ++i) out_Q[i] *= ctx->chi2_factor;
    if (pre) ctx->pre_scaled = true;
    if (pre) {}
    else if (out_niter && out_converged == out_niter + P && out_nevals == out_converged + P)
        HIPCHK(ctx, hipMemcpy(out_niter, ctx->dout_niter.p, 3 * P * 4, hipMemcpyDeviceToHost));
    else {
        if (out_niter) HIPCHK(ctx, hipMemcpy(out_niter, ctx->dout_niter.p, P * 4, hipMemcpyDeviceToHost));
        if (out_converged) HIPCHK(ctx, hipMemcpy(out_converged, ctx->dout_conv.p, P * 4, hipMemcpyDeviceToHost));
        if (out_nevals) HIPCHK(ctx, hipMemcpy(out_nevals, ctx->dout_nevals.p, P * 4, hipMemcpyDeviceToHost));
    }
    if (out_v) {
        std::vector<double> hv(P * NP);
        HIPCHK(ctx, hipMemcpy(hv.data(), ctx->dout_v.p, P * NP * 8, hipMemcpyDeviceToHost));
        for (size_t pidx = 0; pidx < P; ++pidx) {
            const int chain = (int)(pidx / ctx->n_alpha);
            const DataSet& DS = ctx->ds[ctx->elem_ds[ctx->chain_elem[chain]]];
            for (int k = 0; k < ns; ++k) {
                double s;
                if (DS.identity_q) s = hv[pidx * NP + k];
                else {   // v = Q v'
                    s = 0.0;
                    for (int j = 0; j < ns; ++j) s += DS.Q[(size_t)k * ns + j] * hv[pidx * NP + j];
                }
                out_v[pidx * ns + k] = s;
            }
        }
    }
    return MXE_OK;
}
MXE_CATCH_ALL

namespace {
// Results written into page-locked host memory by the compute queue (mxe_chains_prefetch, mxe_select3_prefetch_rows).  A
// hipMemcpyAsync behind a kernel parks a DMA engine until that kernel is through, and the uploads of the NEXT job (new data on
// another context's stream) queued up behind it: mxe_elements_update_data took 0.9 ms instead of 0.3 with four jobs in flight.
struct CopySeg { const uint32_t* src; uint32_t* dst; size_t words; };
__global__ __launch_bounds__(256)
void copy_out_kernel(CopySeg a, CopySeg b)
{
    const size_t stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < a.words; i += stride) a.dst[i] = a.src[i];
    for (size_t i = t0; i < b.words; i += stride) b.dst[i] = b.src[i];
}
hipError_t copy_out(void* dst_a, const void* src_a, size_t bytes_a, void* dst_b, const void* src_b, size_t bytes_b, bool mapped, hipStream_t s)
{
    if (!mapped || (bytes_a & 3) || (bytes_b & 3)) {
        hipError_t e = bytes_a ? hipMemcpyAsync(dst_a, src_a, bytes_a, hipMemcpyDeviceToHost, s) : hipSuccess;
        if (e == hipSuccess && bytes_b) e = hipMemcpyAsync(dst_b, src_b, bytes_b, hipMemcpyDeviceToHost, s);
        return e;
    }
    CopySeg a{(const uint32_t*)src_a, (uint32_t*)dst_a, bytes_a / 4}, b{(const uint32_t*)src_b, (uint32_t*)dst_b, bytes_b / 4};
    const size_t words = std::max(a.words, b.words);
    const int blocks = (int)std::min<size_t>(64, (words + 255) / 256);
    hipLaunchKernelGGL(copy_out_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, s, a, b);
    return hipGetLastError();
}
}

int mxe_chains_prefetch(mxe_ctx* ctx, double* out_chi2_S_Q, int32_t* out_niter_converged_nevals)
try {
    if (!ctx || !out_chi2_S_Q || !out_niter_converged_nevals) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    ctx->pre_d = nullptr; ctx->pre_i = nullptr;
    const bool mapped = g_host.holds(out_chi2_S_Q, 3 * P * 8) && g_host.holds(out_niter_converged_nevals, 3 * P * 4);
    HIPCHK(ctx, copy_out(out_chi2_S_Q, ctx->dout_chi2.p, 3 * P * 8, out_niter_converged_nevals, ctx->dout_niter.p, 3 * P * 4, mapped, ctx->stream));
    ctx->pre_d = out_chi2_S_Q; ctx->pre_i = out_niter_converged_nevals; ctx->pre_scaled = false;
    return MXE_OK;
}
MXE_CATCH_ALL

int mxe_chains_fetch_nact(mxe_ctx* ctx, int32_t* out_nact)
{
    if (!ctx || !out_nact) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, stream_wait(ctx->stream));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    HIPCHK(ctx, hipMemcpy(out_nact, ctx->dout_nact.p, P * 4, hipMemcpyDeviceToHost));
    return MXE_OK;
}

int mxe_solve_chains(mxe_ctx* ctx, int n_chain, int n_alpha,
                     const int32_t* elem_of_chain, const double* alpha_scaled,
                     const double* v0, const mxe_opts* opts,
                     double* out_v, double* out_H, double* out_chi2,
                     double* out_S, double* out_Q, int32_t* out_niter,
                     int32_t* out_converged, int32_t* out_nevals)
{
    int rc = mxe_chains_upload(ctx, n_chain, n_alpha, elem_of_chain, alpha_scaled, v0, opts);
    if (rc != MXE_OK) return rc;
    rc = mxe_chains_launch(ctx);
    if (rc != MXE_OK) return rc;
    rc = mxe_chains_finish(ctx, nullptr);
    if (rc != MXE_OK) return rc;
    return mxe_chains_fetch(ctx, out_v, out_H, out_chi2, out_S, out_Q, out_niter, out_converged, out_nevals);
}

int mxe_result_device_ptrs(mxe_ctx* ctx, void** d_H, void** d_chi2, void** d_S, void** d_Q,
                           void** d_v, void** d_niter, void** d_converged)
{
    if (!ctx) return MXE_ERR_ARG;
    if (!ctx->chains_ready) return MXE_ERR_STATE;
    if (d_H) *d_H = ctx->dout_H.p;
    if (d_chi2) *d_chi2 = ctx->dout_chi2.p;
    if (d_S) *d_S = ctx->dout_S.p;
    if (d_Q) *d_Q = ctx->dout_Q.p;
    if (d_v) *d_v = ctx->dout_v.p;
    if (d_niter) *d_niter = ctx->dout_niter.p;
    if (d_converged) *d_converged = ctx->dout_conv.p;
    return MXE_OK;
}

int mxe_ns_padded(mxe_ctx* ctx) { return ctx ? ctx->NP : MXE_ERR_ARG; }

int mxe_set_result_buffer(mxe_ctx* ctx, int which)
{
    if (!ctx || (which != 0 && which != 1)) return MXE_ERR_ARG;
    if (!ctx->chains_ready) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha, nw = ctx->n_omega;
    if (which == 1) HIPCHK(ctx, ctx->dout_pack2.ensure(P * nw + 3 * P + (size_t)ctx->n_chain * (nw + 1)));
    double* base = which ? ctx->dout_pack2.p : ctx->dout_pack.p;
    ctx->result_buffer = which;
    ctx->dout_H.p = base; ctx->dout_chi2.p = base + P * nw; ctx->dout_S.p = ctx->dout_chi2.p + P; ctx->dout_Q.p = ctx->dout_S.p + P;
    return MXE_OK;
}

int mxe_last_kernel_ms(mxe_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return MXE_OK;
}

const char* mxe_last_kernel_name(mxe_ctx* ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }

void* mxe_stream(mxe_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int mxe_timing_mark(mxe_ctx* ctx)
{
    if (!ctx) return MXE_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventRecord(ctx->ev_mark, ctx->stream));
    return MXE_OK;
}

int mxe_ms_since_mark(mxe_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev_mark, ctx->ev1));
    return MXE_OK;
}

int mxe_last_launch_info(mxe_ctx* ctx, int* waves_per_chain, int* n_workgroups, int* lds_bytes)
{
    if (!ctx) return MXE_ERR_ARG;
    if (waves_per_chain) *waves_per_chain = ctx->last_nw;
    if (n_workgroups) *n_workgroups = ctx->n_wg;
    if (lds_bytes) *lds_bytes = ctx->last_lds;
    return MXE_OK;
}

int mxe_schedule_info(mxe_ctx* ctx, int* n_solo, int* placement_rule)
{
    if (!ctx) return MXE_ERR_ARG;
    if (n_solo) *n_solo = ctx->n_solo;
    if (placement_rule) *placement_rule = ctx->placement_checked;
    return MXE_OK;
}

// depth of the last lock-step launch in rounds (a round = one Newton iteration of the four slots of a workgroup): maximum and
// mean over its workgroups, [0] the launch (or its first pass), [1] the second pass of a two-pass launch (0 when there is none;
// both 0 after a launch of the one-chain layout).  Blocking.
int mxe_launch_depth(mxe_ctx* ctx, int32_t* max_rounds, double* mean_rounds)
try {
    if (!ctx || !max_rounds || !mean_rounds) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, stream_wait(ctx->stream));
    const size_t n = (size_t)ctx->rounds_n[0] + ctx->rounds_n[1];
    std::vector<int> h(n ? n : 1, 0);
    if (n) HIPCHK(ctx, hipMemcpy(h.data(), ctx->drounds.p, n * sizeof(int), hipMemcpyDeviceToHost));
    size_t off = 0;
    for (int k = 0; k < 2; ++k) {
        int mx = 0; double sum = 0.0; int busy = 0;
        for (int i = 0; i < ctx->rounds_n[k]; ++i) { mx = std::max(mx, h[off + i]); sum += h[off + i]; busy += h[off + i] > 0; }
        max_rounds[k] = mx; mean_rounds[k] = busy ? sum / busy : 0.0;
        off += ctx->rounds_n[k];
    }
    return MXE_OK;
}
MXE_CATCH_ALL

} // extern "C"

#ifdef MXE_PROFILE
// diagnostic build only: per-chain phase cycle counters of the last launch
extern "C" int mxe_prof_fetch(mxe_ctx* ctx, long long* out /*[n_sub + 8192][8]*/)
{
    if (!ctx || !out) return MXE_ERR_ARG;
    HIPCHK(ctx, stream_wait(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, ctx->dprof.p, ((size_t)ctx->n_sub + 8 * 1024) * 64, hipMemcpyDeviceToHost));
    return MXE_OK;
}
#endif

// ---- log det(I + M W / alpha) of every problem of the last launch ------------
namespace mxe {
// One workgroup (4 waves) per problem.  In the whitened basis M = diag(c^2), so
// det(I + M W / a) = det(c W c + a I) / a^n_s with W = V^T diag(w) V over ALL n_s
// kept directions (no active-subspace cut here).  w is rebuilt from the stored H
// (normal: w = H; plusminus: w = sqrt(H^2 + 4 D^2), free of cancellation).
//   1. W by v_mfma_f64_16x16x4_f64, the omega rows split over the waves, the
//      upper-triangular 16x16 tiles (mt <= nt) of one tile row per sweep of V;
//   2. B = c W c + a I, Cholesky in LDS (right-looking, all threads);
//   3. log det = 2 sum log L_jj - n_s log a.
template <int NT>
__global__ __launch_bounds__(256)
void logdet_kernel(const double* __restrict__ Vall, const double* __restrict__ call,
                   const int* __restrict__ elem_ds, const int* __restrict__ elem_kind,
                   const double* __restrict__ Dall, const int* __restrict__ elem_of_chain,
                   const double* __restrict__ alpha, const double* __restrict__ H,
                   double* __restrict__ out, int n_alpha, int nw, int nwp, int ns)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int NP = 16 * NT, LD = NP + 1;
    extern __shared__ double sm[];
    double* Bm = sm;                 // [NP][LD]
    double* wsh = Bm + NP * LD;      // [nwp]
    double* red = wsh + nwp;         // [4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t prob = blockIdx.x;
    const int e = elem_of_chain[prob / n_alpha];
    const int ds = elem_ds[e], kind = elem_kind[e];
    const double a = alpha[prob];
    const double* V = Vall + (size_t)ds * nwp * NP;
    const double* cc = call + (size_t)ds * NP;
    const double* Hp = H + prob * nw;
    const double* Dp = Dall + (size_t)e * nwp;
    for (int i = tid; i < nwp; i += 256) {
        double w = 0.0;
        if (i < nw) {
            const double h = Hp[i];
            if (kind == 0) w = h;
            else { const double d2 = 2.0 * Dp[i]; w = sqrt(fma(h, h, d2 * d2)); }
        }
        wsh[i] = w;
    }
    for (int i = tid; i < NP * LD; i += 256) Bm[i] = 0.0;
    __syncthreads();
    const int kq = lane >> 4, cn = lane & 15;
    const int n_groups = nwp >> 2;
    const int ntile = (ns + 15) >> 4;            // tile rows / columns that hold data
    for (int mt = 0; mt < ntile; ++mt) {
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        for (int g = wave; g < n_groups; g += 4) {
            const double* row = V + (size_t)(4 * g + kq) * NP + cn;
            const double wq = wsh[4 * g + kq];
            const double am = row[16 * mt] * wq;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t >= mt && t < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, row[16 * t], acc[t], 0, 0, 0);
        }
        // the four waves add their partial tiles one after the other (fixed order)
        for (int ph = 0; ph < 4; ++ph) {
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
    // B = c W c + a I on the upper triangle (row <= col), mirrored to the lower one
    for (int idx = tid; idx < ns * ns; idx += 256) {
        const int i = idx / ns, j = idx % ns;
        if (i <= j) {
            double b = cc[i] * Bm[i * LD + j] * cc[j];
            if (i == j) b += a;
            Bm[j * LD + i] = b;          // lower triangle: row j >= col i
        }
    }
    __syncthreads();
    // right-looking Cholesky on the lower triangle
    double logsum = 0.0;
    bool ok = true;
    for (int j = 0; j < ns; ++j) {
        const double piv = Bm[j * LD + j];
        if (!(piv > 0.0)) ok = false;
        const double d = sqrt(piv);
        logsum += log(d);
        __syncthreads();                         // everybody has read the pivot
        for (int i = j + 1 + tid; i < ns; i += 256) Bm[i * LD + j] /= d;
        __syncthreads();
        const int m = ns - j - 1;
        for (int idx = tid; idx < m * m; idx += 256) {
            const int i = j + 1 + idx / m, k = j + 1 + idx % m;
            if (k <= i) Bm[i * LD + k] = fma(-Bm[i * LD + j], Bm[k * LD + j], Bm[i * LD + k]);
        }
        __syncthreads();
    }
    if (tid == 0) out[prob] = ok ? 2.0 * logsum - ns * log(a) : __builtin_nan("");
    (void)red;
}
} // namespace mxe

extern "C" int mxe_logdet(mxe_ctx* ctx, double* out_logdet)
try {
    if (!ctx || !out_logdet) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    const int NP = ctx->NP;
    HIPCHK(ctx, ctx->dlogdet.ensure(P));
    HIPCHK(ctx, ctx->dparent_elem.ensure(ctx->chain_elem.size()));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dparent_elem.p, ctx->chain_elem.data(), ctx->chain_elem.size() * sizeof(int),
                               hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = ((size_t)NP * (NP + 1) + ctx->nwp + 4) * sizeof(double);
    if (lds > 160 * 1024) return MXE_ERR_LIMIT;
    hipError_t e;
#define MXE_LAUNCH_LOGDET(NT_) do { \
        e = hipFuncSetAttribute((const void*)mxe::logdet_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) { hipLaunchKernelGGL((mxe::logdet_kernel<NT_>), dim3((unsigned)P), dim3(256), lds, ctx->stream, \
            ctx->dV.p, ctx->dc.p, ctx->delem_ds.p, ctx->delem_kind.p, ctx->dD.p, ctx->dparent_elem.p, ctx->dalpha.p, \
            ctx->dout_H.p, ctx->dlogdet.p, ctx->n_alpha, ctx->n_omega, ctx->nwp, ctx->n_s); e = hipGetLastError(); } } while (0)
    if (NP == 64) MXE_LAUNCH_LOGDET(4); else MXE_LAUNCH_LOGDET(8);
#undef MXE_LAUNCH_LOGDET
    HIPCHK(ctx, e);
    HIPCHK(ctx, hipMemcpyAsync(out_logdet, ctx->dlogdet.p, P * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    return MXE_OK;
}
MXE_CATCH_ALL

// stream + device buffers of a context-free entry point, released on every path
namespace {
struct SvdScratch {
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<void*> bufs;
    template <typename T> hipError_t alloc(T** p, size_t count) {
        hipError_t e = hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) bufs.push_back((void*)*p);
        return e;
    }
    ~SvdScratch() {
        for (void* b : bufs) hipFree(b);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        if (stream) hipStreamDestroy(stream);
    }
};
#define SVDCHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { \
    fprintf(stderr, "[mxe] %s: %s\n", #call, hipGetErrorString(e__)); return MXE_ERR_HIP; } } while (0)
} // namespace

// ---- cost function and derivatives at caller-supplied points; audit of a launch ----------
namespace {
size_t eval_lds_bytes(int NP, int nwp) { return ((size_t)NP * (NP + 1) + 2 * (size_t)nwp + 7 * (size_t)NP + 16) * sizeof(double); }

int launch_eval(mxe_ctx* ctx, const mxe::EvalParams& ep, size_t P)
{
    const size_t lds = eval_lds_bytes(ctx->NP, ctx->nwp);
    if (lds > 160 * 1024) return MXE_ERR_LIMIT;
    hipError_t e;
#define MXE_LAUNCH_EVAL(NT_) do { \
        e = hipFuncSetAttribute((const void*)mxe::eval_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) { hipLaunchKernelGGL((mxe::eval_kernel<NT_>), dim3((unsigned)P), dim3(256), lds, ctx->stream, ep); \
                               e = hipGetLastError(); } } while (0)
    if (ctx->NP == 64) MXE_LAUNCH_EVAL(4); else MXE_LAUNCH_EVAL(8);
#undef MXE_LAUNCH_EVAL
    HIPCHK(ctx, e);
    return MXE_OK;
}

void eval_params_base(mxe_ctx* ctx, mxe::EvalParams& ep)
{
    std::memset(&ep, 0, sizeof(ep));
    ep.nw = ctx->n_omega; ep.nwp = ctx->nwp; ep.ns = ctx->n_s; ep.NP = ctx->NP;
    ep.V = ctx->dV.p; ep.Vt = ctx->dVt.p; ep.c = ctx->dc.p;
    ep.elem_ds = ctx->delem_ds.p; ep.elem_kind = ctx->delem_kind.p;
    ep.ghat = ctx->dghat.p; ep.cperp = ctx->dcperp.p; ep.D = ctx->dD.p;
    ep.eta = 1.0; ep.elem_div = 1;
}
} // namespace

extern "C" int mxe_eval_batch(mxe_ctx* ctx, int P, const int32_t* elem_of_problem, const double* alpha_scaled,
                              const double* x, int input_is_H, double chi2_factor,
                              double* out_Q, double* out_chi2, double* out_S,
                              double* out_H, double* out_u, double* out_w, double* out_q,
                              double* out_h, double* out_g, double* out_W, double* out_W2)
try {
    if (!ctx || P < 1 || !elem_of_problem || !alpha_scaled || !x) return MXE_ERR_ARG;
    if (ctx->n_elem < 1) return MXE_ERR_STATE;
    if (!(chi2_factor > 0.0) || !std::isfinite(chi2_factor)) return MXE_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->ds_dirty) { int rc = upload_bases(ctx); if (rc != MXE_OK) return rc; }
    const int ns = ctx->n_s, NP = ctx->NP, nw = ctx->n_omega;
    const size_t xs = input_is_H ? (size_t)nw : (size_t)NP;
    std::vector<double> hx((size_t)P * xs, 0.0);
    for (int p = 0; p < P; ++p) {
        const int e = elem_of_problem[p];
        if (e < 0 || e >= ctx->n_elem) return MXE_ERR_ARG;
        if (!(alpha_scaled[p] >= 0.0) || !std::isfinite(alpha_scaled[p])) return MXE_ERR_ARG;
        if (input_is_H) { std::copy(x + (size_t)p * nw, x + (size_t)(p + 1) * nw, hx.begin() + (size_t)p * nw); continue; }
        const DataSet& DS = ctx->ds[ctx->elem_ds[e]];
        for (int k = 0; k < ns; ++k) {
            double s;
            if (DS.identity_q) s = x[(size_t)p * ns + k];
            else { s = 0.0; for (int j = 0; j < ns; ++j) s += DS.Q[(size_t)j * ns + k] * x[(size_t)p * ns + j]; }   // v' = Q^T v
            hx[(size_t)p * NP + k] = s;
        }
    }
    const bool wantW = out_W != nullptr, wantW2 = out_W2 != nullptr;
    HIPCHK(ctx, ctx->ev_x.ensure(hx.size()));
    HIPCHK(ctx, ctx->ev_alpha.ensure(P));
    HIPCHK(ctx, ctx->ev_elem.ensure(P));
    HIPCHK(ctx, ctx->ev_scal.ensure((size_t)5 * P));
    HIPCHK(ctx, ctx->ev_vecw.ensure((size_t)4 * P * nw));
    HIPCHK(ctx, ctx->ev_vecs.ensure((size_t)2 * P * NP));
    HIPCHK(ctx, ctx->ev_mat.ensure((size_t)(wantW ? 1 : 0) * P * NP * NP + (size_t)(wantW2 ? 1 : 0) * P * NP * NP));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ev_x.p, hx.data(), hx.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ev_alpha.p, alpha_scaled, (size_t)P * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ev_elem.p, elem_of_problem, (size_t)P * 4, hipMemcpyHostToDevice, ctx->stream));
    mxe::EvalParams ep;
    eval_params_base(ctx, ep);
    ep.elem = ctx->ev_elem.p; ep.alpha = ctx->ev_alpha.p; ep.x = ctx->ev_x.p; ep.x_stride = (int)xs;
    ep.input_is_H = input_is_H ? 1 : 0; ep.eta = chi2_factor;
    ep.want_gram = wantW; ep.want_gram2 = wantW2;
    ep.Q = ctx->ev_scal.p; ep.chi2 = ep.Q + P; ep.S = ep.chi2 + P;
    ep.H = ctx->ev_vecw.p; ep.u = ep.H + (size_t)P * nw; ep.w = ep.u + (size_t)P * nw;
    ep.q = out_q ? ep.w + (size_t)P * nw : nullptr;
    ep.h = ctx->ev_vecs.p; ep.g = ep.h + (size_t)P * NP;
    ep.W = wantW ? ctx->ev_mat.p : nullptr;
    ep.W2 = wantW2 ? ctx->ev_mat.p + (size_t)(wantW ? 1 : 0) * P * NP * NP : nullptr;
    int rc = launch_eval(ctx, ep, (size_t)P);
    if (rc != MXE_OK) return rc;
    if (out_Q) HIPCHK(ctx, hipMemcpyAsync(out_Q, ep.Q, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_chi2) HIPCHK(ctx, hipMemcpyAsync(out_chi2, ep.chi2, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_S) HIPCHK(ctx, hipMemcpyAsync(out_S, ep.S, (size_t)P * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_H) HIPCHK(ctx, hipMemcpyAsync(out_H, ep.H, (size_t)P * nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_u) HIPCHK(ctx, hipMemcpyAsync(out_u, ep.u, (size_t)P * nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_w) HIPCHK(ctx, hipMemcpyAsync(out_w, ep.w, (size_t)P * nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_q) HIPCHK(ctx, hipMemcpyAsync(out_q, ep.q, (size_t)P * nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> hh, hg, hW, hW2;
    if (out_h) { hh.resize((size_t)P * NP); HIPCHK(ctx, hipMemcpyAsync(hh.data(), ep.h, hh.size() * 8, hipMemcpyDeviceToHost, ctx->stream)); }
    if (out_g) { hg.resize((size_t)P * NP); HIPCHK(ctx, hipMemcpyAsync(hg.data(), ep.g, hg.size() * 8, hipMemcpyDeviceToHost, ctx->stream)); }
    if (wantW) { hW.resize((size_t)P * NP * NP); HIPCHK(ctx, hipMemcpyAsync(hW.data(), ep.W, hW.size() * 8, hipMemcpyDeviceToHost, ctx->stream)); }
    if (wantW2) { hW2.resize((size_t)P * NP * NP); HIPCHK(ctx, hipMemcpyAsync(hW2.data(), ep.W2, hW2.size() * 8, hipMemcpyDeviceToHost, ctx->stream)); }
    HIPCHK(ctx, stream_wait(ctx->stream));
    // whitened basis -> caller basis:  x = Q x',  X = Q X' Q^T
    std::vector<double> tmp((size_t)ns * ns);
    for (int p = 0; p < P; ++p) {
        const DataSet& DS = ctx->ds[ctx->elem_ds[elem_of_problem[p]]];
        auto vec_out = [&](const std::vector<double>& src, double* dst) {
            for (int k = 0; k < ns; ++k) {
                double s;
                if (DS.identity_q) s = src[(size_t)p * NP + k];
                else { s = 0.0; for (int j = 0; j < ns; ++j) s += DS.Q[(size_t)k * ns + j] * src[(size_t)p * NP + j]; }
                dst[(size_t)p * ns + k] = s;
            }
        };
        auto mat_out = [&](const std::vector<double>& src, double* dst) {
            const double* Xp = src.data() + (size_t)p * NP * NP;
            double* out = dst + (size_t)p * ns * ns;
            if (DS.identity_q) {
                for (int i = 0; i < ns; ++i) for (int j = 0; j < ns; ++j) out[(size_t)i * ns + j] = Xp[(size_t)i * NP + j];
                return;
            }
            for (int i = 0; i < ns; ++i) for (int j = 0; j < ns; ++j) {        // tmp = X' Q^T
                double s = 0.0;
                for (int k = 0; k < ns; ++k) s += Xp[(size_t)i * NP + k] * DS.Q[(size_t)j * ns + k];
                tmp[(size_t)i * ns + j] = s;
            }
            for (int i = 0; i < ns; ++i) for (int j = 0; j < ns; ++j) {        // out = Q tmp
                double s = 0.0;
                for (int k = 0; k < ns; ++k) s += DS.Q[(size_t)i * ns + k] * tmp[(size_t)k * ns + j];
                out[(size_t)i * ns + j] = s;
            }
        };
        if (out_h) vec_out(hh, out_h);
        if (out_g) vec_out(hg, out_g);
        if (wantW) mat_out(hW, out_W);
        if (wantW2) mat_out(hW2, out_W2);
    }
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_entropy(int device, int kind, int n_omega, int P, const double* H, const double* D,
                           double* out_S, double* out_dS, double* out_ddS)
try {
    if (n_omega < 1 || P < 1 || !H || !D || (kind != MXE_ENTROPY_NORMAL && kind != MXE_ENTROPY_PLUSMINUS)) return MXE_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MXE_ERR_NODEVICE;
    if (device < 0 || device >= ndev) return MXE_ERR_ARG;
    SvdScratch sc;                    // stream + buffers released on every path
    SVDCHK(hipSetDevice(device));
    SVDCHK(hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking));
    const size_t n = n_omega;
    double *dH, *dD, *dS, *dd, *ddd;
    SVDCHK(sc.alloc(&dH, (size_t)P * n)); SVDCHK(sc.alloc(&dD, n)); SVDCHK(sc.alloc(&dS, (size_t)P));
    SVDCHK(sc.alloc(&dd, (size_t)P * n)); SVDCHK(sc.alloc(&ddd, (size_t)P * n));
    SVDCHK(hipMemcpyAsync(dH, H, (size_t)P * n * 8, hipMemcpyHostToDevice, sc.stream));
    SVDCHK(hipMemcpyAsync(dD, D, n * 8, hipMemcpyHostToDevice, sc.stream));
    hipLaunchKernelGGL(mxe::entropy_kernel, dim3((unsigned)P), dim3(256), 0, sc.stream, kind, n_omega, dH, dD, dS, dd, ddd);
    SVDCHK(hipGetLastError());
    if (out_S) SVDCHK(hipMemcpyAsync(out_S, dS, (size_t)P * 8, hipMemcpyDeviceToHost, sc.stream));
    if (out_dS) SVDCHK(hipMemcpyAsync(out_dS, dd, (size_t)P * n * 8, hipMemcpyDeviceToHost, sc.stream));
    if (out_ddS) SVDCHK(hipMemcpyAsync(out_ddS, ddd, (size_t)P * n * 8, hipMemcpyDeviceToHost, sc.stream));
    SVDCHK(hipStreamSynchronize(sc.stream));
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_audit(mxe_ctx* ctx, double* out_corr, double* out_gmax)
try {
    if (!ctx || (!out_corr && !out_gmax)) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    HIPCHK(ctx, ctx->ev_scal.ensure(2 * P));
    HIPCHK(ctx, ctx->dparent_elem.ensure(ctx->chain_elem.size()));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dparent_elem.p, ctx->chain_elem.data(), ctx->chain_elem.size() * sizeof(int),
                               hipMemcpyHostToDevice, ctx->stream));
    mxe::EvalParams ep;
    eval_params_base(ctx, ep);
    ep.elem = ctx->dparent_elem.p; ep.elem_div = ctx->n_alpha;
    ep.alpha = ctx->dalpha.p;                 // alpha / eta, as the chain kernel iterated on it
    ep.x = ctx->dout_v.p; ep.x_stride = ctx->NP;
    ep.want_audit = 1;
    ep.corr = ctx->ev_scal.p; ep.gmax = ctx->ev_scal.p + P;
    int rc = launch_eval(ctx, ep, P);
    if (rc != MXE_OK) return rc;
    if (out_corr) HIPCHK(ctx, hipMemcpyAsync(out_corr, ep.corr, P * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_gmax) HIPCHK(ctx, hipMemcpyAsync(out_gmax, ep.gmax, P * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    return MXE_OK;
}
MXE_CATCH_ALL

// ---- the default analyzer's alpha on the device, and the compact result pack ----------------
extern "C" int mxe_select_launch(mxe_ctx* ctx, int p2_deg)
try {
    if (!ctx || (p2_deg != 0 && p2_deg != 1)) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    double* sel = ctx->dout_Q.p + P;
    double* idx = sel + (size_t)ctx->n_chain * ctx->n_omega;
    hipLaunchKernelGGL(mxe::linefit_kernel, dim3(ctx->n_chain), dim3(256), ((size_t)11 * ctx->n_alpha + 6 + 28) * sizeof(double), ctx->stream,
                       ctx->dalpha.p, ctx->dout_chi2.p, ctx->dout_H.p, ctx->n_alpha, ctx->n_omega, p2_deg, sel, idx);
    HIPCHK(ctx, hipGetLastError());
    return MXE_OK;
}
MXE_CATCH_ALL

// The three default analyzers of the reference in one small launch: the line fit as above (its row and index also land in
// the compact pack), the alpha of the largest curvature of log10 chi2 over gamma log10 alpha, the alpha of the flattest
// entropy -- indices and the three H rows per scan in a buffer of their own, fetched with ONE copy (mxe_select3_fetch).
extern "C" int mxe_select3_launch(mxe_ctx* ctx, int p2_deg, double gamma)
try {
    if (!ctx || (p2_deg != 0 && p2_deg != 1) || !(gamma > 0.0)) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha, nc = ctx->n_chain, nw = ctx->n_omega;
    double* sel = ctx->dout_Q.p + P;
    double* idx = sel + nc * nw;
    HIPCHK(ctx, ctx->dsel3.ensure(3 * nc * (nw + 1)));
    ctx->pre_rows = nullptr; ctx->pre_index = false;
    hipLaunchKernelGGL(mxe::linefit_kernel, dim3(ctx->n_chain), dim3(256), ((size_t)11 * ctx->n_alpha + 6 + 28) * sizeof(double), ctx->stream,
                       ctx->dalpha.p, ctx->dout_chi2.p, ctx->dout_H.p, ctx->n_alpha, ctx->n_omega, p2_deg, sel, idx,
                       ctx->dout_S.p, gamma, ctx->dsel3.p, ctx->dsel3.p + 3 * nc);
    HIPCHK(ctx, hipGetLastError());
    ctx->sel3_nc = ctx->n_chain;
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_select3_fetch_rows(mxe_ctx* ctx, int32_t* out_index /*[3][n_chain] or NULL*/, int first, int count,
                                      double* out_H_selected /*[count][n_chain][n_omega] or NULL*/)
try {
    if (!ctx || first < 0 || count < 0 || first + count > 3 || (!out_index && !out_H_selected)) return MXE_ERR_ARG;
    if (!ctx->launched || !ctx->dsel3.p || ctx->sel3_nc != ctx->n_chain) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nc = ctx->n_chain, nw = ctx->n_omega;
    if (ctx->pre_index && ctx->h_sel3_pinned_n >= 3 * nc && (count == 0 || (out_H_selected == ctx->pre_rows && first == ctx->pre_first && count == ctx->pre_count))) {
        // (mxe_select3_prefetch_rows queued these copies behind the selection kernel)
        HIPCHK(ctx, stream_wait(ctx->stream));
        if (out_index) for (size_t i = 0; i < 3 * nc; ++i) out_index[i] = (int32_t)ctx->h_sel3_pinned[i];
        return MXE_OK;
    }
    if (out_index) {
        ctx->h_sel3.resize(3 * nc);
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_sel3.data(), ctx->dsel3.p, 3 * nc * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, stream_wait(ctx->stream));       // (the selection kernel has finished before any path of the row copy starts)
    if (out_H_selected && count > 0)
        HIPCHK(ctx, d2h_pipelined(out_H_selected, ctx->dsel3.p + 3 * nc + (size_t)first * nc * nw, (size_t)count * nc * nw * 8, ctx->stream));
    if (out_index) for (size_t i = 0; i < 3 * nc; ++i) out_index[i] = (int32_t)ctx->h_sel3[i];
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_select3_prefetch_rows(mxe_ctx* ctx, int first, int count, double* out_H_selected /*[count][n_chain][n_omega] or NULL*/)
try {
    if (!ctx || first < 0 || count < 0 || first + count > 3 || (count > 0 && !out_H_selected)) return MXE_ERR_ARG;
    if (!ctx->launched || !ctx->dsel3.p || ctx->sel3_nc != ctx->n_chain) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nc = ctx->n_chain, nw = ctx->n_omega;
    ctx->pre_rows = nullptr; ctx->pre_index = false;
    if (ctx->h_sel3_pinned_n < 3 * nc) {
        if (ctx->h_sel3_pinned) { (void)hipHostFree(ctx->h_sel3_pinned); ctx->h_sel3_pinned = nullptr; ctx->h_sel3_pinned_n = 0; }
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_sel3_pinned, 3 * nc * 8, hipHostMallocDefault));
        ctx->h_sel3_pinned_n = 3 * nc;
    }
    const size_t row_bytes = (size_t)count * nc * nw * 8;
    const bool mapped = count == 0 || g_host.holds(out_H_selected, row_bytes);
    if (mapped)
        HIPCHK(ctx, copy_out(ctx->h_sel3_pinned, ctx->dsel3.p, 3 * nc * 8, out_H_selected, ctx->dsel3.p + 3 * nc + (size_t)first * nc * nw, row_bytes, true, ctx->stream));
    else {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_sel3_pinned, ctx->dsel3.p, 3 * nc * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out_H_selected, ctx->dsel3.p + 3 * nc + (size_t)first * nc * nw, row_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    ctx->pre_index = true; ctx->pre_rows = count > 0 ? out_H_selected : nullptr; ctx->pre_first = first; ctx->pre_count = count;
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_select3_fetch(mxe_ctx* ctx, int32_t* out_index /*[3][n_chain]*/, double* out_H_selected /*[3][n_chain][n_omega]*/)
{
    if (!ctx || !out_index) return MXE_ERR_ARG;
    return mxe_select3_fetch_rows(ctx, out_index, 0, out_H_selected ? 3 : 0, out_H_selected);
}

extern "C" int mxe_select_fetch(mxe_ctx* ctx, int32_t* out_index, double* out_H_selected)
try {
    if (!ctx) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha, nc = ctx->n_chain, nw = ctx->n_omega;
    const double* sel = ctx->dout_Q.p + P;
    HIPCHK(ctx, stream_wait(ctx->stream));
    if (out_H_selected) HIPCHK(ctx, hipMemcpy(out_H_selected, sel, nc * nw * 8, hipMemcpyDeviceToHost));
    if (out_index) {
        std::vector<double> hi(nc);
        HIPCHK(ctx, hipMemcpy(hi.data(), sel + nc * nw, nc * 8, hipMemcpyDeviceToHost));
        for (size_t c = 0; c < nc; ++c) out_index[c] = (int32_t)hi[c];
    }
    return MXE_OK;
}
MXE_CATCH_ALL

namespace {
// one workgroup per requested row: out[r][:] = H[idx[r]][:]
__global__ __launch_bounds__(256)
void rows_gather_kernel(const double* __restrict__ H, const int* __restrict__ idx, int n_omega, double* __restrict__ out)
{
    const double* src = H + (size_t)idx[blockIdx.x] * n_omega;
    double* dst = out + (size_t)blockIdx.x * n_omega;
    for (int i = threadIdx.x; i < n_omega; i += 256) dst[i] = src[i];
}
}  // namespace

extern "C" int mxe_fetch_rows(mxe_ctx* ctx, int n_rows, const int32_t* problem_index, double* out_H)
try {
    if (!ctx || n_rows < 1 || !problem_index || !out_H) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha, nw = ctx->n_omega;
    for (int r = 0; r < n_rows; ++r)
        if (problem_index[r] < 0 || (size_t)problem_index[r] >= P) return MXE_ERR_ARG;
    // the rows are collected on the device and come over in ONE copy (a copy per row cost 7 us each: 5 ms for the
    // 768 rows the analyzers of a 16 x 16 job pick)
    HIPCHK(ctx, ctx->rows_idx.ensure(n_rows));
    HIPCHK(ctx, ctx->rows_out.ensure((size_t)n_rows * nw));
    HIPCHK(ctx, hipMemcpyAsync(ctx->rows_idx.p, problem_index, (size_t)n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(rows_gather_kernel, dim3(n_rows), dim3(256), 0, ctx->stream,
                       (const double*)ctx->dout_H.p, (const int*)ctx->rows_idx.p, (int)nw, ctx->rows_out.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out_H, ctx->rows_out.p, (size_t)n_rows * nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    return MXE_OK;
}
MXE_CATCH_ALL

// ---- gather between GPUs: RCCL over xGMI, called directly (no framework in between) --------
#include <rccl/rccl.h>
#include <dlfcn.h>
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) return false;
#define MXE_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); if (!field) { dlclose(lib); lib = nullptr; return false; }
        MXE_SYM(GetUniqueId, "ncclGetUniqueId"); MXE_SYM(CommInitRank, "ncclCommInitRank"); MXE_SYM(CommInitAll, "ncclCommInitAll");
        MXE_SYM(CommDestroy, "ncclCommDestroy"); MXE_SYM(Send, "ncclSend"); MXE_SYM(Recv, "ncclRecv");
        MXE_SYM(AllReduce, "ncclAllReduce"); MXE_SYM(GroupStart, "ncclGroupStart"); MXE_SYM(GroupEnd, "ncclGroupEnd"); MXE_SYM(GetErrorString, "ncclGetErrorString");
#undef MXE_SYM
        return true;
    }
};
Rccl g_rccl;
#define NCCLCHK(ctx, call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { \
    (ctx)->hip_err = std::string(#call) + ": " + g_rccl.GetErrorString(r__); return MXE_ERR_HIP; } } while (0)
} // namespace

struct mxe_comm_state {
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    bool copy_transport = false;                 // ranks of ONE process on one device: plain device copies
    bool loopback = false;                       // mxe_comm_set_loopback: the root's own pack goes through ncclSend / ncclRecv to itself
    std::vector<mxe_ctx*> local;                 // ... the contexts of that process, by rank
    DevBuf<double> recv;                         // root: [sum of counts]
    DevBuf<double> small;                        // mxe_comm_allreduce
};

extern "C" int mxe_comm_unique_id(char* out_id128)
{
    if (!out_id128) return MXE_ERR_ARG;
    if (!g_rccl.load()) return MXE_ERR_NODEVICE;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return MXE_ERR_HIP;
    std::memcpy(out_id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return MXE_OK;
}

extern "C" int mxe_comm_init(mxe_ctx* ctx, int n_ranks, int rank, const char* id128)
try {
    if (!ctx || n_ranks < 1 || rank < 0 || rank >= n_ranks || !id128) return MXE_ERR_ARG;
    if (!g_rccl.load()) { ctx->hip_err = "librccl.so.1 could not be loaded"; return MXE_ERR_NODEVICE; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    comm_release(ctx);
    ctx->comm = new mxe_comm_state();
    ctx->comm->n_ranks = n_ranks; ctx->comm->rank = rank;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    NCCLCHK(ctx, g_rccl.CommInitRank(&ctx->comm->comm, n_ranks, id, rank));
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_comm_init_local(mxe_ctx** ctxs, int n)
try {
    if (!ctxs || n < 1) return MXE_ERR_ARG;
    for (int r = 0; r < n; ++r) if (!ctxs[r]) return MXE_ERR_ARG;
    bool distinct = true;
    for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) if (ctxs[a]->device == ctxs[b]->device) distinct = false;
    std::vector<ncclComm_t> comms(n, nullptr);
    if (distinct && n > 1) {
        if (!g_rccl.load()) { ctxs[0]->hip_err = "librccl.so.1 could not be loaded"; return MXE_ERR_NODEVICE; }
        std::vector<int> devs(n);
        for (int r = 0; r < n; ++r) devs[r] = ctxs[r]->device;
        // xGMI peers reach each other directly; where a pair cannot (hipDeviceCanAccessPeer), RCCL stages through the host --
        // slower, still correct.  Said once on stderr; an error only when the caller insists (MXE_REQUIRE_PEER_ACCESS)
        for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) {
            int can = 1;
            if (a != b && hipDeviceCanAccessPeer(&can, devs[a], devs[b]) == hipSuccess && !can) {
                if (getenv("MXE_REQUIRE_PEER_ACCESS")) {
                    ctxs[0]->hip_err = "device " + std::to_string(devs[a]) + " cannot access device " + std::to_string(devs[b]) + " directly (hipDeviceCanAccessPeer)";
                    return MXE_ERR_HIP;
                }
                static bool said = false;
                if (!said) { said = true; fprintf(stderr, "[mxe] devices %d and %d have no direct peer access: the gather goes through host memory\n", devs[a], devs[b]); }
            }
        }
        NCCLCHK(ctxs[0], g_rccl.CommInitAll(comms.data(), n, devs.data()));
    }
    for (int r = 0; r < n; ++r) {
        comm_release(ctxs[r]);
        ctxs[r]->comm = new mxe_comm_state();
        ctxs[r]->comm->n_ranks = n; ctxs[r]->comm->rank = r;
        ctxs[r]->comm->comm = comms[r];
        ctxs[r]->comm->copy_transport = !(distinct && n > 1);
        ctxs[r]->comm->local.assign(ctxs, ctxs + n);
    }
    return MXE_OK;
}
MXE_CATCH_ALL

static void comm_release(mxe_ctx* ctx)
{
    if (!ctx || !ctx->comm) return;
    if (ctx->comm->comm && g_rccl.lib) g_rccl.CommDestroy(ctx->comm->comm);
    ctx->comm->recv.release();
    ctx->comm->small.release();
    delete ctx->comm;
    ctx->comm = nullptr;
}

// Test plumbing for a box with ONE GPU: with it on, the root's own pack travels through ncclSend / ncclRecv to itself
// (inside ncclGroupStart / ncclGroupEnd) instead of a device copy and mxe_comm_allreduce runs ncclAllReduce with the
// one rank -- every RCCL call of the multi-GPU path executes.  Communicators of mxe_comm_init only.
extern "C" int mxe_comm_set_loopback(mxe_ctx* ctx, int on)
{
    if (!ctx) return MXE_ERR_ARG;
    if (!ctx->comm || !ctx->comm->comm || !ctx->comm->local.empty()) return MXE_ERR_STATE;
    ctx->comm->loopback = on != 0;
    return MXE_OK;
}

extern "C" int mxe_comm_destroy(mxe_ctx* ctx)
{
    if (!ctx) return MXE_ERR_ARG;
    hipSetDevice(ctx->device);
    comm_release(ctx);
    return MXE_OK;
}

// what a rank contributes: the compact pack (chi2, S, Q of every alpha + the H row and index of the
// analyzer's alpha per scan) or the full one (all H in front of it)
static void gather_span(mxe_ctx* ctx, int what, double** p, size_t* count)
{
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha, nw = ctx->n_omega, nc = ctx->n_chain;
    const size_t compact = 3 * P + nc * (nw + 1);
    if (what == MXE_GATHER_FULL) { *p = ctx->dout_H.p; *count = P * nw + compact; }
    else { *p = ctx->dout_chi2.p; *count = compact; }
}

// enqueue this rank's part of the gather on its stream (inside a group when called for several local ranks)
static int gather_enqueue(mxe_ctx* ctx, int root, int what, const int64_t* counts)
{
    mxe_comm_state* cm = ctx->comm;
    double* mine; size_t n_mine;
    gather_span(ctx, what, &mine, &n_mine);
    if ((int64_t)n_mine != counts[cm->rank]) return MXE_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (cm->rank == root) {
        // (the root's receive buffer was sized by the caller BEFORE ncclGroupStart: gather_reserve -- an allocation, i.e. a
        //  possible device synchronisation, has no place between the calls of a group)
        size_t off = 0;
        for (int r = 0; r < cm->n_ranks; ++r) {
            if (r == root && cm->loopback && !cm->copy_transport) {
                // (send and receive of one rank to itself, matched inside the caller's group: the same calls a peer's pack takes)
                NCCLCHK(ctx, g_rccl.Send(mine, n_mine, ncclDouble, root, cm->comm, ctx->stream));
                NCCLCHK(ctx, g_rccl.Recv(cm->recv.p + off, n_mine, ncclDouble, root, cm->comm, ctx->stream));
            }
            else if (r == root) HIPCHK(ctx, hipMemcpyAsync(cm->recv.p + off, mine, n_mine * 8, hipMemcpyDeviceToDevice, ctx->stream));
            else if (!cm->copy_transport) NCCLCHK(ctx, g_rccl.Recv(cm->recv.p + off, (size_t)counts[r], ncclDouble, r, cm->comm, ctx->stream));
            off += (size_t)counts[r];
        }
    } else if (!cm->copy_transport) {
        NCCLCHK(ctx, g_rccl.Send(mine, n_mine, ncclDouble, root, cm->comm, ctx->stream));
    }
    return MXE_OK;
}

// the root's receive buffer for a gather of these counts (outside any RCCL group)
static int gather_reserve(mxe_ctx* root_ctx, int n_ranks, const int64_t* counts)
{
    size_t total = 0;
    for (int r = 0; r < n_ranks; ++r) { if (counts[r] < 0) return MXE_ERR_ARG; total += (size_t)counts[r]; }
    HIPCHK(root_ctx, hipSetDevice(root_ctx->device));
    HIPCHK(root_ctx, root_ctx->comm->recv.ensure(total));
    return MXE_OK;
}

extern "C" int mxe_gather(mxe_ctx* ctx, int root, int what, const int64_t* counts, double* recv_host)
try {
    if (!ctx || !counts || (what != MXE_GATHER_COMPACT && what != MXE_GATHER_FULL)) return MXE_ERR_ARG;
    if (!ctx->comm || !ctx->launched) return MXE_ERR_STATE;
    mxe_comm_state* cm = ctx->comm;
    if (root < 0 || root >= cm->n_ranks) return MXE_ERR_ARG;
    if (!cm->local.empty()) return MXE_ERR_STATE;            // ranks of one process gather together: mxe_gather_local
    if (cm->rank == root) { const int rr = gather_reserve(ctx, cm->n_ranks, counts); if (rr != MXE_OK) return rr; }
    const bool grouped = cm->n_ranks > 1 || cm->loopback;
    if (grouped) NCCLCHK(ctx, g_rccl.GroupStart());
    const int rc = gather_enqueue(ctx, root, what, counts);
    if (grouped) NCCLCHK(ctx, g_rccl.GroupEnd());
    if (rc != MXE_OK) return rc;
    if (recv_host && cm->rank == root) {
        size_t total = 0;
        for (int r = 0; r < cm->n_ranks; ++r) total += (size_t)counts[r];
        HIPCHK(ctx, hipMemcpyAsync(recv_host, cm->recv.p, total * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, stream_wait(ctx->stream));
    }
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_gather_local(mxe_ctx** ctxs, int n, int root, int what, const int64_t* counts, double* recv_host)
try {
    if (!ctxs || n < 1 || root < 0 || root >= n || !counts || (what != MXE_GATHER_COMPACT && what != MXE_GATHER_FULL)) return MXE_ERR_ARG;
    for (int r = 0; r < n; ++r)
        if (!ctxs[r] || !ctxs[r]->comm || ctxs[r]->comm->n_ranks != n || ctxs[r]->comm->rank != r || !ctxs[r]->launched) return MXE_ERR_STATE;
    mxe_ctx* rt = ctxs[root];
    const bool copy = rt->comm->copy_transport;
    { const int rr = gather_reserve(rt, n, counts); if (rr != MXE_OK) return rr; }
    if (copy) {
        // ranks that share a device (or a single rank): the root waits for the others' launches with events
        // and copies their packs itself
        for (int r = 0; r < n; ++r) {
            if (r == root) continue;
            HIPCHK(ctxs[r], hipSetDevice(ctxs[r]->device));
            HIPCHK(ctxs[r], hipEventRecord(ctxs[r]->ev_mark, ctxs[r]->stream));
            HIPCHK(rt, hipSetDevice(rt->device));
            HIPCHK(rt, hipStreamWaitEvent(rt->stream, ctxs[r]->ev_mark, 0));
        }
    } else {
        NCCLCHK(rt, g_rccl.GroupStart());
    }
    int rc = MXE_OK;
    for (int r = 0; r < n && rc == MXE_OK; ++r) rc = gather_enqueue(ctxs[r], root, what, counts);
    if (!copy) NCCLCHK(rt, g_rccl.GroupEnd());
    if (rc != MXE_OK) return rc;
    if (copy) {
        HIPCHK(rt, hipSetDevice(rt->device));
        size_t off = 0;
        for (int r = 0; r < n; ++r) {
            if (r != root) {
                double* src; size_t cnt;
                gather_span(ctxs[r], what, &src, &cnt);
                HIPCHK(rt, hipMemcpyAsync(rt->comm->recv.p + off, src, cnt * 8, hipMemcpyDeviceToDevice, rt->stream));
            }
            off += (size_t)counts[r];
        }
    }
    if (recv_host) {
        size_t total = 0;
        for (int r = 0; r < n; ++r) total += (size_t)counts[r];
        HIPCHK(rt, hipSetDevice(rt->device));
        HIPCHK(rt, hipMemcpyAsync(recv_host, rt->comm->recv.p, total * 8, hipMemcpyDeviceToHost, rt->stream));
        HIPCHK(rt, stream_wait(rt->stream));
    }
    return MXE_OK;
}
MXE_CATCH_ALL

extern "C" int mxe_comm_allreduce(mxe_ctx* ctx, double* inout_host, int n, int op)
try {
    if (!ctx || !inout_host || n < 1 || n > 64 || (op != 0 && op != 1)) return MXE_ERR_ARG;
    if (!ctx->comm) return MXE_ERR_STATE;
    mxe_comm_state* cm = ctx->comm;
    if ((cm->n_ranks == 1 && !cm->loopback) || !cm->local.empty()) return MXE_OK;        // one process: nothing to agree on
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, cm->small.ensure(64));
    HIPCHK(ctx, hipMemcpyAsync(cm->small.p, inout_host, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(ctx, g_rccl.AllReduce(cm->small.p, cm->small.p, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, cm->comm, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(inout_host, cm->small.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    return MXE_OK;
}
MXE_CATCH_ALL

// element e -> rank e mod n_ranks (an element's alpha scan stays on one device); pure host arithmetic
extern "C" int mxe_shard_plan(int n_elem, int n_ranks, int32_t* rank_of_elem, int32_t* local_index, int32_t* n_local)
{
    if (n_elem < 0 || n_ranks < 1 || !rank_of_elem || !local_index || !n_local) return MXE_ERR_ARG;
    for (int r = 0; r < n_ranks; ++r) n_local[r] = 0;
    for (int e = 0; e < n_elem; ++e) {
        const int r = e % n_ranks;
        rank_of_elem[e] = r;
        local_index[e] = n_local[r]++;
    }
    return MXE_OK;
}

// ---- output map A = B H ----------------------------------------------------
namespace mxe {
// one workgroup per problem; thread j computes A_j = sum_k B[j][k] H[k]
// with Bt (= B^T, [k][j]) so that the loads are coalesced along j.
__global__ __launch_bounds__(256)
void output_map_kernel(const double* __restrict__ Bt, const double* __restrict__ H,
                       double* __restrict__ A, int nw)
{
    extern __shared__ double hs[];
    const double* Hp = H + (size_t)blockIdx.x * nw;
    for (int k = threadIdx.x; k < nw; k += blockDim.x) hs[k] = Hp[k];
    __syncthreads();
    for (int j = threadIdx.x; j < nw; j += blockDim.x) {
        double a0 = 0.0, a1 = 0.0;
        int k = 0;
        for (; k + 1 < nw; k += 2) {
            a0 = fma(Bt[(size_t)k * nw + j], hs[k], a0);
            a1 = fma(Bt[(size_t)(k + 1) * nw + j], hs[k + 1], a1);
        }
        if (k < nw) a0 = fma(Bt[(size_t)k * nw + j], hs[k], a0);
        A[(size_t)blockIdx.x * nw + j] = a0 + a1;
    }
}
} // namespace mxe

extern "C" int mxe_apply_output_map(mxe_ctx* ctx, const double* B, double* out_A)
try {
    if (!ctx || !B || !out_A) return MXE_ERR_ARG;
    if (!ctx->launched) return MXE_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int nw = ctx->n_omega;
    const size_t P = (size_t)ctx->n_chain * ctx->n_alpha;
    std::vector<double> Bt((size_t)nw * nw);
    for (int j = 0; j < nw; ++j) for (int k = 0; k < nw; ++k) Bt[(size_t)k * nw + j] = B[(size_t)j * nw + k];
    HIPCHK(ctx, ctx->dB.ensure(Bt.size()));
    HIPCHK(ctx, ctx->dA.ensure(P * nw));
    HIPCHK(ctx, hipMemcpyAsync(ctx->dB.p, Bt.data(), Bt.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mxe::output_map_kernel, dim3((unsigned)P), dim3(256), (size_t)nw * 8, ctx->stream,
                       ctx->dB.p, ctx->dout_H.p, ctx->dA.p, nw);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out_A, ctx->dA.p, P * nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, stream_wait(ctx->stream));
    return MXE_OK;
}
MXE_CATCH_ALL

// ---- kernel matrix staging on the device: fill, preblur product, truncated SVD -------------
#include "mxe_svd.hip.h"


extern "C" int mxe_kernel_svd(int device, int n_tau, int n_omega, const double* tau, const double* omega,
                              const double* delta, double beta, int n_b, const double* preblur_b,
                              double threshold, int ns_max, double* out_K, double* out_U, double* out_S,
                              double* out_V, int32_t* out_ns, int32_t* out_info, float* out_ms)
try {
    if (n_tau < 1 || n_omega < 1 || n_b < 1 || !tau || !omega || !delta || !preblur_b || !out_U || !out_S ||
        !out_V || !out_ns || ns_max < 1 || ns_max > mxe::SVD_RCAP || !(threshold >= 0.0)) return MXE_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MXE_ERR_NODEVICE;
    if (device < 0 || device >= ndev) return MXE_ERR_ARG;
    const size_t lds = ((size_t)n_tau + 16 + 8 + mxe::SVD_RCAP + mxe::SVD_RCAP / 2) * sizeof(double);
    if (lds > 60 * 1024) return MXE_ERR_LIMIT;
    SVDCHK(hipSetDevice(device));
    SvdScratch sc;
    SVDCHK(hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking));
    SVDCHK(hipEventCreate(&sc.e0));
    SVDCHK(hipEventCreate(&sc.e1));
    const size_t m = n_tau, n = n_omega, R = mxe::SVD_RCAP;
    double *dtau, *dom, *ddel, *dKt0, *dr1, *ddc, *dB;
    mxe::SvdParams sp;
    sp.m = n_tau; sp.n = n_omega; sp.ns_max = ns_max; sp.threshold = threshold;
    SVDCHK(sc.alloc(&dtau, m)); SVDCHK(sc.alloc(&dom, n)); SVDCHK(sc.alloc(&ddel, n));
    SVDCHK(sc.alloc(&dKt0, n * m)); SVDCHK(sc.alloc(&dr1, n)); SVDCHK(sc.alloc(&ddc, n)); SVDCHK(sc.alloc(&dB, n * n));
    SVDCHK(sc.alloc(&sp.A, (size_t)n_b * n * m)); SVDCHK(sc.alloc(&sp.Vh, (size_t)n_b * R * m));
    SVDCHK(sc.alloc(&sp.Rm, (size_t)n_b * R * n)); SVDCHK(sc.alloc(&sp.Jt, (size_t)n_b * R * R));
    SVDCHK(sc.alloc(&sp.Qc, (size_t)n_b * R * m)); SVDCHK(sc.alloc(&sp.cn2, (size_t)n_b * n));
    SVDCHK(sc.alloc(&sp.perm, (size_t)n_b * n));
    SVDCHK(sc.alloc(&sp.out_U, (size_t)n_b * m * ns_max)); SVDCHK(sc.alloc(&sp.out_S, (size_t)n_b * ns_max));
    SVDCHK(sc.alloc(&sp.out_V, (size_t)n_b * n * ns_max)); SVDCHK(sc.alloc(&sp.out_info, (size_t)n_b * 4));
    SVDCHK(hipMemcpyAsync(dtau, tau, m * 8, hipMemcpyHostToDevice, sc.stream));
    SVDCHK(hipMemcpyAsync(dom, omega, n * 8, hipMemcpyHostToDevice, sc.stream));
    SVDCHK(hipMemcpyAsync(ddel, delta, n * 8, hipMemcpyHostToDevice, sc.stream));
    SVDCHK(hipEventRecord(sc.e0, sc.stream));
    const int nel = n_tau * n_omega;
    hipLaunchKernelGGL(mxe::tau_kernel_fill, dim3((nel + 255) / 256), dim3(256), 0, sc.stream, dtau, dom, beta, n_tau, n_omega, dKt0);
    SVDCHK(hipGetLastError());
    for (int ib = 0; ib < n_b; ++ib) {
        double* Ai = sp.A + (size_t)ib * n * m;
        const double b = preblur_b[ib];
        if (b > 0.0) {
            hipLaunchKernelGGL(mxe::preblur_rows, dim3(n_omega), dim3(256), 0, sc.stream, dom, ddel, b, n_omega, dr1);
            hipLaunchKernelGGL(mxe::preblur_cols, dim3(n_omega), dim3(256), 0, sc.stream, dom, ddel, b, n_omega, dr1, ddc);
            hipLaunchKernelGGL(mxe::preblur_matrix, dim3((n_omega * n_omega + 255) / 256), dim3(256), 0, sc.stream, dom, b, n_omega, dr1, ddc, dB);
            hipLaunchKernelGGL(mxe::preblur_product, dim3(n_omega), dim3(256), n * 8, sc.stream, dKt0, ddel, dB, n_tau, n_omega, Ai);
            SVDCHK(hipGetLastError());
        } else {
            SVDCHK(hipMemcpyAsync(Ai, dKt0, n * m * 8, hipMemcpyDeviceToDevice, sc.stream));
        }
    }
    std::vector<double> hKt;
    if (out_K) {
        hKt.resize((size_t)n_b * n * m);
        SVDCHK(hipMemcpyAsync(hKt.data(), sp.A, hKt.size() * 8, hipMemcpyDeviceToHost, sc.stream));
    }
    SVDCHK(hipFuncSetAttribute((const void*)mxe::svd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mxe::svd_kernel, dim3(n_b), dim3(mxe::SVD_T), lds, sc.stream, sp);
    SVDCHK(hipGetLastError());
    SVDCHK(hipEventRecord(sc.e1, sc.stream));
    std::vector<int> hinfo((size_t)n_b * 4);
    SVDCHK(hipMemcpyAsync(out_U, sp.out_U, (size_t)n_b * m * ns_max * 8, hipMemcpyDeviceToHost, sc.stream));
    SVDCHK(hipMemcpyAsync(out_S, sp.out_S, (size_t)n_b * ns_max * 8, hipMemcpyDeviceToHost, sc.stream));
    SVDCHK(hipMemcpyAsync(out_V, sp.out_V, (size_t)n_b * n * ns_max * 8, hipMemcpyDeviceToHost, sc.stream));
    SVDCHK(hipMemcpyAsync(hinfo.data(), sp.out_info, hinfo.size() * 4, hipMemcpyDeviceToHost, sc.stream));
    SVDCHK(hipStreamSynchronize(sc.stream));
    if (out_ms) SVDCHK(hipEventElapsedTime(out_ms, sc.e0, sc.e1));
    int rc = MXE_OK;
    for (int ib = 0; ib < n_b; ++ib) {
        out_ns[ib] = hinfo[(size_t)ib * 4];
        if (out_info) { out_info[ib * 3] = hinfo[(size_t)ib * 4 + 1]; out_info[ib * 3 + 1] = hinfo[(size_t)ib * 4 + 2]; out_info[ib * 3 + 2] = hinfo[(size_t)ib * 4 + 3]; }
        if (hinfo[(size_t)ib * 4 + 3] == 2) rc = MXE_ERR_LIMIT;
        else if (hinfo[(size_t)ib * 4 + 3] == 1 && rc == MXE_OK) rc = MXE_ERR_NUMERIC;
    }
    if (out_K)
        for (int ib = 0; ib < n_b; ++ib)
            for (size_t j = 0; j < n; ++j)
                for (size_t i = 0; i < m; ++i)
                    out_K[((size_t)ib * m + i) * n + j] = hKt[((size_t)ib * n + j) * m + i];
    return rc;
}
MXE_CATCH_ALL
