#!/usr/bin/env python3
"""bench.py -- alpha-solves/s of the MI355X alpha-scan solver (BASELINE.json's metric).

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM: one launch of the chain kernel, the device line fit that picks the analyzer's alpha per scan
(one tiny kernel) and, with more than one GPU, the ONE gather of the compact result packs to rank 0.

  N = 1   BASELINE cfg4 on ONE GPU: 16 x 16 matrix elements x 100 alpha = 25 600 alpha-solves per
          step (n_tau = 200, n_omega = 500, fp64) -- the batch the north-star target is quoted on.
  N > 1   --scaling strong (default): the SAME 256-element batch sharded over the ranks, element e on
          rank e mod N (mxe_shard_plan), timed until every rank's results are on rank 0 -- BASELINE cfg4
          as written.  --scaling weak: one full batch per rank (different noise per rank).
          The gather is RCCL send / recv issued by libmaxent_hip.so itself (mxe_gather); no framework is
          imported: ranks find each other through RANK / WORLD_SIZE / LOCAL_RANK and a file in /tmp that
          carries the ncclUniqueId.  --gather compact (default): chi2, S, Q of every alpha + the H row and
          index of the line-fit alpha per scan (1.6 MB for the whole batch); --gather full: all H as well.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1 under ``python -m
torch.distributed.run``, which only serves as the process launcher); rank 0 prints ONE JSON line.
"""

import os
# (streams of one process share this many hardware queues; the default of four is what the four contexts in flight need -- with a
#  communicator per context RCCL's own streams come on top and the steps in flight were SLOWER than one at a time, 1.77 against 0.85 ms)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from maxent_amd import device, synthetic, hostprep   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
L2_PEAK_GBS = 34500.0          # ... aggregate L2
N_EVAL_NOMINAL = {'normal': 160, 'plusminus': 84}   # the reference's passes per alpha-solve
# exit codes behind the printed line (0: everything checked out)
EXIT_GATHER_MISMATCH = 3       # --gpus N: a rank's gathered chi2 differs from rank 0's one-GPU solve of the whole batch
EXIT_IN_FLIGHT_CHECK = 4       # a batch in flight came back with an alpha not converged / left to finish / over the audit gate
EXIT_WATCHDOG = 5              # --in-flight-comm: the region did not come back within the time limit
EXIT_IN_FLIGHT_FAILED = 6      # --in-flight-comm: the region raised
AUDIT_GATE = 1e-6

# Counter values per launch of the default workload come from profiles/<tag>_pmc_summary.csv (rocprofv3 --pmc passes on
# this very command, condensed by tools/summarize_pmc.py), whose first line records the source hash of the library
# the counters were taken on.  They count events, not time, and do not depend on the clock -- but they belong to ONE
# build: load_pmc() only accepts a summary whose hash equals mxe_source_hash() of the library being benched.
PMC_NAMES = dict(valu_active_quadcycles='SQ_ACTIVE_INST_VALU',      # (counts quad-cycles: MI355X_MICROARCH.md, cycle constants)
                 mfma_busy_cycles='SQ_VALU_MFMA_BUSY_CYCLES',       # (cycles)
                 coexec_cycles='SQ_VALU_MFMA_COEXEC_CYCLES',        # both at once, counted once
                 any_active_quadcycles='SQ_ACTIVE_INST_ANY', wave_quadcycles='SQ_WAVE_CYCLES',
                 wait_inst_quadcycles='SQ_WAIT_INST_ANY', wait_any_quadcycles='SQ_WAIT_ANY',
                 gui_active_cycles_all_xcd='GRBM_GUI_ACTIVE',       # sum over the 8 XCDs
                 fetch_kb='FETCH_SIZE', write_kb='WRITE_SIZE',      # (FETCH_SIZE x 2 on gfx950: MI355X_MICROARCH.md, section HBM)
                 l2_hit='TCC_HIT_sum', l2_miss='TCC_MISS_sum')


def load_pmc(lib_hash, which='one_launch'):
    """(counters, None) from the newest profiles/*_pmc_summary.csv recorded on this build, or (None, reason)"""
    import glob
    seen = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_summary.csv')), reverse=True):
        # (``*_in_flight_pmc_summary.csv``: the counters of the cut for several batches in flight, launched alone;
        #  ``*_lv_pmc_summary.csv``: chain_kernel_lv; the rest: the launch of one batch at a time)
        base = os.path.basename(path)
        kind = 'in_flight' if base.endswith('_in_flight_pmc_summary.csv') else 'lv' if base.endswith('_lv_pmc_summary.csv') else 'one_launch'
        if kind != which:
            continue
        with open(path) as f:
            lines = f.read().splitlines()
        if not lines or not lines[0].startswith('#source_hash,'):
            continue                          # (summaries of earlier rounds: no hash, never applied)
        h = lines[0].split(',', 1)[1].strip()
        seen.append('%s: %s' % (os.path.basename(path), h))
        if h != lib_hash:
            continue
        vals = {}
        for ln in lines[2:]:
            parts = ln.split(',')
            if len(parts) >= 3:
                vals[parts[0]] = float(parts[2])
        missing = [c for c in PMC_NAMES.values() if c not in vals]
        if missing:
            return None, '%s lacks the counters %s' % (os.path.relpath(path, ROOT), ', '.join(missing))
        pmc = {k: vals[c] for k, c in PMC_NAMES.items()}
        pmc['source'] = os.path.relpath(path, ROOT)
        return pmc, None
    return None, ('no counter profile under profiles/ was recorded on this build (library source hash %s; profiles: %s): '
                  're-run tools/round_profile.sh' % (lib_hash, '; '.join(seen) if seen else 'none with a hash'))


N_SIMD = 256 * 4
CLOCK_PEAK_GHZ = 2.4                      # MI355X_MICROARCH.md: max clock


def build_batch(n_orb, n_tau, n_omega, n_alpha, rank):
    """Synthetic cfg3/cfg4 batch."""
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_omega, noise_seed=2025 + rank)
    t0 = time.perf_counter()
    K.reduce_singular_space(1e-14)
    t_svd = time.perf_counter() - t0
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    elems = [(i, j) for i in range(n_orb) for j in range(n_orb)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for (i, j) in elems]
    v0n = hostprep.initial_v(K.V, D, omega.delta, device.ENTROPY_NORMAL)
    v0p = hostprep.initial_v(K.V, D, omega.delta, device.ENTROPY_PLUSMINUS)
    v0 = np.stack([v0n if k == device.ENTROPY_NORMAL else v0p for k in kinds])
    return dict(tau=tau, omega=omega, K=K, Gmat=Gmat, D=D, err=err, alphas=alphas, elems=elems, kinds=kinds,
                v0=v0, t_svd=t_svd)


def stage(batch, dev, which=None):
    """DeviceContext with the elements ``which`` (default: all) of the batch staged"""
    K = batch['K']
    which = list(range(len(batch['elems']))) if which is None else list(which)
    ctx = device.DeviceContext(K.U, K.S, K.V, device=dev)
    ds = ctx.add_dataset(batch['err'])
    ctx.set_elements([ds] * len(which), [batch['Gmat'][batch['elems'][e]] for e in which],
                     np.tile(batch['D'], (len(which), 1)), [batch['kinds'][e] for e in which])
    return ctx


# ---------------------------------------------------------------------------------------------------
#  CPU side-by-side: the oracle port of the reference's algorithm on the host cores
# ---------------------------------------------------------------------------------------------------
def cpu_baseline(batch, out_gpu, n_chains=2):
    """oracle/ref_numpy.py (step-faithful numpy port of LevenbergMinimizer + MaxEntCostFunction) timed on
    one core on a bounded sample: the first diagonal and the first off-diagonal element, all alphas"""
    from oracle import ref_numpy as R
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    K = batch['K']
    n_alpha, n_tau = len(batch['alphas']), len(batch['tau'])
    mesh = batch['alphas'] / n_tau
    t_total, solves, err_ref = 0.0, 0, 0.0
    for c in [0, 1][:n_chains]:
        i, j = batch['elems'][c]
        ent = 'normal' if batch['kinds'][c] == device.ENTROPY_NORMAL else 'plusminus'
        p = R.Problem(np.array(K.K), K.U, K.S, K.V, batch['Gmat'][i, j], batch['err'], batch['D'], entropy=ent)
        timing = []
        ref = R.alpha_loop(p, batch['omega'].delta, mesh, timing=timing)
        t_total += timing[0]
        solves += n_alpha
        H = out_gpu['H'][c]
        err_ref = max(err_ref, float(np.max(np.linalg.norm(H - ref['H'], axis=1) / np.linalg.norm(ref['H'], axis=1))))
    return dict(value=solves / t_total, unit='alpha-solves/s', cores=1, kind='port',
                sample='oracle/ref_numpy.py (step-faithful numpy port of LevenbergMinimizer + MaxEntCostFunction) '
                       'on elements (0,0) normal and (0,1) plusminus x %d alpha, %.1f s' % (n_alpha, t_total),
                gpu_vs_port_max_rel_l2=err_ref)


def _pool_worker(c):
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import ref_numpy as R
    batch = _POOL_BATCH
    K = batch['K']
    i, j = batch['elems'][c]
    ent = 'normal' if batch['kinds'][c] == device.ENTROPY_NORMAL else 'plusminus'
    p = R.Problem(np.array(K.K), K.U, K.S, K.V, batch['Gmat'][i, j], batch['err'], batch['D'], entropy=ent)
    timing = []
    R.alpha_loop(p, batch['omega'].delta, batch['alphas'] / len(batch['tau']), timing=timing)
    return timing[0]


_POOL_BATCH = None


def cpu_pool_baseline(batch):
    """the same port on all host cores, one process per core, one matrix element (100 alpha) per process.
    Runs BEFORE this process touches the GPU (the workers are forked)."""
    global _POOL_BATCH
    import multiprocessing as mp
    n = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 16,
                   len(batch['elems'])))
    _POOL_BATCH = batch
    try:
        ctx = mp.get_context('fork')
        t0 = time.perf_counter()
        with ctx.Pool(processes=n) as pool:
            pool.map(_pool_worker, list(range(n)))
        wall = time.perf_counter() - t0
    except Exception as exc:              # a baseline must not break the bench
        return dict(error=repr(exc))
    finally:
        _POOL_BATCH = None
    return dict(value=n * len(batch['alphas']) / wall, cores=n,
                sample='%d matrix elements x %d alpha, one process each, %.1f s wall' % (n, len(batch['alphas']), wall))


# ---------------------------------------------------------------------------------------------------
#  parity and audit blocks
# ---------------------------------------------------------------------------------------------------
def parity_per_alpha():
    """BASELINE cfg2 from the committed reference fixture (tests/golden/cfg2_normal.npz: the reference's own
    run and its optimum polished in extended precision): relative L2 of H per stored alpha -- the GPU
    against the reference, the reference against the truth, the GPU against the truth.  The first two
    belong together: what separates the GPU from the reference is the reference's own stopping slack."""
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'cfg2_normal.npz'))
    ctx = device.DeviceContext(z['U'], z['S'], z['V'])
    ds = ctx.add_dataset(z['err'])
    ctx.set_elements([ds], [z['G']], z['D'][np.newaxis, :], [device.ENTROPY_NORMAL])
    v0 = hostprep.initial_v(z['V'], z['D'], z['delta'], device.ENTROPY_NORMAL)
    out = ctx.solve_chains([0], z['alpha'], v0[np.newaxis, :])
    ctx.close()
    rows = z['rows']
    H = out['H'][0][rows]

    def rel(a, b):
        return [float(x) for x in np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)]
    out = dict(config='cfg2: n_tau=200 n_omega=500, 100 alpha, normal entropy; fixture rows (alpha index)',
               alpha_index=[int(r) for r in rows], alpha_scaled=[float(a) for a in z['alpha'][rows]],
               gpu_vs_ref=rel(H, z['H_ref']), ref_vs_truth=rel(z['H_ref'], z['H_truth']),
               gpu_vs_truth=rel(H, z['H_truth']), gate_gpu_vs_truth=1e-6)
    tight = os.path.join(ROOT, 'tests', 'golden', 'tight_ref.npz')
    if os.path.exists(tight):
        # the reference itself under MaxDerivativeConvergenceMethod(1e-7) (tests/golden/make_golden.py: tight_case) and
        # the reference's own binary64 Newton correction at its default result, its tight result and the truth
        t = np.load(tight)
        out.update(tight_ref_vs_truth=rel(t['cfg2_H_tight_ref'], t['cfg2_H_truth']), gpu_vs_tight_ref=rel(H, t['cfg2_H_tight_ref']),
                   tight_ref_iterations=int(t['cfg2_n_iter_tight_ref'].sum()), tight_ref_converged=int(t['cfg2_converged_tight_ref'].sum()),
                   reference_newton_correction=dict(at_H_ref=[float(x) for x in t['cfg2_ref_newton_corr'][0]],
                                                    at_H_tight_ref=[float(x) for x in t['cfg2_ref_newton_corr'][1]],
                                                    at_H_truth=[float(x) for x in t['cfg2_ref_newton_corr'][2]]))
    return out


def audit_block(ctx):
    """mxe_audit over every problem of the last launch: the exact Newton correction at the returned v
    (binary64, all n_s directions), as ||w * V delta|| / ||H|| -- to first order the distance of the returned
    H from the minimiser"""
    c = ctx.audit()['corr'].ravel()
    return dict(problems=int(c.size), corr_max=float(np.nanmax(c)), corr_p99=float(np.nanpercentile(c, 99)),
                corr_median=float(np.nanmedian(c)), above_1e_6=int(np.sum(~(c <= 1e-6))),
                definition='exact Newton correction ||w * V delta||_2 / ||H||_2 at the returned v, every problem')


def end_to_end_block(batch, n_orb, n_alpha):
    """what a caller of the reference's API sees: ElementwiseMaxEnt on the same input -- H2D, one launch,
    D2H of what the result object needs, records, analyzers -- next to the device-resident figure"""
    import maxent_amd as mx

    from maxent_amd.batch_solver import BatchSolver

    def make(k=0):
        ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
        ew.set_verbosity(mx.VerbosityFlags.Quiet)
        ew.set_G_tau_data(batch['tau'], batch['Gmat'] * (1.0 + 1e-7 * k))        # (k: other data on the same grids)
        ew.omega = batch['omega']
        ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=n_alpha)
        ew.set_error(synthetic.SIGMA)
        return ew

    def fresh(n, k0):
        out = []
        for k in range(n):
            t0 = time.perf_counter()
            ew = make(k0 + k)
            ew.run()
            out.append(time.perf_counter() - t0)
        return ew, out
    pool_size, BatchSolver.POOL_SIZE = BatchSolver.POOL_SIZE, 0
    try:
        ew, cold_own = fresh(2, 1)                  # every object creates (and destroys) device contexts of its own
    finally:
        BatchSolver.POOL_SIZE = pool_size
    del ew
    ew, cold = fresh(3, 3)                          # the contexts of an earlier object with the same decomposition are taken over
    warm, res = [], None
    for _ in range(3):
        ew.maxent_result = res = None     # (a result that is still held claims its H: it would be fetched first)
        t0 = time.perf_counter()
        res = ew.run()
        warm.append(time.perf_counter() - t0)
    # (the rows of the result's default analyzer -- what res.A_out shows -- come with the solve; those of the other two
    #  analyzers when somebody looks: one copy of 1 MB per analyzer + the division by delta, timed here for every element)
    t0 = time.perf_counter()
    for name in ('Chi2CurvatureAnalyzer', 'EntropyAnalyzer'):
        for (i, j) in ((0, 0), (0, 1)):                    # (one element of each worker's batch: the batch's rows are formed together)
            res.analyzer_results[i][j][name]['A_out']
    t_other = time.perf_counter() - t0
    t0 = time.perf_counter()
    nbytes = np.asarray(res.H).nbytes
    t_H = time.perf_counter() - t0
    # (the first fetch of a process page-locks its destination, 22 ms for 102 MB; the block goes back to the library's pool
    #  with the result and the next result's fetch is one DMA into it)
    for _ in range(2):                       # (the first of the two page-locks the block A goes into)
        ew.maxent_result = res = None
        res = ew.run()
        t0 = time.perf_counter()
        np.asarray(res.H)
        t_H2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        np.asarray(res.A)
        t_A2 = time.perf_counter() - t0
    many = many_objects_block(make, res, n_orb, n_alpha)
    del res
    # BASELINE config 2 through the API: TauMaxEnt.run() of ONE scan of n_alpha alphas (the reference's most common call)
    tau1, omega1, _, G1 = synthetic.single_G(200, 500)

    def single(k=0):
        tm = mx.TauMaxEnt()
        tm.set_verbosity(mx.VerbosityFlags.Quiet)
        tm.set_G_tau_data(tau1, G1 * (1.0 + 1e-7 * k))
        tm.set_error(synthetic.SIGMA)
        tm.omega = omega1
        tm.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=n_alpha)
        return tm
    tm = single(); tm.run()
    one_fresh, one_warm = [], []
    for k in range(1, 5):
        t0 = time.perf_counter(); tm = single(k); r1 = tm.run(); one_fresh.append(time.perf_counter() - t0)
    for _ in range(5):
        r1 = None
        t0 = time.perf_counter(); r1 = tm.run(); one_warm.append(time.perf_counter() - t0)
    del r1, tm
    P = n_orb * n_orb * n_alpha
    return dict(api='ElementwiseMaxEnt(use_hermiticity=False).run()', problems=P,
                single_scan=dict(api='TauMaxEnt.run(), one scan of %d alpha (BASELINE config 2)' % n_alpha,
                                 same_object_ms=1e3 * min(one_warm), fresh_object_ms=1e3 * sorted(one_fresh)[1]),
                fresh_object_ms=1e3 * min(cold), fresh_object_own_contexts_ms=1e3 * min(cold_own), same_object_ms=1e3 * min(warm),
                alpha_solves_per_s_same_object=P / min(warm), alpha_solves_per_s_fresh_object=P / min(cold),
                includes='fresh object: a new ElementwiseMaxEnt on new data of the same grids -- kernel fill + SVD + staging of the data; '
                         'the device contexts (with U, S, V staged) of an earlier object with an equal decomposition are taken over '
                         '(maxent_amd.batch_solver.BatchSolver.for_kernel; fresh_object_own_contexts_ms: without that, as in round 2). '
                         'The same object again uploads nothing when the job is unchanged.  Both: one '
                         'launch (diagonal and off-diagonal elements together) + the selection kernel of the LineFit / Chi2Curvature / '
                         'Entropy analyzers, D2H of chi2 / S / Q / flags, of the three analyzers\' indices and of the rows of the default '
                         'analyzer (v, H and the rows of the other two analyzers stay on the device until looked at: '
                         'other_analyzers_rows_ms), records (built while the kernel runs), the analysis batch',
                many_objects=many,
                other_analyzers_rows_ms=1e3 * t_other,
                first_access_of_all_H_ms=1e3 * t_H, all_H_of_a_later_result_ms=1e3 * t_H2, all_A_of_it_ms=1e3 * t_A2, all_H_MB=nbytes / 1e6)


def many_objects_block(make, res_one, n_orb, n_alpha, n_jobs=4, repeats=6):
    """Jobs in flight behind the reference's API: ``n_jobs`` ElementwiseMaxEnt objects on DIFFERENT data of the
    same grids through ``maxent_amd.run_many`` -- every object prepared, staged and launched before the first is waited for -- against
    the same objects' ``run()`` one after the other.  A_out and chi2 of the two ways are compared."""
    import maxent_amd as mx
    jobs = [make(100 + k) for k in range(n_jobs)]
    seq = [ew.run() for ew in jobs]                     # (first run of every object: contexts, staging)
    seq_keep = [(np.array(r.A_out), np.array(r.chi2)) for r in seq]
    del seq
    P = n_orb * n_orb * n_alpha * n_jobs

    def clear():
        for ew in jobs:
            ew.maxent_result = None
    t_seq = []
    for _ in range(repeats):
        clear()
        t0 = time.perf_counter()
        out = [ew.run() for ew in jobs]
        t_seq.append(time.perf_counter() - t0)
        del out
    t_many, out = [], None
    for _ in range(repeats + 2):
        clear()
        out = None
        t0 = time.perf_counter()
        out = mx.run_many(jobs)
        t_many.append(time.perf_counter() - t0)
    t_many = t_many[2:]                                 # (the first passes re-cut the chains for mxe_opts.in_flight = n_jobs)
    # the jobs cut for the company they have (same_cut=False: mxe_opts.in_flight = n_jobs, fewer cold-started pieces)
    t_cut, out_cut = [], None
    for _ in range(repeats + 2):
        clear()
        out_cut = None
        t0 = time.perf_counter()
        out_cut = mx.run_many(jobs, same_cut=False)
        t_cut.append(time.perf_counter() - t0)
    t_cut = t_cut[2:]
    rel_A = lambda rs: float(max(np.max(np.abs(np.asarray(r.A_out) - a)) / np.max(np.abs(a)) for r, (a, _) in zip(rs, seq_keep)))
    cut_A = rel_A(out_cut)
    cut_chi2 = float(max(np.nanmax(np.abs(np.asarray(r.chi2) - c) / np.abs(c)) for r, (_, c) in zip(out_cut, seq_keep)))
    del out_cut
    # two sequential run() calls of the SAME object (round 5: the full batch repeats bit for bit -- the partial h of a workgroup's
    # four waves are summed pairwise, no longer by four atomic additions in whatever order they arrive)
    clear()
    rerun_A = rel_A([ew.run() for ew in jobs])
    for _ in range(2):                                  # (back to the cut of run())
        clear()
        out = mx.run_many(jobs)
    same_A = bool(all(np.array_equal(np.asarray(r.A_out), a) for r, (a, _) in zip(out, seq_keep)))
    worst_A = float(max(np.max(np.abs(np.asarray(r.A_out) - a)) / np.max(np.abs(a)) for r, (a, _) in zip(out, seq_keep)))
    worst_chi2 = float(max(np.nanmax(np.abs(np.asarray(r.chi2) - c) / np.abs(c)) for r, (_, c) in zip(out, seq_keep)))
    conv = bool(all(np.all(np.asarray(r.converged)[~np.isnan(np.asarray(r.converged))] == 1) for r in out))
    kernels = sorted(set(info['kernel'] for ew in jobs for info in ew.last_launches[-1:]))
    # new data on the same objects before every pass (what a self-consistency loop does): the staging uploads G
    t_new = []
    for rep in range(repeats):
        for k, ew in enumerate(jobs):
            ew.set_G_tau_data(ew.G_mat[0], np.asarray(ew.G_mat[1]) * (1.0 + 1e-8 * (rep + 1)))
        t0 = time.perf_counter()
        out2 = mx.run_many(jobs)
        t_new.append(time.perf_counter() - t0)
        del out2
    # eight objects (what the pool of solvers holds): nearer the steady state of launches in flight than a burst of four
    more = jobs + [make(300 + k) for k in range(8 - n_jobs)] if n_jobs < 8 else jobs
    t_eight = []
    if len(more) == 8:
        o8 = mx.run_many(more)
        del o8
        for _ in range(repeats):
            for ew in more:
                ew.maxent_result = None
            t0 = time.perf_counter()
            o8 = mx.run_many(more)
            t_eight.append(time.perf_counter() - t0)
            del o8
    return dict(api='maxent_amd.run_many([ew0 .. ew%d]) -- ElementwiseMaxEnt.run_async() on every object, then .result() in turn' % (n_jobs - 1),
                eight_objects_ms=(1e3 * min(t_eight) if t_eight else None),
                alpha_solves_per_s_eight_objects=(8 * n_orb * n_orb * n_alpha / min(t_eight) if t_eight else None),
                jobs=n_jobs, problems=P,
                run_many_ms=1e3 * min(t_many), sequential_runs_ms=1e3 * min(t_seq), run_many_new_data_ms=1e3 * min(t_new),
                run_many_cut_for_in_flight_ms=1e3 * min(t_cut), alpha_solves_per_s_cut_for_in_flight=P / min(t_cut),
                cut_for_in_flight_A_out_max_rel_diff=cut_A, cut_for_in_flight_chi2_max_rel_diff=cut_chi2,
                sequential_rerun_A_out_max_rel_diff=rerun_A,
                alpha_solves_per_s=P / min(t_many), alpha_solves_per_s_sequential=P / min(t_seq),
                alpha_solves_per_s_new_data=P / min(t_new),
                A_out_bitwise_equal_to_sequential_runs=same_A, A_out_max_rel_diff=worst_A, chi2_max_rel_diff=worst_chi2,
                all_converged=conv, kernels=kernels,
                note='run_many keeps the cut of run(): every field bit for bit what the sequential calls return (A_out_bitwise_equal_to_'
                     'sequential_runs; A_out_max_rel_diff and chi2_max_rel_diff are then 0).  cut_for_in_flight: run_many(jobs, '
                     'same_cut=False), mxe_opts.in_flight = %d (fewer cold-started pieces): other iterates, the same minimisers within '
                     'the stopping tolerance; the analyzers pick the same alphas unless two candidates tie at that level' % n_jobs)


def scaling_projection(batch, opts, k_ms_full, n_alpha, n_launch=30):
    """What a strong-scaling run of the ONE batch can give, measured on this one GPU: for N = 2, 4, 8 the kernel time of
    EVERY rank's shard (element e on rank e mod N, mxe_shard_plan) -- a job ends with its slowest rank -- against the
    time of the whole batch.  No gather in these numbers (the compact pack is 1.6 MB in total)."""
    n_elem = len(batch['elems'])
    out = dict(whole_batch_kernel_ms=k_ms_full, method='one GPU solves each rank\'s shard in turn (HIP events over %d launches '
               'back to back); speedup = whole-batch kernel time / slowest shard' % n_launch)
    for N in (2, 4, 8):
        times, kern, depths = [], None, []
        for r in range(N):
            which = [e for e in range(n_elem) if e % N == r]
            c = stage(batch, 0, which)
            c.upload_chains(np.arange(len(which), dtype=np.int32), batch['alphas'], batch['v0'][which], opts)
            for _ in range(3):
                c.launch()
            c.sync()
            c.timing_mark()
            for _ in range(n_launch):
                c.launch()
            times.append(c.ms_since_mark() / n_launch)
            kern = c.last_launch_info()['kernel']
            depths.append(c.launch_depth())
            c.close()
        slow = int(np.argmax(times))
        out['N=%d' % N] = dict(shard_kernel_ms=[round(t, 4) for t in times], slowest_ms=max(times), fastest_ms=min(times),
                               slowest_rank=slow, kernel=kern, alpha_solves_per_rank=len(which) * n_alpha,
                               speedup_before_gather=k_ms_full / max(times),
                               rounds_deepest_workgroup=[d['max_rounds'][0] for d in depths],
                               us_per_round_slowest_rank=1e3 * times[slow] / max(1, depths[slow]['max_rounds'][0]))
        # the slowest rank's shard with FOUR steps in flight on its GPU (four contexts, mxe_opts.in_flight = 4: what bench.py does on
        # one rank; with --gpus N every rank keeps one context so far -- the gather would need a communicator per context): a
        # shard alone does not fill the GPU and is bound by the depth of its chains, several of them side by side are not
        which = [e for e in range(n_elem) if e % N == slow]
        # (N = 8 with EIGHT as well: at 3 200 alpha-solves per rank four steps in flight fill half the GPU -- VERDICT r04 item 5)
        for n_fl, key in ((4, 'four_in_flight'),) + (((8, 'eight_in_flight'),) if N == 8 else ()):
            fl_opts = device.default_opts(in_flight=n_fl)
            lanes = []
            for _ in range(n_fl):
                c = stage(batch, 0, which)
                c.upload_chains(np.arange(len(which), dtype=np.int32), batch['alphas'], batch['v0'][which], fl_opts)
                lanes.append(c)
            for k in range(4 * n_fl):
                lanes[k % n_fl].launch(); lanes[k % n_fl].select_launch(0)
            for c in lanes:
                c.sync()
            t0 = time.perf_counter()
            for k in range(n_fl * n_launch):
                lanes[k % n_fl].launch(); lanes[k % n_fl].select_launch(0)
            for c in lanes:
                c.sync()
            fl_ms = 1e3 * (time.perf_counter() - t0) / (n_fl * n_launch)
            lanes[0].launch()
            fl_left = lanes[0].finish()
            fl_res = lanes[0].fetch(want_v=False, want_H=False)
            fl_aud = float(np.nanmax(lanes[0].audit()['corr']))
            fl_info = lanes[0].last_launch_info()
            for c in lanes:
                c.close()
            out['N=%d' % N][key] = dict(
                ms_per_step_slowest_rank=fl_ms, speedup_before_gather=k_ms_full / fl_ms, kernel=fl_info['kernel'],
                workgroups=fl_info['n_workgroups'], converged=int(fl_res['converged'].sum()), left_to_finish=int(fl_left), audit_max=fl_aud,
                note='step = chain kernel + device line fit of the shard, %d contexts in turn, timed by the host clock; speed-up against '
                     'the one-batch kernel time of the whole batch, like the line above it' % n_fl)
    # the bound, from this run: a shard that does not fill the GPU is as long as its deepest workgroup (rounds counted by the
    # kernel, mxe_launch_depth) times what a round takes there (the shard's kernel time / that depth)
    deep = [out['N=%d' % N]['rounds_deepest_workgroup'][out['N=%d' % N]['slowest_rank']] for N in (2, 4, 8)]
    usr = [out['N=%d' % N]['us_per_round_slowest_rank'] for N in (2, 4, 8)]
    out['bound'] = dict(
        rounds_deepest_chain=[min(deep), max(deep)], us_per_round_one_workgroup_per_cu=float(np.mean(usr[1:])),
        floor_ms=[1e-3 * min(deep) * float(np.mean(usr[1:])), 1e-3 * max(deep) * float(np.mean(usr[1:]))],
        speedup_ceiling_at_this_round_structure=k_ms_full / (1e-3 * min(deep) * float(np.mean(usr[1:]))),
        measured='rounds: mxe_launch_depth of the slowest rank\'s shard; us per round: that shard\'s kernel time / its rounds (N = 4, 8: '
                 'one workgroup per CU); floor = rounds x us; the why is in DESIGN.md section 7')
    return out


def underfilled_block(n_launch=40):
    """The BASELINE configurations that do not fill one GPU, in the same run: cfg2 (one scan of 100 alpha) and cfg3 (4 x 4: 16
    scans), binary64 and binary32 (mxe_opts.precision: the LDS-resident kernel), kernel time by HIP events over launches back to
    back, depth in rounds, every problem audited; and a binary32 request on the cfg4 batch -- as the library runs it (promoted to the
    binary64 kernel at two workgroups per CU: cfg4_f32) and held in chain_kernel_lv (cfg4_f32_lv)."""
    out = {}
    for name, n_orb in (('cfg2', 1), ('cfg3', 4), ('cfg4', 16)):
        batch = build_batch(max(n_orb, 2), 200, 500, 100, 0)
        if n_orb == 1:
            _, _, _, G1 = synthetic.single_G(200, 500)
            batch['Gmat'] = G1[None, None, :]
            batch['elems'], batch['kinds'], batch['v0'] = [(0, 0)], batch['kinds'][:1], batch['v0'][:1]
        n = len(batch['elems'])
        for tag, o in (('', {}), ('_f32', dict(precision=device.PRECISION_F32)),
                       ('_f32_lv', dict(precision=device.PRECISION_F32, wg_per_cu=1))):
            if tag == '_f32_lv' and name != 'cfg4':
                continue                                  # (cfg4 only: a binary32 request on a batch that fills the GPU is promoted to
                                                          #  the binary64 kernel; one workgroup per CU keeps it in chain_kernel_lv)
            c = stage(batch, 0)
            c.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(**o))
            for _ in range(3):
                c.launch()
            c.sync()
            c.timing_mark()
            for _ in range(n_launch):
                c.launch()
            ms = c.ms_since_mark() / n_launch
            c.launch()
            info, depth, left = c.last_launch_info(), c.launch_depth(), c.finish()
            res = c.fetch(want_v=False, want_H=False)
            aud = c.audit()['corr'].ravel()
            c.close()
            out[name + tag] = dict(kernel_ms=ms, alpha_solves=int(res['converged'].size), converged=int(res['converged'].sum()),
                                   left_to_finish=int(left), kernel=info['kernel'], workgroups=info['n_workgroups'],
                                   rounds_deepest_workgroup=depth['max_rounds'][0], evals_per_solve=float(res['n_evals'].mean()),
                                   audit_max=float(np.nanmax(aud)), audit_p99=float(np.nanpercentile(aud, 99)))
    # the same small jobs four in flight (four contexts take the launches in turn): a launch that does not fill the GPU leaves room for
    # the others -- its latency is the depth of its chains, its throughput is not
    for name, n_orb in (('cfg2', 1), ('cfg3', 4)):
        batch = build_batch(max(n_orb, 2), 200, 500, 100, 0)
        if n_orb == 1:
            _, _, _, G1 = synthetic.single_G(200, 500)
            batch['Gmat'] = G1[None, None, :]
            batch['elems'], batch['kinds'], batch['v0'] = [(0, 0)], batch['kinds'][:1], batch['v0'][:1]
        n = len(batch['elems'])
        lanes = []
        for _ in range(4):
            c = stage(batch, 0)
            c.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts())
            lanes.append(c)
        for k in range(16):
            lanes[k % 4].launch()
        for c in lanes:
            c.sync()
        t0 = time.perf_counter()
        for k in range(4 * n_launch):
            lanes[k % 4].launch()
        for c in lanes:
            c.sync()
        out[name + '_four_in_flight_ms'] = 1e3 * (time.perf_counter() - t0) / (4 * n_launch)
        for c in lanes:
            c.close()
    out['cfg2_ms'], out['cfg3_ms'] = out['cfg2']['kernel_ms'], out['cfg3']['kernel_ms']
    out['cfg2_f32_ms'], out['cfg3_f32_ms'], out['cfg4_f32_ms'] = out['cfg2_f32']['kernel_ms'], out['cfg3_f32']['kernel_ms'], out['cfg4_f32']['kernel_ms']
    out['cfg4_f32_lv_ms'] = out['cfg4_f32_lv']['kernel_ms']
    # (cfg4 itself by the method of this block -- launches from Python back to back, 5 % above the headline's step of one graph --
    #  so that the binary32 figures beside it compare like with like)
    out['cfg4_ms'] = out['cfg4']['kernel_ms']
    return out


# ---------------------------------------------------------------------------------------------------
#  ranks in separate processes
# ---------------------------------------------------------------------------------------------------
def comm_setup(ctx, rank, world, suffix=''):
    """ncclUniqueId from rank 0 to the others through a file (single node: one /tmp); ``suffix``: one communicator per context
    when a rank keeps several in flight"""
    tag = '%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'))
    path = os.path.join('/tmp', 'mxe_bench_id_%s_%d%s' % (tag, os.getppid(), suffix))
    if rank == 0:
        uid = device.comm_unique_id()
        with open(path + '.tmp', 'wb') as f:
            f.write(uid)
        os.replace(path + '.tmp', path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 300:
                raise SystemExit('bench.py: rank 0 never published the communicator id')
            time.sleep(0.01)
        with open(path, 'rb') as f:
            uid = f.read()
    # RCCL prints a version banner on stdout when it initialises: keep stdout for the one JSON line
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        ctx.comm_init(world, rank, uid)
        if world == 1:
            # --force-comm on a one-GPU box: the rank's own pack goes through ncclSend / ncclRecv to itself and the
            # reductions through ncclAllReduce, so that the RCCL calls of the multi-GPU step execute (and are timed)
            ctx.comm_set_loopback(True)
        ctx.allreduce([0.0])                   # everybody is in
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass


def in_flight_comm_region(batch, mine, local_rank, rank, world, counts, full, args, n):
    """The timed region with ``n`` steps in flight on EVERY rank (--gpus N): n contexts per rank, uploaded with mxe_opts.in_flight = n,
    each with a communicator of its own (context k of all ranks form communicator k), take the steps in turn -- launch, device line
    fit, gather to rank 0 on the context's stream.  All ranks issue the steps in the same order.  Returns (elapsed seconds, max
    over the ranks; on rank 0 also what a step in flight returned; the seconds the n communicators took to initialise).  A shard alone does not fill its GPU and is bound by the depth
    of its chains (scaling_projection: 0.39 ms at N = 8); four of them side by side are not (0.2 ms)."""
    opts_fl = device.default_opts(waves_per_chain=args.waves_per_chain, chains_per_wg=args.chains_per_wg,
                                  alpha_split=args.alpha_split, wg_per_cu=args.wg_per_cu, in_flight=n)
    lanes = []
    for k in range(n):
        c = stage(batch, local_rank, mine)
        c.upload_chains(np.arange(len(mine), dtype=np.int32), batch['alphas'], batch['v0'][mine], opts_fl)
        lanes.append(c)
    # every communicator BEFORE any launch, one after the other (comm_setup ends with an all-reduce of its own: the ranks leave
    # communicator k together before anybody starts on k + 1), timed apart from the region
    t_comm0 = time.perf_counter()
    for k, c in enumerate(lanes):
        comm_setup(c, rank, world, suffix='_lane%d' % k)
    t_comm = time.perf_counter() - t_comm0

    def step(k):
        c = lanes[k % n]
        c.launch()
        c.select_launch(0)
        c.gather(0, counts, full=full)

    def barrier():
        for c in lanes:
            c.sync()
        lanes[0].allreduce([0.0])

    for k in range(max(args.warmup, 2 * n)):
        step(k)
    barrier()
    t_settle, k = time.perf_counter(), 0
    while True:
        # (the ranks agree on the number of settling passes -- every step ends with a gather that pairs them up)
        go = lanes[0].allreduce([1.0 if time.perf_counter() - t_settle < 0.5 else 0.0], 'max')[0] > 0.5
        if not go:
            break
        for _ in range(n):
            step(k)
            k += 1
        for c in lanes:
            c.sync()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    for c in lanes:
        c.sync()
    barrier()
    elapsed = float(lanes[0].allreduce([time.perf_counter() - t0], 'max')[0])
    check = None
    if rank == 0:
        lanes[-1].launch()
        left = lanes[-1].finish()
        o = lanes[-1].fetch(want_v=False, want_H=False)
        a = lanes[-1].audit()['corr'].ravel()
        info = lanes[-1].last_launch_info()
        check = dict(kernel=info['kernel'], workgroups=info['n_workgroups'], converged=int(o['converged'].sum()),
                     alpha_solves=int(o['converged'].size), left_to_finish=int(left), evals_per_solve=float(o['n_evals'].mean()),
                     audit_max=float(np.nanmax(a)), audit_p99=float(np.nanpercentile(a, 99)))
    barrier()
    for c in lanes:
        c.comm_destroy()
        c.close()
    return elapsed, check, t_comm


PHASE = ['start-up']         # what a rank is doing, for the watchdog of a multi-rank run


def start_watchdog(rank, world, args, seconds=None):
    """The first run between two GPUs must say what happened: a rank that sits in a collective nobody else
    reaches would hang until the driver's limit with no record.  Every rank names the phase it is in (``PHASE``); when the whole run
    takes longer than MXE_BENCH_TIMEOUT seconds (900) the rank says where it was on stderr, rank 0 prints a line with value = null
    and the error, and the process leaves with code 7."""
    def stuck():
        msg = 'bench.py: rank %d of %d did not finish within the time limit; it was in: %s' % (rank, world, PHASE[0])
        print(msg, file=sys.stderr, flush=True)
        if rank == 0:
            print(json.dumps(dict(metric='alpha-solves/s', value=None, unit='alpha-solves/s', n_gpus=world, steps=args.steps,
                                  warmup=args.warmup, ms_per_step=None, higher_is_better=True, scaling=args.scaling, vs_baseline=None,
                                  dtype='f64', data='synthetic', error=msg, watchdog_fired=True)), flush=True)
        os._exit(7)
    dog = threading.Timer(float(os.environ.get('MXE_BENCH_TIMEOUT', '900')) if seconds is None else seconds, stuck)
    dog.daemon = True
    dog.start()
    return dog


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # a step is ~0.9 ms: the default 1000 keep the timed region long (0.9 s) against the tens of milliseconds by
    # which a fresh submission is sometimes picked up late on the MI355X boxes
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--scaling', choices=('strong', 'weak'), default='strong')
    ap.add_argument('--gather', choices=('compact', 'full'), default='compact')
    ap.add_argument('--n-orb', type=int, default=16)
    ap.add_argument('--n-tau', type=int, default=200)
    ap.add_argument('--n-omega', type=int, default=500)
    ap.add_argument('--n-alpha', type=int, default=100)
    ap.add_argument('--waves-per-chain', type=int, default=0)
    ap.add_argument('--chains-per-wg', type=int, default=0)
    ap.add_argument('--alpha-split', type=int, default=0)
    ap.add_argument('--wg-per-cu', type=int, default=0)
    ap.add_argument('--in-flight', type=int, default=4, choices=tuple(range(1, 9)),
                    help='one rank: behind the timed region (ONE context, one batch at a time: value / ms_per_step) a second region with n '
                         'batches in flight -- n device contexts (n streams) take the steps in turn, every batch cut into 1 / n as many '
                         'cold-started pieces (mxe_opts.in_flight) -- reported as value_in_flight / ms_per_step_in_flight; 1: skip it')
    ap.add_argument('--in-flight-comm', type=int, default=0, choices=tuple(range(0, 9)),
                    help='--gpus N: behind the region with one context per rank, a region with n contexts per rank in flight, a communicator '
                         'each (under a watchdog).  OFF by default: it has never run between two GPUs (ADVICE r04); 0 or 1: skip it')
    ap.add_argument('--cut-for-in-flight', type=int, default=0,
                    help='counter collection only (tools/round_profile.sh): the ONE context of the timed region is uploaded with '
                         'mxe_opts.in_flight = n -- the launch the batches in flight consist of, alone on the GPU; not a bench line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the audit / parity / end-to-end blocks')
    ap.add_argument('--shard-of', type=int, default=0,
                    help='one process, one GPU: solve only the rank-0 shard of an N-way split of the batch (what one '
                         'rank of --gpus N does per step, without the gather); not a bench line')
    ap.add_argument('--shard-rank', type=int, default=0, help='with --shard-of N: time the shard of this rank')
    ap.add_argument('--force-comm', action='store_true',
                    help='initialise the communicator and run the gather even with one rank (plumbing test)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d; launch with python -m torch.distributed.run '
                  '--nproc-per-node %d' % (args.gpus, world, args.gpus), file=sys.stderr)
        if args.gpus > 1:
            sys.exit(2)
    use_comm = world > 1 or args.force_comm
    if world > 1:
        start_watchdog(rank, world, args)
    default_workload = (args.n_orb, args.n_tau, args.n_omega, args.n_alpha) == (16, 200, 500, 100) and args.cut_for_in_flight <= 1

    pool_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        # host cores first, before anything initialises the GPU in this process (forked workers)
        pool_baseline = cpu_pool_baseline(build_batch(args.n_orb, args.n_tau, args.n_omega, args.n_alpha, 0))

    if device.device_count() < 1:
        raise SystemExit('bench.py needs a GPU: the solver has no CPU fallback')

    strong = args.scaling == 'strong'
    batch = build_batch(args.n_orb, args.n_tau, args.n_omega, args.n_alpha, 0 if strong else rank)
    n_elem = len(batch['elems'])
    rank_of, local_of, n_local = device.shard_plan(n_elem, world)
    mine = [e for e in range(n_elem) if rank_of[e] == rank] if strong else list(range(n_elem))
    if args.shard_of > 1 and world == 1:
        mine = [e for e in range(n_elem) if e % args.shard_of == args.shard_rank % args.shard_of]
    ctx = stage(batch, local_rank, mine)
    opts = device.default_opts(waves_per_chain=args.waves_per_chain, chains_per_wg=args.chains_per_wg,
                               alpha_split=args.alpha_split, wg_per_cu=args.wg_per_cu, in_flight=max(args.cut_for_in_flight, 0))
    ctx.upload_chains(np.arange(len(mine), dtype=np.int32), batch['alphas'], batch['v0'][mine], opts)
    P_rank = len(mine) * args.n_alpha
    P_job = n_elem * args.n_alpha * (1 if strong else world)
    full = args.gather == 'full'
    counts = None
    if use_comm:
        PHASE[0] = 'communicator set-up (ncclCommInitRank of all ranks, one all-reduce)'
        comm_setup(ctx, rank, world)
        PHASE[0] = 'after the communicator set-up'

        def per(n):
            return n * args.n_alpha * (args.n_omega if full else 0) + 3 * n * args.n_alpha + n * (args.n_omega + 1)
        counts = [per(int(n_local[r]) if strong else n_elem) for r in range(world)]
        if args.shard_of > 1 and world == 1:
            counts = [per(len(mine))]               # (--force-comm with one rank's shard of a bigger job: the pack of that shard)

    # Several batches in flight (one rank).  The persistent workgroups of a launch finish unevenly (the slowest takes 12 % longer
    # than the mean: profiles/*_phases_mc_wg2.txt) and the next launch of the SAME stream waits for the last of them; a launch of
    # another stream starts on the CUs that are free.  And a GPU that n batches share is full without most of the cold-started
    # pieces one batch alone is cut into (mxe_opts.in_flight = n: 4 instead of 15 pieces per scan for n = 4, a cold start costs
    # 4-17 evaluations).  --in-flight n: n device contexts -- each with its own stream, staging and result buffers, uploaded with
    # mxe_opts.in_flight = n -- take the steps in turn; every step is still one pass of the hot path over one batch.  ``ctx`` (the
    # library's choice for ONE batch: what rounds 1-3 timed) keeps the one-at-a-time steps, the kernel time, the roofline.
    in_flight = args.in_flight if (world == 1 and not use_comm) else 1
    lanes = [ctx]
    turn = [0]

    def one_step(only=None):
        c = only if only is not None else lanes[turn[0] % len(lanes)]
        turn[0] += 1
        c.launch()                   # enqueued back to back: no host synchronisation inside the timed region
        c.select_launch(0)
        if use_comm:
            c.gather(0, counts, full=full)            # to rank 0's device, on the ctx stream

    def barrier():
        for c in lanes:
            c.sync()
        if use_comm:
            ctx.allreduce([0.0])

    def settle():
        # with only a few warm-up passes the first submission after a synchronisation was picked up 30-50 ms late in a quarter
        # of the processes on the MI355X boxes; a quarter of a second of untimed passes (half a second with a communicator,
        # whose first collectives finish initialising in the background)
        t_settle = time.perf_counter()
        while True:
            go = time.perf_counter() - t_settle < (0.5 if use_comm else 0.25)
            if use_comm:
                # (the ranks must agree on the number of passes -- each ends with a gather that pairs the ranks up: one clock
                #  decides, not every rank its own)
                go = bool(ctx.allreduce([1.0 if go else 0.0], 'max')[0] > 0.5)
            if not go:
                break
            one_step()
            for c in lanes:
                c.sync()
        barrier()

    PHASE[0] = 'warm-up steps (launch, line fit, gather to rank 0)'
    for _ in range(args.warmup):
        one_step()
    PHASE[0] = 'barrier behind the warm-up'
    barrier()
    settle()
    # ---- THE timed region: K steps through ONE context, one batch at a time (what rounds 1-3 timed; ``value`` is this region
    # again since round 5 -- ADVICE r04: rounds compare like with like; the steps-in-flight figure is value_in_flight) ----
    barrier()
    PHASE[0] = 'the timed region: K steps (launch, line fit, gather to rank 0)'
    ctx.timing_mark()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    t_enq = time.perf_counter()
    ctx.sync()                       # on rank 0: every gather of the region has landed
    t_drained = time.perf_counter()
    PHASE[0] = 'barrier behind the timed region'
    barrier()
    elapsed = time.perf_counter() - t0
    if use_comm:
        elapsed = float(ctx.allreduce([elapsed], 'max')[0])
    PHASE[0] = 'behind the timed region (kernel time, self-check, extras)'
    host_split = dict(enqueue_ms=1e3 * (t_enq - t0), wait_ms=1e3 * (t_drained - t_enq))
    # ... and behind it (as in rounds 1-3) the dominant kernel's own duration: HIP events around launches on the library's
    # stream, back to back, one launch at a time.  (Where it is taken matters by 1-3 % on these boxes: right behind the
    # warm-up the clocks have not come up (0.84 ms), behind a region with launches in flight they are lower (0.825).)
    ctx.sync()
    ctx.timing_mark()
    for _ in range(50):
        ctx.launch()
    k_ms = ctx.ms_since_mark() / 50

    in_flight_region = None
    in_flight_check = None
    if in_flight > 1:
        # ---- a SECOND region, reported beside the line (value_in_flight): the same steps with ``in_flight`` batches in flight.
        # Its contexts -- and ONLY they: the streams of a process share four hardware queues by default, a fifth stream halves the
        # rate of the one it shares a queue with; ``ctx`` comes back behind the region
        ctx.close()
        opts_fl = device.default_opts(waves_per_chain=args.waves_per_chain, chains_per_wg=args.chains_per_wg,
                                      alpha_split=args.alpha_split, wg_per_cu=args.wg_per_cu, in_flight=in_flight)
        lanes = []
        for _ in range(in_flight):
            c = stage(batch, local_rank, mine)
            c.upload_chains(np.arange(len(mine), dtype=np.int32), batch['alphas'], batch['v0'][mine], opts_fl)
            lanes.append(c)
        turn[0] = 0
        for _ in range(max(args.warmup, 2 * in_flight)):
            one_step()
        barrier()
        settle()
        barrier()
        # (a) as a driver's clock sees K steps: from an idle GPU to an idle GPU, filling and draining included
        t0f = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        for c in lanes:
            c.sync()
        e_fl = time.perf_counter() - t0f
        # (b) with the pipeline full at both ends: one un-timed step per lane, an event behind it on every lane's stream, K steps,
        # the time to the end of every lane's last kernel (events; every lane's interval covers ~K steps of all lanes together)
        steps_b = max(in_flight, (args.steps // in_flight) * in_flight)
        turn[0] = 0
        for _ in range(in_flight):
            one_step()
        for c in lanes:
            c.timing_mark()
        for _ in range(steps_b):
            one_step()
        lane_ms = [c.ms_since_mark() for c in lanes]
        for c in lanes:
            c.sync()
        # what the batches in flight returned: every alpha converged, the exact Newton correction at the returned v of every problem
        lanes[-1].launch()
        left = lanes[-1].finish()
        o_fl = lanes[-1].fetch(want_v=False, want_H=False)
        a_fl = lanes[-1].audit()['corr'].ravel()
        in_flight_check = dict(kernel=lanes[-1].last_launch_info()['kernel'], workgroups=lanes[-1].last_launch_info()['n_workgroups'],
                               converged=int(o_fl['converged'].sum()), alpha_solves=int(o_fl['converged'].size),
                               left_to_finish=int(left), evals_per_solve=float(o_fl['n_evals'].mean()),
                               audit_max=float(np.nanmax(a_fl)), audit_p99=float(np.nanpercentile(a_fl, 99)))
        in_flight_region = dict(contexts=in_flight, steps=args.steps, ms_per_step=1e3 * e_fl / args.steps,
                                value=P_job * args.steps / e_fl,
                                ms_per_step_steady=float(np.mean(lane_ms)) / steps_b, steps_steady=steps_b,
                                lane_interval_ms=[float(x) for x in lane_ms])
        for c in lanes:
            c.close()
        lanes = []
        ctx = stage(batch, local_rank, mine)          # (the context of one batch at a time again, for what follows)
        ctx.upload_chains(np.arange(len(mine), dtype=np.int32), batch['alphas'], batch['v0'][mine], opts)
        lanes = [ctx]

    gather_checked = None
    if use_comm and rank == 0:
        # what arrived: rank 0's own block must be its own results, and (strong scaling) everybody's chi2 must
        # be what rank 0 gets when it solves the whole batch itself
        one_step()
        recv = np.empty(int(np.sum(counts)))
        ctx.gather(0, counts, full=full, recv=recv)
        own = ctx.fetch(want_v=False, want_H=False)
        lead = P_rank * args.n_omega if full else 0
        ok = np.array_equal(recv[lead:lead + P_rank], own['chi2'].ravel())
        if strong and world > 1:
            whole = stage(batch, local_rank)
            whole.upload_chains(np.arange(n_elem, dtype=np.int32), batch['alphas'], batch['v0'], opts)
            whole.launch()
            ref = whole.fetch(want_v=False, want_H=False)['chi2']
            whole.close()
            off = 0
            for r in range(world):
                nr = int(n_local[r])
                lead_r = nr * args.n_alpha * args.n_omega if full else 0
                got = recv[off + lead_r: off + lead_r + nr * args.n_alpha].reshape(nr, args.n_alpha)
                elems_r = [e for e in range(n_elem) if rank_of[e] == r]
                ok = ok and bool(np.allclose(got, ref[elems_r], rtol=1e-6, atol=0))
                off += counts[r]
        gather_checked = bool(ok)
    elif use_comm:
        one_step()
        ctx.gather(0, counts, full=full)
        ctx.sync()
    if use_comm:
        ctx.allreduce([0.0])

    # --gpus N: behind the region above (one context per rank: what every round so far would have timed) the same steps with
    # several in flight on every rank.  It is the LAST thing a rank does, under a watchdog: communicators per context have run
    # on one GPU only (--force-comm); if the region does not come back, rank 0 prints the line of the region above and
    # everybody leaves.
    n_fl_comm = args.in_flight_comm if (use_comm and args.in_flight_comm > 1) else 1
    if rank != 0:
        ctx.comm_destroy()
        ctx.close()
        if n_fl_comm > 1:
            import threading

            def bail():
                print('bench.py: rank %d: the region with steps in flight did not come back within the time limit (watchdog); '
                      'rank 0 prints the line of the region with one context per rank' % rank, file=sys.stderr)
                sys.stderr.flush()
                os._exit(EXIT_WATCHDOG)
            dog = threading.Timer(float(os.environ.get('MXE_BENCH_IN_FLIGHT_TIMEOUT', '180')), bail)
            dog.daemon = True
            dog.start()
            try:
                in_flight_comm_region(batch, mine, local_rank, rank, world, counts, full, args, n_fl_comm)
            except Exception as exc:          # (rank 0 reports the region above)
                print('bench.py: rank %d left the region with steps in flight: %r' % (rank, exc), file=sys.stderr)
                dog.cancel()
                sys.exit(EXIT_IN_FLIGHT_FAILED)
            dog.cancel()
        return

    # the rounds of the workgroups of the timed launch (mxe_launch_depth): the build of the full batch does not count them in the
    # timed region (a memset node per launch); one launch with the switch on, outside it -- the tail of the launch as a number
    os.environ['MXE_COUNT_ROUNDS'] = '1'
    try:
        ctx.launch()
        ctx.sync()
        depth_timed = ctx.launch_depth()
    finally:
        del os.environ['MXE_COUNT_ROUNDS']
    ctx.launch()
    info = ctx.last_launch_info()
    # (mxe_chains_finish -- the alphas a lock-step launch gives up on, solved again in the one-chain layout -- is not part
    #  of the timed step: on this workload it has nothing to do, which is what this records)
    n_left_to_finish = ctx.finish()
    out = ctx.fetch()
    n_conv = int(out['converged'].sum())
    value = P_job * args.steps / elapsed

    # ---- roofline of the dominant kernel (mxe::chain_kernel_mc) ----
    n_s = ctx.n_s
    b_eval = args.n_omega * n_s * 8 + 2 * args.n_omega * 8       # one evaluation pass streams V once
    kinds_mine = [batch['kinds'][e] for e in mine]
    n_diag = sum(1 for k in kinds_mine if k == device.ENTROPY_NORMAL)
    bytes_nominal = args.n_alpha * b_eval * (n_diag * N_EVAL_NOMINAL['normal'] +
                                             (len(mine) - n_diag) * N_EVAL_NOMINAL['plusminus'])
    rounds = float(out['n_evals'].sum())          # one evaluation pass = one Newton round of one chain
    # binary64 work the kernel executes per chain-round (counted from the code):
    #   row pass  du = V delta        2 n_omega_pad NP            (v_mfma_f64_4x4x4, padded columns included)
    #   fused     h = V^T H           2 n_omega_pad NP
    #   exp / entropy / sums          ~40 n_omega
    #   (Gauss-Jordan: 2 N^3, N = 32 -- binary32 since round 3, listed as fp32_solve_tflops)
    nwp = ((args.n_omega + 127) // 128) * 128
    f64_per_round = 2 * nwp * 64 + 2 * nwp * 64 + 40 * args.n_omega
    f16_per_round = 3 * 2 * nwp * 3 * 256          # three binary16 products for the three 16 x 16 tiles of W
    lib_hash = device.source_hash()
    pmc, pmc_reason = (load_pmc(lib_hash) if (default_workload and world == 1 and args.shard_of <= 1 and
                                                info['kernel'] == 'mxe::chain_kernel_mc<32, 2>')
                       else (None, 'counters are collected for the default workload on one GPU only'))
    achieved = peak = frac = traffic = counters = None
    peak = N_SIMD * CLOCK_PEAK_GHZ                              # every SIMD busy every cycle at the maximum clock
    if pmc:
        # SIMD-cycles per launch in which a vector or an FP-MFMA instruction was executing (events, clock free)
        busy = 4 * pmc['valu_active_quadcycles'] + pmc['mfma_busy_cycles'] - pmc['coexec_cycles']
        achieved = busy / (k_ms * 1e-3) / 1e9                   # G busy SIMD-cycles per second, live kernel time
        frac = achieved / peak
        traffic = 2 * pmc['fetch_kb'] * 1e3 + pmc['write_kb'] * 1e3
        kcyc = pmc['gui_active_cycles_all_xcd'] / 8
        counters = dict(source=pmc['source'], library_source_hash=lib_hash,
                        busy_frac_by_counters_alone=busy / (N_SIMD * kcyc),
                        any_instruction_active_frac=4 * pmc['any_active_quadcycles'] / (N_SIMD * kcyc),
                        wave_time_split=dict(issuing=pmc['any_active_quadcycles'] / pmc['wave_quadcycles'],
                                             waiting_for_issue=pmc['wait_inst_quadcycles'] / pmc['wave_quadcycles'],
                                             waiting_waitcnt_or_barrier=pmc['wait_any_quadcycles'] / pmc['wave_quadcycles']),
                        clock_GHz_in_kernel=kcyc / (k_ms * 1e-3) / 1e9,
                        l2_hit_rate=pmc['l2_hit'] / (pmc['l2_hit'] + pmc['l2_miss']))
    achieved_tflops = rounds * f64_per_round / (k_ms * 1e-3) / 1e12
    roofline = dict(
        bound='simd-issue',
        kernel=info['kernel'], kernel_ms=k_ms,
        achieved=achieved, peak=peak, unit='G busy SIMD-cycles/s', frac=frac,
        frac_null_reason=(None if pmc else pmc_reason),
        traffic=traffic, counters=counters,
        definition='The kernel is bound by the instruction issue of the four SIMDs of a CU: on gfx950 the binary64 and '
                   'binary32 MFMAs run at the rate of -- and instead of -- the vector instructions (profiles/'
                   'r01_f_microbench_mfma_shadow.txt), so the solve of the home waves (binary32 Gauss-Jordan: v_fma_f32 behind '
                   'ds_swizzle / v_permlane32_swap broadcasts), the exp / entropy arithmetic, the binary64 matrix products '
                   'and the operand splitting of the Gram tiles all queue for the same pipe; only the binary16 Gram MFMAs '
                   'have a pipe of their own.  achieved = (4 SQ_ACTIVE_INST_VALU + SQ_VALU_MFMA_BUSY_CYCLES - '
                   'SQ_VALU_MFMA_COEXEC_CYCLES) per launch, read from the rocprofv3 --pmc summary under profiles/ that was '
                   'recorded on THIS build of the library (%s), / the kernel time measured here with HIP events; peak = 1024 '
                   'SIMDs x 2.4 GHz.  frac is null when no summary carries the hash of the library being benched.  It is an '
                   'occupancy of the issue slots, not useful work: useful_f64_frac prices the binary64 arithmetic alone.  Two '
                   'workgroups per CU run the serial and the streaming phases of different workgroups side by side; the '
                   'streaming passes also sit at the rate one CU gets out of its L2 (8 B per lane: 32 B per cycle, 16 B per '
                   'lane: 64, profiles/r03_a_l2_stream_rate.txt).  HBM is far from binding (hbm_frac); traffic = 2 FETCH_SIZE '
                   '+ WRITE_SIZE per launch, of which 117 MB are the compulsory per-alpha results.'
                   % (pmc['source'] if pmc else 'none: ' + str(pmc_reason)),
        hbm_frac=(None if traffic is None else traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
        l2_frac=(rounds / 4.0 * (2 * nwp * 64 * 8) / (k_ms * 1e-3) / 1e9 / L2_PEAK_GBS),
        l2_definition='V^T + V (2 x %d KB) streamed once per round of a workgroup of four chains / kernel time, against '
                      'the 34.5 TB/s aggregate L2' % (nwp * 64 * 8 // 1024),
        fp64_tflops=achieved_tflops, fp64_peak_tflops=78.6, fp64_frac=achieved_tflops / 78.6,
        useful_f64_frac=achieved_tflops / 78.6,
        fp32_solve_tflops=rounds * 2 * 32 ** 3 / (k_ms * 1e-3) / 1e12,
        fp16_gram_tflops=rounds * f16_per_round / (k_ms * 1e-3) / 1e12,
        evals_per_solve=float(out['n_evals'].mean()), newton_iters_per_solve=float(out['n_iter'].mean()),
        survey_8d=dict(
            note='SURVEY 8(d) prices the path as HBM-bound with B_eval = %d B per evaluation pass; V (224 KB) is L2 '
                 'resident, so these algorithmic rates are not bounded by the HBM peak and are reported as labelled '
                 'extras, not as the roofline' % b_eval,
            algorithmic_GBs_executed_passes=rounds * b_eval / (k_ms * 1e-3) / 1e9,
            algorithmic_GBs_reference_nominal_passes=bytes_nominal / (k_ms * 1e-3) / 1e9,
            hbm_peak_GBs=HBM_PEAK_GBS))

    what = 'chi2, S, Q of every alpha + H row and index of the line-fit alpha per scan' if not full else \
        'all H, chi2, S, Q + H row and index of the line-fit alpha per scan'
    line = dict(metric='alpha-solves/s', value=value, unit='alpha-solves/s',
                n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                scaling=('strong' if strong else 'weak'), vs_baseline=None, dtype='f64', data='synthetic',
                config=dict(
                    workload=('cfg4: ElementwiseMaxEnt %dx%d G(tau) = %d alpha scans x %d alpha = %d alpha-solves, '
                              'n_tau=%d n_omega=%d n_s=%d; ' % (args.n_orb, args.n_orb, n_elem, args.n_alpha,
                                                               n_elem * args.n_alpha, args.n_tau, args.n_omega, n_s)) +
                             ('the one batch sharded over %d GPU(s), element e on rank e mod N' % world if strong
                              else 'one such batch per GPU (%d GPUs, weak scaling)' % world),
                    in_flight=1,
                    in_flight_note=('value / ms_per_step: ONE device context, one batch at a time (the region of rounds 1-3).  '
                                    'value_in_flight / ms_per_step_in_flight: a second region of the same run with %d contexts (%d streams), '
                                    'uploaded with mxe_opts.in_flight = %d, taking the steps in turn -- what ElementwiseMaxEnt jobs reach through '
                                    'maxent_amd.run_many (end_to_end.many_objects); in_flight_check: what a batch in flight returned'
                                    % (in_flight, in_flight, in_flight) if in_flight > 1 else 'one batch at a time'),
                    in_flight_check=in_flight_check,
                    step='chain kernel + device line fit' + (' + one RCCL gather (%s) to rank 0: %.2f MB per step'
                                                            % (what, float(np.sum(counts)) * 8 / 1e6) if use_comm else ''),
                    problems_per_step=P_job, problems_on_rank0=P_rank,
                    waves_per_chain=info['waves_per_chain'], workgroups=info['n_workgroups'],
                    lds_bytes=info['lds_bytes'], converged_on_rank0=n_conv, alphas_left_to_mxe_chains_finish=n_left_to_finish,
                    svd_seconds_host=batch['t_svd'], host_split=host_split,
                    gather_checked=gather_checked,
                    self_check='with --gpus N rank 0 solves the whole batch itself and compares the chi2 of every rank as gathered: '
                               'gather_checked, exit code 3 on a mismatch',
                    multi_gpu_note='N > 1 has not been run by the builders (one-GPU boxes); the RCCL calls of the gather '
                                   '(group start / send / recv / all-reduce) are executed with one rank (--force-comm, '
                                   'tests/test_gpu_multi.py) and several contexts on one device go through device copies; '
                                   'scaling_projection holds every rank\'s shard timed on this GPU, alone and with four steps in flight.  '
                                   'With --gpus N the region with one context per rank is timed first and kept as the line unless the region '
                                   'behind it -- four contexts per rank, a communicator each, under a watchdog -- comes back faster'),
                roofline=roofline)
    # ---- flat scalars (the driver's record keeps top-level keys): kernel, one batch at a time, batches in flight ----
    line['kernel_ms'] = k_ms
    line['one_at_a_time_ms'] = line['ms_per_step']
    line['one_at_a_time_value'] = line['value']
    line['evals_per_solve'] = float(out['n_evals'].mean())
    line['launch_depth'] = dict(max_rounds=int(depth_timed['max_rounds'][0]), mean_rounds=float(depth_timed['mean_rounds'][0]),
                                note='Newton rounds of the persistent workgroups of the timed launch (lock-step: one round = one evaluation of '
                                     'every busy slot of a workgroup): the launch is as long as its slowest workgroups, the mean is what a '
                                     'perfectly balanced launch would take (profiles/r05_experiments.txt)')
    line['watchdog_fired'] = False
    line['in_flight_ok'] = None
    if in_flight_region is not None:
        line['in_flight'] = in_flight_region
        line['in_flight_contexts'] = in_flight_region['contexts']
        line['value_in_flight'] = in_flight_region['value']
        line['ms_per_step_in_flight'] = in_flight_region['ms_per_step']
        line['ms_per_step_in_flight_steady'] = in_flight_region['ms_per_step_steady']
        line['value_in_flight_steady'] = P_job / (1e-3 * in_flight_region['ms_per_step_steady'])
        line['in_flight_converged'] = in_flight_check['converged']
        line['in_flight_alpha_solves'] = in_flight_check['alpha_solves']
        line['in_flight_left_to_finish'] = in_flight_check['left_to_finish']
        line['in_flight_audit_max'] = in_flight_check['audit_max']
        line['in_flight_evals_per_solve'] = in_flight_check['evals_per_solve']
        line['in_flight_ok'] = bool(in_flight_check['converged'] == in_flight_check['alpha_solves'] and
                                    in_flight_check['left_to_finish'] == 0 and in_flight_check['audit_max'] < AUDIT_GATE)
        pmc_fl, why_fl = (load_pmc(lib_hash, 'in_flight') if (default_workload and world == 1 and args.shard_of <= 1)
                          else (None, 'counters are collected for the default workload on one GPU only'))
        if pmc_fl:
            # the issue slots that are busy over the region with launches in flight: the counters of THAT cut (256 workgroups, four
            # pieces per scan) launched alone -- profiles/*_in_flight_pmc_summary.csv, same source hash -- x steps / elapsed time
            busy_fl = 4 * pmc_fl['valu_active_quadcycles'] + pmc_fl['mfma_busy_cycles'] - pmc_fl['coexec_cycles']
            roofline['frac_in_flight_region'] = busy_fl / (1e-3 * in_flight_region['ms_per_step']) / 1e9 / peak
            roofline['frac_in_flight_region_steady'] = busy_fl / (1e-3 * in_flight_region['ms_per_step_steady']) / 1e9 / peak
            roofline['frac_in_flight_source'] = pmc_fl['source']
        else:
            roofline['frac_in_flight_region'] = None
            roofline['frac_in_flight_null_reason'] = why_fl
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(batch, out)
        line['cpu_baseline']['all_cores'] = pool_baseline
    else:
        line['cpu_baseline'] = None
    if world == 1 and not args.no_extras:
        line['audit'] = audit_block(ctx)
        if default_workload:
            line['parity_per_alpha'] = parity_per_alpha()
            if args.shard_of <= 1:
                line['scaling_projection'] = scaling_projection(batch, opts, k_ms, args.n_alpha)
                # weak scaling (--scaling weak: one such batch per GPU): no shard gets smaller, every rank runs the step timed above
                # and the root additionally receives N compact packs.  A projection from this run's step time; the link rate
                # is the guide's figure (7 xGMI links x ~153 GB/s per GPU, ~76 GB/s per direction and link), not a measurement
                pack = (3 * P_rank + n_elem * (args.n_omega + 1)) * 8
                step_ms = 1e3 * elapsed / args.steps
                line['scaling_projection']['weak'] = {
                    'step_ms_one_gpu_measured': step_ms, 'compact_pack_bytes_per_rank': pack,
                    'assumed_link_GBs_per_direction': 76.0,
                    **{'N=%d' % N: dict(gather_ms_projected=1e3 * pack / 76e9,
                                        alpha_solves_per_s_projected=N * P_rank / (1e-3 * step_ms + pack / 76e9))
                       for N in (2, 4, 8)}}
    if use_comm:
        ctx.comm_destroy()
    ctx.close()
    if world == 1 and not args.no_extras:
        line['end_to_end'] = end_to_end_block(batch, args.n_orb, args.n_alpha)
        if default_workload and args.shard_of <= 1:
            line['underfilled'] = underfilled_block()
    exit_code = 0
    if n_fl_comm > 1:
        import threading
        line['in_flight_comm_region'] = dict(ran=False, contexts_per_rank=n_fl_comm, ms_per_step=None, value=None, error=None)

        def give_up():
            line['watchdog_fired'] = True
            line['in_flight_comm_region']['error'] = 'did not come back within the time limit (watchdog)'
            print(json.dumps(line))
            sys.stdout.flush()
            print('bench.py: the region with %d steps in flight on every rank did not come back within the time limit: the line is '
                  'that of the region with one context per rank; exit code %d' % (n_fl_comm, EXIT_WATCHDOG), file=sys.stderr)
            sys.stderr.flush()
            os._exit(EXIT_WATCHDOG)
        dog = threading.Timer(float(os.environ.get('MXE_BENCH_IN_FLIGHT_TIMEOUT', '180')), give_up)
        dog.daemon = True
        dog.start()
        try:
            e_fl, check_fl, t_comm = in_flight_comm_region(batch, mine, local_rank, rank, world, counts, full, args, n_fl_comm)
            dog.cancel()
            ok_fl = bool(check_fl['converged'] == check_fl['alpha_solves'] and check_fl['left_to_finish'] == 0 and
                         check_fl['audit_max'] < AUDIT_GATE)
            line['in_flight_comm_region'].update(ran=True, ms_per_step=1e3 * e_fl / args.steps, value=P_job * args.steps / e_fl,
                                                 check=check_fl, ok=ok_fl, communicators_init_s=t_comm)
            line['value_in_flight'] = line['in_flight_comm_region']['value']
            line['ms_per_step_in_flight'] = line['in_flight_comm_region']['ms_per_step']
            line['in_flight_contexts'] = n_fl_comm
            line['in_flight_ok'] = ok_fl
            if not ok_fl:
                exit_code = EXIT_IN_FLIGHT_CHECK
        except Exception as exc:
            dog.cancel()
            line['in_flight_comm_region']['error'] = repr(exc)
            print('bench.py: the region with steps in flight failed: %r' % (exc,), file=sys.stderr)
            exit_code = EXIT_IN_FLIGHT_FAILED
    if line.get('in_flight_ok') is False and exit_code == 0:
        print('bench.py: a batch in flight did not pass its check: %r' % (in_flight_check,), file=sys.stderr)
        exit_code = EXIT_IN_FLIGHT_CHECK
    print(json.dumps(line))
    if gather_checked is False:
        # every rank's chi2 as gathered must be what rank 0 gets when it solves the whole batch itself
        print('bench.py: the gathered results of the ranks differ from the one-GPU solve of the same batch', file=sys.stderr)
        sys.exit(EXIT_GATHER_MISMATCH)
    if exit_code:
        sys.exit(exit_code)


if __name__ == '__main__':
    main()
