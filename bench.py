#!/usr/bin/env python3
"""bench.py -- alpha-solves/s of the MI355X alpha-scan solver.

A "step" is one pass of the hot path (one launch of the chain kernel) over one
batch of synthetic input that is already resident in HBM.  At N=1 the workload
is BASELINE.json's cfg4 batch on ONE GPU: 16x16 matrix elements x 100 alpha
(25 600 alpha-solves; n_tau=200, n_omega=500, fp64) -- the batch the
north-star target is quoted on.  For N>1 every rank solves its own 16x16x100
batch (different noise seed per rank: weak scaling, no data-path collective
inside the solve); after each pass the per-alpha results chi2/S/Q/H are
gathered on rank 0 with one RCCL gather, inside the timed region.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 under
``python -m torch.distributed.run``); rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

# torchrun exports OMP_NUM_THREADS=1 when the variable is unset; with a single
# OpenMP thread the RCCL gather path of this script ran 2x slower per step on the
# MI355X box (measured: 14.1 vs 6.5 ms), so give the few host threads back.
if os.environ.get('TORCHELASTIC_RUN_ID') and os.environ.get('OMP_NUM_THREADS') == '1':
    os.environ['OMP_NUM_THREADS'] = str(max(1, min(8, (os.cpu_count() or 8) //
                                                   max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))))))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from maxent_amd import device, synthetic, hostprep   # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
N_EVAL_NOMINAL = {'normal': 160, 'plusminus': 84}
# HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md);
# filled in from profiles/ when measured, else None
TRAFFIC_PMC_BYTES_PER_LAUNCH = None      # see main(): set for the default workload


def build_batch(n_orb, n_tau, n_omega, n_alpha, rank):
    """Synthetic cfg3/cfg4 batch -> staged DeviceContext."""
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_omega,
                                                noise_seed=2025 + rank)
    t0 = time.perf_counter()
    K.reduce_singular_space(1e-14)
    t_svd = time.perf_counter() - t0
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    elems = [(i, j) for i in range(n_orb) for j in range(n_orb)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS
             for (i, j) in elems]
    v0n = hostprep.initial_v(K.V, D, omega.delta, device.ENTROPY_NORMAL)
    v0p = hostprep.initial_v(K.V, D, omega.delta, device.ENTROPY_PLUSMINUS)
    v0 = np.stack([v0n if k == device.ENTROPY_NORMAL else v0p for k in kinds])
    return dict(tau=tau, omega=omega, K=K, Gmat=Gmat, D=D, err=err,
                alphas=alphas, elems=elems, kinds=kinds, v0=v0, t_svd=t_svd)


def stage(batch, dev):
    K = batch['K']
    ctx = device.DeviceContext(K.U, K.S, K.V, device=dev)
    ds = ctx.add_dataset(batch['err'])
    n_elem = len(batch['elems'])
    ctx.set_elements([ds] * n_elem,
                     [batch['Gmat'][i, j] for (i, j) in batch['elems']],
                     np.tile(batch['D'], (n_elem, 1)), batch['kinds'])
    return ctx


class _DevArray(object):
    """zero-copy view of a library-owned device buffer for torch."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr,
                                             data=(int(ptr), False), version=2)


def cpu_baseline(batch, out_gpu, n_chains=2):
    """The oracle port of the reference's algorithm, timed on the host on a
    bounded sample (first diagonal + first off-diagonal element, all alphas).
    Also reports how far the GPU result is from it and from the
    extended-precision truth on that sample."""
    from oracle import ref_numpy as R, hp_truth
    try:                                  # really one core: no BLAS worker threads for the 56 x 500 products
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    K = batch['K']
    n_alpha = len(batch['alphas'])
    n_tau = len(batch['tau'])
    mesh = batch['alphas'] / n_tau
    picks = [0, 1][:n_chains]
    t_total, solves, err_ref, err_truth = 0.0, 0, 0.0, 0.0
    for c in picks:
        i, j = batch['elems'][c]
        ent = 'normal' if batch['kinds'][c] == device.ENTROPY_NORMAL else 'plusminus'
        p = R.Problem(np.array(K.K), K.U, K.S, K.V, batch['Gmat'][i, j],
                      batch['err'], batch['D'], entropy=ent)
        timing = []
        ref = R.alpha_loop(p, batch['omega'].delta, mesh, timing=timing)
        t_total += timing[0]
        solves += n_alpha
        H = out_gpu['H'][c]
        err_ref = max(err_ref, float(np.max(
            np.linalg.norm(H - ref['H'], axis=1) / np.linalg.norm(ref['H'], axis=1))))
        for ia in (0, n_alpha // 2, n_alpha - 1):
            _, Ht = hp_truth.polish(p.K, p.G, p.err, p.D, p.V, p.S,
                                    batch['alphas'][ia], out_gpu['v'][c, ia],
                                    ent, iters=4)
            err_truth = max(err_truth, float(np.linalg.norm(H[ia] - Ht) /
                                             np.linalg.norm(Ht)))
    return dict(value=solves / t_total, unit='alpha-solves/s', cores=1,
                kind='port',
                sample='oracle/ref_numpy.py (step-faithful numpy port of '
                       'LevenbergMinimizer + MaxEntCostFunction) on elements '
                       '(0,0) normal and (0,1) plusminus x %d alpha, %.1f s'
                       % (n_alpha, t_total),
                gpu_vs_port_max_rel_l2=err_ref,
                gpu_vs_extended_precision_truth_max_rel_l2=err_truth)


def _pool_worker(c):
    """one alpha scan of element c with the oracle port (forked worker: inherits _POOL_BATCH)"""
    try:                                  # one BLAS thread per worker (the pool is the parallelism)
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
    """the same oracle port on all host cores, one process per core, one matrix element
    (100 alpha) per process.  Runs BEFORE this process touches the GPU (workers are forked)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # a step is 1.7 ms: 200 of them keep the timed region long enough (0.33 s) for the tens of
    # milliseconds by which a fresh submission is sometimes picked up late on the MI355X boxes (seen in a
    # quarter of the processes: device time of the region unchanged, host-side wait 33-47 ms longer)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--n-orb', type=int, default=16)
    ap.add_argument('--n-tau', type=int, default=200)
    ap.add_argument('--n-omega', type=int, default=500)
    ap.add_argument('--n-alpha', type=int, default=100)
    ap.add_argument('--waves-per-chain', type=int, default=0)
    ap.add_argument('--chains-per-wg', type=int, default=0)
    ap.add_argument('--alpha-split', type=int, default=0)
    ap.add_argument('--wg-per-cu', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise torch.distributed and run the RCCL gather even with one rank (plumbing test)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d; launch with '
                  'python -m torch.distributed.run --nproc-per-node %d'
                  % (args.gpus, world, args.gpus), file=sys.stderr)
        if args.gpus > 1:
            sys.exit(2)

    pool_baseline = None
    if world == 1 and not args.force_dist and not args.no_cpu_baseline:
        # host cores first, before anything initialises the GPU in this process (forked workers)
        pool_baseline = cpu_pool_baseline(build_batch(args.n_orb, args.n_tau, args.n_omega, args.n_alpha, rank))

    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))

    if device.device_count() < 1:
        raise SystemExit('bench.py needs a GPU: the solver has no CPU fallback')

    batch = build_batch(args.n_orb, args.n_tau, args.n_omega, args.n_alpha, rank)
    ctx = stage(batch, local_rank)
    n_chain = len(batch['elems'])
    P = n_chain * args.n_alpha
    opts = device.default_opts(waves_per_chain=args.waves_per_chain,
                               chains_per_wg=args.chains_per_wg,
                               alpha_split=args.alpha_split, wg_per_cu=args.wg_per_cu)
    ctx.upload_chains(np.arange(n_chain, dtype=np.int32), batch['alphas'],
                      batch['v0'], opts)

    gather_bufs = None
    if use_dist:
        import torch
        nw = args.n_omega
        packs, gathered = [], []
        for b in (0, 1):
            # H, chi2, S, Q are contiguous in one device allocation of the library
            ctx.set_result_buffer(b)
            ptrs = ctx.result_device_ptrs()
            assert ptrs['chi2'] == ptrs['H'] + P * nw * 8 and ptrs['Q'] == ptrs['H'] + (P * nw + 2 * P) * 8
            t = torch.as_tensor(_DevArray(ptrs['H'], (P * nw + 3 * P,), '<f8'), device='cuda')
            assert t.data_ptr() == ptrs['H'], 'zero-copy view of the result buffer failed'
            packs.append(t)
            gathered.append([torch.empty_like(t) for _ in range(world)] if rank == 0 else None)
        gather_bufs = dict(packs=packs, gathered=gathered, pending=[None, None], k=0,
                           ext=torch.cuda.ExternalStream(ctx.stream_handle(), device=torch.device('cuda', local_rank)))

    def one_step():
        """one pass of the solver; with several ranks the ONE RCCL gather of the
        packed per-alpha results of pass k runs while pass k+1 computes into the
        other result buffer (the gather of pass k-1 is waited for first)."""
        if not use_dist:
            ctx.launch()        # enqueued back to back: no host synchronisation inside the timed region
            return
        g = gather_bufs
        b = g['k'] % 2
        g['k'] += 1
        cur = torch.cuda.current_stream()
        if g['pending'][b] is not None:
            # buffer b is free again once its gather has run: the wait is a dependency of the current
            # stream, passed on to the library's stream with an event -- no host synchronisation
            g['pending'][b].wait()
            ev = torch.cuda.Event()
            ev.record(cur)
            g['ext'].wait_event(ev)
        ctx.set_result_buffer(b)
        ctx.launch()
        done = torch.cuda.Event()
        done.record(g['ext'])
        cur.wait_event(done)                  # the gather is enqueued behind the pass that fills its source
        g['pending'][b] = dist.gather(g['packs'][b], g['gathered'][b], dst=0, async_op=True)

    def drain():
        if not use_dist:
            ctx.sync()
        if use_dist:
            for w in gather_bufs['pending']:
                if w is not None:
                    w.wait()
            torch.cuda.synchronize()

    def barrier():
        if use_dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    if use_dist:
        import torch
    for _ in range(args.warmup):
        one_step()
    drain()
    # the first collectives of a process group finish initialising in the background
    # (measured with one rank: a single 75 ms stall of one launch, 20-40 ms after the
    # first barrier); take the barrier here and let that settle outside the timed region
    # ... and without a process group: with only a few warm-up passes, the first submission after the
    # synchronisation that opens the timed region was picked up 30-50 ms late in a quarter of the
    # processes on the MI355X boxes (12 of 12 clean with ten warm-up passes).  Both settle in untimed
    # passes: half a second of them with a process group, a quarter of a second without.
    barrier()
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < (0.5 if use_dist else 0.25):
        one_step()
        if not use_dist:
            ctx.sync()
    drain()
    kernel_ms = []
    barrier()
    if not use_dist:
        ctx.sync()
        ctx.timing_mark()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    t_enq = time.perf_counter()
    drain()                     # every gather has landed on rank 0 inside the timed region
    if use_dist:
        kernel_ms = [ctx.last_kernel_ms()]      # the last pass of the region (HIP events on the library's stream)
    if not use_dist:
        # device time from the first launch of the region to the end of the last one / steps (HIP events
        # on the library's stream; includes the few microseconds between consecutive launches)
        kernel_ms = [ctx.ms_since_mark() / args.steps]
    t_drained = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    host_split = dict(enqueue_ms=1e3 * (t_enq - t0), wait_ms=1e3 * (t_drained - t_enq))
    if use_dist:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    out = ctx.fetch()
    info = ctx.last_launch_info()
    n_conv = int(out['converged'].sum())
    total_solves = P * world
    value = total_solves * args.steps / elapsed
    k_ms = float(np.mean(kernel_ms))

    # algorithmic bytes per launch: one cost evaluation pass
    # streams V once: B_eval = n_omega*n_s*8 + 2*n_omega*8
    n_s = ctx.n_s
    b_eval = args.n_omega * n_s * 8 + 2 * args.n_omega * 8
    n_diag = sum(1 for k in batch['kinds'] if k == device.ENTROPY_NORMAL)
    n_off = n_chain - n_diag
    bytes_nominal = args.n_alpha * b_eval * (n_diag * N_EVAL_NOMINAL['normal'] +
                                             n_off * N_EVAL_NOMINAL['plusminus'])
    bytes_actual = float(out['n_evals'].sum()) * b_eval
    # every Newton iteration also streams V once more for the Gram matrix
    bytes_streamed = float(out['n_evals'].sum() + out['n_iter'].sum()) * b_eval
    # achieved: evaluation passes the launch actually executed (counted by the
    # kernel) x B_eval / kernel time.  Extra fields: the same with the Gram
    # passes (each Newton iteration streams the active columns of V once
    # more), and SURVEY 8d's nominal figure that prices the kernel's time
    # against the reference's 160/84 passes per alpha-solve.
    achieved = bytes_actual / (k_ms * 1e-3) / 1e9
    # profiles/r01_f_pmc_hbm_traffic.csv: FETCH_SIZE 8 063 KB (x2, gfx950
    # correction) + WRITE_SIZE 116 548 KB per launch of the default workload
    traffic = (2 * 8063.0e3 + 116548.0e3) if (args.n_orb, args.n_tau, args.n_omega, args.n_alpha) == (16, 200, 500, 100) else None
    roofline = dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS,
                    unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic,
                    kernel=info['kernel'], kernel_ms=k_ms,
                    definition='evaluation passes executed (kernel counter) x '
                               'B_eval = %d B / kernel time (HIP events).  V (224 KB) is '
                               'L2 resident: the algorithmic rate may exceed the HBM peak '
                               '(frac > 1), the HBM traffic measured with PMC counters is '
                               'the `traffic` bytes per launch (`hbm_measured`); the kernel '
                               'is bound by the matrix / vector pipes, see DESIGN.md '
                               'section 4' % b_eval,
                    hbm_measured=(None if traffic is None else traffic / (k_ms * 1e-3) / 1e9),
                    algorithmic_bytes_per_launch=bytes_actual,
                    achieved_incl_gram_passes=bytes_streamed / (k_ms * 1e-3) / 1e9,
                    achieved_survey_nominal_reference_work=bytes_nominal / (k_ms * 1e-3) / 1e9,
                    evals_per_solve=float(out['n_evals'].mean()),
                    newton_iters_per_solve=float(out['n_iter'].mean()),
                    fp64_tflops_gram_dense_equiv=float(out['n_iter'].sum()) * 2.0 *
                    args.n_omega * n_s * n_s / (k_ms * 1e-3) / 1e12)

    line = dict(metric='alpha-solves/s', value=value, unit='alpha-solves/s',
                n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                config=dict(workload='cfg4: ElementwiseMaxEnt %dx%d G(tau), %d '
                                     'chains x %d alpha = %d alpha-solves per GPU, '
                                     'n_tau=%d n_omega=%d n_s=%d'
                                     % (args.n_orb, args.n_orb, n_chain,
                                        args.n_alpha, P, args.n_tau,
                                        args.n_omega, n_s),
                            waves_per_chain=info['waves_per_chain'],
                            workgroups=info['n_workgroups'],
                            lds_bytes=info['lds_bytes'],
                            converged=n_conv, problems=P,
                            gather='one torch.distributed nccl (RCCL) gather per step of the packed H, chi2, S, Q '
                                   '(%.1f MB per rank) to rank 0, double buffered against the next pass, inside the timed region' % ((P * args.n_omega + 3 * P) * 8 / 1e6)
                            if use_dist else 'none (1 GPU)',
                            svd_seconds_host=batch['t_svd'], host_split=host_split),
                roofline=roofline)
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(batch, out)
        line['cpu_baseline']['all_cores'] = pool_baseline
    else:
        line['cpu_baseline'] = None
    if use_dist:
        # rank 0 holds every rank's results: check its own block against the source
        for b in (0, 1):
            assert torch.equal(gather_bufs['gathered'][b][0], gather_bufs['packs'][b]), 'gathered block differs'
        line['config']['gather_checked'] = True
    print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
