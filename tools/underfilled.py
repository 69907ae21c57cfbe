#!/usr/bin/env python3
"""Launches that do not fill the GPU, one process, kernel time by HIP events:
  * BASELINE cfg2 (one scan of 100 alpha), cfg3 (4 x 4 matrix: 16 scans),
  * EVERY rank's shard of the cfg4 batch split over N = 2, 4, 8 GPUs (element e on rank e mod N): a job ends with
    its slowest rank.
--waves 4 8 compares the lock-step kernel without and with helper waves (mxe_opts.waves_per_chain)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device

ap = argparse.ArgumentParser()
ap.add_argument('--waves', type=int, nargs='*', default=[0])
ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--shards', type=int, nargs='*', default=[2, 4, 8])
ap.add_argument('--split', type=int, default=0, help='mxe_opts.alpha_split (0: the library decides)')
args = ap.parse_args()


def timed(ctx, steps):
    for _ in range(5):
        ctx.launch()
    ctx.sync()
    ctx.timing_mark()
    for _ in range(steps):
        ctx.launch()
    return ctx.ms_since_mark() / steps


def run(batch, which, waves, steps):
    ctx = bench.stage(batch, 0, which)
    ctx.upload_chains(np.arange(len(which), dtype=np.int32), batch['alphas'], batch['v0'][which],
                      device.default_opts(waves_per_chain=waves, alpha_split=args.split))
    ms = timed(ctx, steps)
    ctx.launch()
    info = ctx.last_launch_info()
    left = ctx.finish()
    out = ctx.fetch(want_v=False, want_H=False)
    ctx.close()
    return ms, info, int(out['converged'].sum()), out['converged'].size, float(out['n_evals'].sum()), left


for waves in args.waves:
    print('== mxe_opts.waves_per_chain = %d' % waves, flush=True)
    for name, n_orb, sel in (('cfg2: one scan', 1, None), ('cfg3: 4 x 4', 4, None)):
        batch = bench.build_batch(max(n_orb, 2), 200, 500, 100, 0)
        if n_orb == 1:
            # BASELINE cfg2: the two-Gaussian spectrum of SURVEY 8(d) (synthetic.single_G), one normal-entropy scan
            from maxent_amd import synthetic
            _, _, _, G1 = synthetic.single_G(200, 500)
            batch['Gmat'] = G1[None, None, :]
            batch['elems'], batch['kinds'], batch['v0'] = [(0, 0)], batch['kinds'][:1], batch['v0'][:1]
        which = list(range(len(batch['elems'])))
        ms, info, nc, n, ev, left = run(batch, which, waves, args.steps)
        print('%-16s %5d solves  kernel %.3f ms  %s  workgroups %d  converged %d/%d  evals %.0f  left to finish %d' %
              (name, n, ms, info['kernel'], info['n_workgroups'], nc, n, ev, left), flush=True)
    batch = bench.build_batch(16, 200, 500, 100, 0)
    n_elem = len(batch['elems'])
    ms1, info, nc, n, ev, left = run(batch, list(range(n_elem)), waves, args.steps)
    print('%-16s %5d solves  kernel %.3f ms  %s' % ('cfg4: N = 1', n, ms1, info['kernel']), flush=True)
    for N in args.shards:
        row = []
        for r in range(N):
            which = [e for e in range(n_elem) if e % N == r]
            ms, info, nc, n, ev, left = run(batch, which, waves, max(20, args.steps // 2))
            row.append(ms)
            assert nc == n, (N, r, nc, n)
        print('cfg4 / %d: slowest rank %.3f ms (rank %d), fastest %.3f, all: %s   %s   -> %.2fx before the gather' %
              (N, max(row), int(np.argmax(row)), min(row), ' '.join('%.3f' % x for x in row), info['kernel'], ms1 / max(row)),
              flush=True)
