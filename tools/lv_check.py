#!/usr/bin/env python3
"""chain_kernel_lv (V^T resident in LDS as binary32) against the paths it replaces, on the launches that do not fill the GPU:
BASELINE cfg2 (one scan), cfg3 (4 x 4), the shards of cfg4 / N.  For every case and option set: kernel time (HIP events over
whole launches), which kernels ran, converged / evaluations / left to the finishing pass, the all-problem audit (exact Newton
correction at the returned v) and the distance of H from the single-pass binary64 result.

    python tools/lv_check.py [--steps 50] [--cases cfg2 cfg3 shard8 ...] [--f32]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device, synthetic

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=50)
ap.add_argument('--cases', nargs='*', default=['cfg2', 'cfg3', 'shard8', 'shard4', 'shard2'])
ap.add_argument('--f32', action='store_true', help='also the binary32 launches (precision = F32), old and new kernel')
ap.add_argument('--full', action='store_true', help='also the whole cfg4 batch (with lds_basis = 1: not the default there)')
args = ap.parse_args()


def timed(ctx, steps):
    for _ in range(3):
        ctx.launch()
    ctx.sync()
    ctx.timing_mark()
    for _ in range(steps):
        ctx.launch()
    return ctx.ms_since_mark() / steps


def run(batch, which, **opts):
    ctx = bench.stage(batch, 0, which)
    ctx.upload_chains(np.arange(len(which), dtype=np.int32), batch['alphas'], batch['v0'][which], device.default_opts(**opts))
    ms = timed(ctx, args.steps)
    ctx.launch()
    info = ctx.last_launch_info()
    left = ctx.finish()
    out = ctx.fetch(want_v=False, want_H=True)
    aud = ctx.audit()['corr'].ravel()
    ctx.close()
    return dict(ms=ms, kernel=info['kernel'], n_wg=info['n_workgroups'], left=left, conv=int(out['converged'].sum()), n=out['converged'].size,
                evals=float(out['n_evals'].sum()), iters=float(out['n_iter'].sum()), H=np.array(out['H']), chi2=out['chi2'].copy(),
                aud_max=float(np.nanmax(aud)), aud_p99=float(np.nanpercentile(aud, 99)))


def relerr(H, H0):
    return np.linalg.norm(H - H0, axis=-1) / np.linalg.norm(H0, axis=-1)


def case(name):
    if name == 'cfg2':
        batch = bench.build_batch(2, 200, 500, 100, 0)
        _, _, _, G1 = synthetic.single_G(200, 500)
        batch['Gmat'] = G1[None, None, :]
        batch['elems'], batch['kinds'], batch['v0'] = [(0, 0)], batch['kinds'][:1], batch['v0'][:1]
        return batch, [0]
    if name == 'cfg3':
        batch = bench.build_batch(4, 200, 500, 100, 0)
        return batch, list(range(16))
    if name.startswith('shard'):
        N = int(name[5:])
        batch = bench.build_batch(16, 200, 500, 100, 0)
        r = {8: 7, 4: 3, 2: 1}.get(N, 0)          # (the slowest ranks of r03)
        return batch, [e for e in range(256) if e % N == r]
    if name == 'cfg4':
        batch = bench.build_batch(16, 200, 500, 100, 0)
        return batch, list(range(256))
    raise ValueError(name)


cases = list(args.cases) + (['cfg4'] if args.full else [])
for name in cases:
    batch, which = case(name)
    print('== %s: %d scans x %d alphas' % (name, len(which), len(batch['alphas'])), flush=True)
    ref = run(batch, which, lds_basis=2)
    sets = [('two-pass (lds_basis = 1)', dict(lds_basis=1))]
    if args.f32:
        sets += [('binary32, one-chain kernel (lds_basis = 2)', dict(precision=device.PRECISION_F32, lds_basis=2)),
                 ('binary32, chain_kernel_lv', dict(precision=device.PRECISION_F32))]
    rows = [('binary64 single pass (lds_basis = 2)', ref)] + [(lab, run(batch, which, **o)) for lab, o in sets]
    for lab, r in rows:
        e = relerr(r['H'], ref['H'])
        print('  %-44s %.3f ms  conv %d/%d  evals %.0f  iters %.0f  left %d  audit max %.2e p99 %.2e  |H - H64| max %.2e p99 %.2e  wg %d\n      %s' %
              (lab, r['ms'], r['conv'], r['n'], r['evals'], r['iters'], r['left'], r['aud_max'], r['aud_p99'],
               float(np.nanmax(e)), float(np.nanpercentile(e, 99)), r['n_wg'], r['kernel']), flush=True)
