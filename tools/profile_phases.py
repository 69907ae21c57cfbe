#!/usr/bin/env python3
"""Phase shares of the chain kernel from the in-kernel stamp build.

    make -C maxent_amd/csrc prof
    MAXENT_AMD_LIB=maxent_amd/lib/libmaxent_hip_prof.so python tools/profile_phases.py [--waves N]

Reads SHARES, not run time (the stamped build is slower than the real one).
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MAXENT_AMD_LIB', os.path.join(ROOT, 'maxent_amd', 'lib', 'libmaxent_hip_prof.so'))
import bench                      # noqa: E402
from maxent_amd import device     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--waves', type=int, default=0)
ap.add_argument('--n-orb', type=int, default=16)
ap.add_argument('--theta', type=float, default=1e-6)
ap.add_argument('--layout', type=int, default=1)
ap.add_argument('--split', type=int, default=1)
ap.add_argument('--n-omega', type=int, default=500)
ap.add_argument('--wgpc', type=int, default=0)
ap.add_argument('--home', action='store_true', help='library built with -DMXE_PROFILE_HOME: split the home phase')
ap.add_argument('--accept', action='store_true', help='library built with -DMXE_PROFILE_ACCEPT: split the accept step')
args = ap.parse_args()
batch = bench.build_batch(args.n_orb, 200, args.n_omega, 100, 0)
ctx = bench.stage(batch, 0)
n_chain = len(batch['elems'])
ctx.upload_chains(np.arange(n_chain, dtype=np.int32), batch['alphas'], batch['v0'],
                  device.default_opts(waves_per_chain=args.waves, decouple_tol=args.theta,
                                      chains_per_wg=args.layout, alpha_split=args.split, wg_per_cu=args.wgpc))
for _ in range(2):
    ctx.launch(); ctx.sync()
lib = device.load_library()
lib.mxe_prof_fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
info = ctx.last_launch_info()
n_sub = n_chain * 100            # upper bound on the pieces (alpha_split = 0: the library chooses)
prof = np.zeros((n_sub + 8192, 8), dtype=np.int64)
rc = lib.mxe_prof_fetch(ctx._h, prof.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
assert rc == 0
out = ctx.fetch(want_v=False, want_H=False)
print('kernel ms %.3f  info %s' % (ctx.last_kernel_ms(), info))
if args.layout == 4:
    # lock-step kernel: one row per (workgroup, wave); slot 7 of wave 0 = rounds, of the others = wait
    n_wg, nwv = info['n_workgroups'], info['waves_per_chain']
    pw = prof[:n_wg * 8].reshape(n_wg, 8, 8)[:, :nwv, :]
    rounds = pw[:, 0, 7].astype(float)
    names = ['row:matvec', 'fused:loop', 'home', 'row:exp+sums', 'fused:accum', 'accept', 'fused:shuffle', 'row:barrier']
    order = (2, 0, 3, 7, 1, 6, 4, 5)
    if args.accept:
        names = ['acc:h+rho+sums', 'acc:load slot', 'acc:decide', 'acc:outputs', 'acc:predictor+next', 'everything else', 'acc:store slot', 'row:barrier']
        order = (0, 1, 2, 3, 4, 6, 5)
    elif args.home:
        names = ['home:refill+rhs', 'home:load A', 'home:barrier', 'home:factor', 'home:solve', 'all passes', '#factorisations', 'row:barrier']
        order = (0, 1, 3, 4, 6, 2, 5)
    print('workgroups %d, rounds per workgroup: mean %.1f (min %d, median %d, max %d)' %
          (n_wg, rounds.mean(), rounds.min(), np.median(rounds), rounds.max()))
    out_all = ctx.fetch(want_v=False, want_H=False)
    print('slot use: %d evaluations / (4 slots x %d rounds) = %.3f' %
          (out_all['n_evals'].sum(), rounds.sum(), out_all['n_evals'].sum() / (4.0 * rounds.sum())))
    wg_cycles = pw[:, 0, :7].sum(axis=1)
    print('cycles per workgroup: mean %.3e, min %.3e, max %.3e' % (wg_cycles.mean(), wg_cycles.min(), wg_cycles.max()))
    top = np.argsort(-wg_cycles)[:8]
    print('slowest workgroups (index: rounds, cycles, cycles per round): ' +
          '; '.join('%d: %d, %.2e, %.0f' % (w, rounds[w], wg_cycles[w], wg_cycles[w] / rounds[w]) for w in top))
    by_r = np.argsort(-rounds)[:8]
    print('most rounds (index: rounds, cycles per round): ' + '; '.join('%d: %d, %.0f' % (w, rounds[w], wg_cycles[w] / rounds[w]) for w in by_r))
    print('cycles per round over workgroups: mean %.0f, min %.0f, max %.0f' % ((wg_cycles / rounds).mean(), (wg_cycles / rounds).min(), (wg_cycles / rounds).max()))
    print('cycles per round, by wave (mean over workgroups):')
    print('  %-16s' % 'phase' + ''.join('   wave%d' % w for w in range(nwv)))
    tot = np.zeros(nwv)
    for q in order:
        per = pw[:, :, q].sum(axis=0) / rounds.sum()
        if q == 7:
            per[0] = np.nan
        if q == 7 and not args.home:
            pass
        else:
            tot += per
        print('  %-16s' % names[q] + ''.join((' %7.2f' if names[q].startswith('#') else ' %7.0f') % x for x in per))
    print('  %-16s' % 'total' + ''.join(' %7.0f' % x for x in tot))
else:
    n_sub = n_chain * max(args.split, 1)
    prof = prof[:n_sub]
    names = ['prep', 'gram', 'chol+solve', 'step/norm', 'eval', 'accept', 'output', '-']
    iters = np.full(n_sub, out['n_iter'].sum() / n_sub)
    tot = prof.sum(axis=1)
    print('rows %d, newton iterations per row: mean %.1f' % (n_sub, iters.mean()))
    print('cycles per chain: mean %.3e (max %.3e)' % (tot.mean(), tot.max()))
    for q, nme in enumerate(names[:7]):
        print('  %-11s %6.2f %%   %8.0f cycles / newton iteration' % (nme, 100.0 * prof[:, q].sum() / tot.sum(),
                                                                     prof[:, q].sum() / iters.sum()))
