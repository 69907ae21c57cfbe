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
args = ap.parse_args()
batch = bench.build_batch(args.n_orb, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
n_chain = len(batch['elems'])
ctx.upload_chains(np.arange(n_chain, dtype=np.int32), batch['alphas'], batch['v0'],
                  device.default_opts(waves_per_chain=args.waves, decouple_tol=args.theta,
                                      chains_per_wg=args.layout, alpha_split=args.split))
for _ in range(2):
    ctx.launch(); ctx.sync()
lib = device.load_library()
lib.mxe_prof_fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
n_sub = n_chain * max(args.split, 1)
n_rows = ctx.last_launch_info()['n_workgroups'] if args.layout == 4 else n_sub
prof = np.zeros((n_sub, 8), dtype=np.int64)
rc = lib.mxe_prof_fetch(ctx._h, prof.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
assert rc == 0
prof = prof[:n_rows]
out = ctx.fetch(want_v=False, want_H=False)
names = ['prep', 'gram', 'chol+solve', 'step/norm', 'eval', 'accept', 'output', '-']
if args.layout == 4:
    rounds = prof[:, 7].copy(); prof[:, 7] = 0
    iters = rounds          # per workgroup: rounds (one Newton iteration of up to 4 chains each)
else:
    iters = np.full(n_rows, out['n_iter'].sum() / n_rows)
tot = prof.sum(axis=1)
print('kernel ms %.3f  info %s' % (ctx.last_kernel_ms(), ctx.last_launch_info()))
print('rows %d, newton iterations (rounds) per row: mean %.1f' % (n_rows, iters.mean()))
print('cycles per chain: mean %.3e (max %.3e)' % (tot.mean(), tot.max()))
for q, nme in enumerate(names[:7]):
    print('  %-11s %6.2f %%   %8.0f cycles / newton iteration' % (nme, 100.0 * prof[:, q].sum() / tot.sum(),
                                                                 prof[:, q].sum() / iters.sum()))
