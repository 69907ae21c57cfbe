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
args = ap.parse_args()
batch = bench.build_batch(args.n_orb, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
n_chain = len(batch['elems'])
ctx.upload_chains(np.arange(n_chain, dtype=np.int32), batch['alphas'], batch['v0'],
                  device.default_opts(waves_per_chain=args.waves, decouple_tol=args.theta))
for _ in range(2):
    ctx.launch(); ctx.sync()
lib = device.load_library()
lib.mxe_prof_fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
prof = np.zeros((n_chain, 8), dtype=np.int64)
rc = lib.mxe_prof_fetch(ctx._h, prof.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
assert rc == 0
out = ctx.fetch(want_v=False, want_H=False)
names = ['prep', 'gram', 'chol+solve', 'step/norm', 'eval', 'accept', 'output', '-']
tot = prof.sum(axis=1)
iters = out['n_iter'].sum(axis=1)
print('kernel ms %.3f  info %s' % (ctx.last_kernel_ms(), ctx.last_launch_info()))
print('chains %d, newton iterations per chain: mean %.1f' % (n_chain, iters.mean()))
print('cycles per chain: mean %.3e (max %.3e)' % (tot.mean(), tot.max()))
for q, nme in enumerate(names[:7]):
    print('  %-11s %6.2f %%   %8.0f cycles / newton iteration' % (nme, 100.0 * prof[:, q].sum() / tot.sum(),
                                                                 prof[:, q].sum() / iters.sum()))
