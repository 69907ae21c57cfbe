#!/usr/bin/env python3
"""Kernel time, Newton iterations and the ALL-problem accuracy audit (mxe_audit) of the cfg4 batch for a
sweep of the stopping tolerance and the decoupling threshold."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device

batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
n_chain = len(batch['elems'])
for wg in (1, 2):
    for tol in (1e-9, 3e-9, 1e-8, 1e-7):
        for theta in (1e-5, 1e-4):
            opts = device.default_opts(tol_h=tol, decouple_tol=theta, wg_per_cu=wg)
            ctx.upload_chains(np.arange(n_chain, dtype=np.int32), batch['alphas'], batch['v0'], opts)
            ms = []
            for _ in range(6):
                ctx.launch(); ctx.sync(); ms.append(ctx.last_kernel_ms())
            out = ctx.fetch(want_v=False, want_H=False)
            a = ctx.audit()
            c = a['corr'].ravel()
            print('wg %d tol_h %.0e theta %.0e: kernel %.3f ms, iters/solve %.3f, converged %d; audit corr max %.2e p99 %.2e median %.2e, > 1e-6: %d, gmax max %.1e'
                  % (wg, tol, theta, min(ms), out['n_iter'].mean(), int(out['converged'].sum()), c.max(), np.percentile(c, 99),
                     np.median(c), int((c > 1e-6).sum()), a['gmax'].max()), flush=True)
