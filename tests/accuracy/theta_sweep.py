"""cost / accuracy of the active-subspace threshold theta (decouple_tol), tol_h and the stopping rule"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from maxent_amd import device
from oracle import hp_truth
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
K = batch['K']
truth = {}
GRID = ((1e-6, 1e-9, 0), (1e-6, 1e-9, 1), (1e-5, 1e-9, 1), (1e-4, 1e-9, 1), (3e-4, 1e-9, 1), (1e-3, 1e-9, 1),
        (1e-6, 1e-8, 1), (1e-6, 1e-7, 1), (1e-6, 1e-6, 1), (1e-4, 1e-7, 1))
for theta, tol, est in GRID:
    out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'],
                           device.default_opts(tol_h=tol, decouple_tol=theta, stop_estimate=est))
    ms = []
    for _ in range(3):
        ctx.launch(); ctx.sync(); ms.append(ctx.last_kernel_ms())
    worst = 0.0
    for c in (0, 1, 17, 40, 100, 255):
        i, j = batch['elems'][c]
        ent = 'normal' if batch['kinds'][c] == 0 else 'plusminus'
        for ia in (0, 5, 12, 37, 50, 70, 87, 99):
            key = (c, ia)
            if key not in truth:
                truth[key] = hp_truth.polish(np.array(K.K), batch['Gmat'][i, j], batch['err'], batch['D'], K.V, K.S,
                                             batch['alphas'][ia], out['v'][c, ia], ent, iters=4)[1]
            Ht = truth[key]
            worst = max(worst, np.linalg.norm(out['H'][c, ia] - Ht) / np.linalg.norm(Ht))
    print('theta %.0e tol_h %.0e estimate %d: kernel %.3f ms  iters/solve %.3f  max iters %d  converged %d  max rel L2 vs truth %.2e' % (
        theta, tol, est, min(ms), out['n_iter'].mean(), out['n_iter'].max(), out['converged'].sum(), worst))
