"""histogram of the active-block size n_act over the cfg4 batch (weighted by Newton iterations)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
theta = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'],
                       device.default_opts(decouple_tol=theta), want_v=False, want_H=False)
na, it = ctx.fetch_n_act().ravel(), out['n_iter'].ravel()
print('theta %.0e: n_act min %d max %d mean %.1f; iterations %d' % (theta, na.min(), na.max(), na.mean(), it.sum()))
h = np.bincount(na, weights=it, minlength=65)
c = np.cumsum(h) / h.sum()
for n in range(65):
    if h[n] > 0:
        print('  n_act %2d: %5.1f %%  (cum %5.1f %%)' % (n, 100 * h[n] / h.sum(), 100 * c[n]))
