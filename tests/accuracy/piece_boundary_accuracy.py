"""accuracy against the extended-precision truth right at the cold-started first alpha of every piece
of the default scan split (cfg4 batch, default options)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from maxent_amd import device
from oracle import hp_truth
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
K = batch['K']
out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
print('iters/solve %.3f, max iters %d, converged %d' % (out['n_iter'].mean(), out['n_iter'].max(), out['converged'].sum()))
split = 10
firsts = sorted(set(int(100 * s / split) for s in range(split)))
worst = 0.0
for c in (0, 1, 17, 100, 255):
    i, j = batch['elems'][c]
    ent = 'normal' if batch['kinds'][c] == 0 else 'plusminus'
    for ia in firsts + [99]:
        Ht = hp_truth.polish(np.array(K.K), batch['Gmat'][i, j], batch['err'], batch['D'], K.V, K.S,
                             batch['alphas'][ia], out['v'][c, ia], ent, iters=4)[1]
        e = np.linalg.norm(out['H'][c, ia] - Ht) / np.linalg.norm(Ht)
        worst = max(worst, e)
        if e > 1e-9:
            print('chain %d alpha index %d: n_iter %d, rel L2 vs truth %.2e' % (c, ia, out['n_iter'][c, ia], e))
print('worst rel L2 vs truth at piece starts: %.2e' % worst)
