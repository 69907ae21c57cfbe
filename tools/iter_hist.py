"""Newton iterations per alpha by position inside the cold-started pieces (cfg4 batch, default options)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
it = out['n_iter']
split = 10
starts = [int(100 * s / split) for s in range(split)] + [100]
pos = np.zeros(100, dtype=int)
for a, b in zip(starts[:-1], starts[1:]):
    pos[a:b] = np.arange(b - a)
kinds = np.array(batch['kinds'])
for name, sel in (('diagonal (normal)', kinds == 0), ('off-diagonal (plusminus)', kinds == 1)):
    print(name, 'mean iterations by position in the piece:',
          ' '.join('%d:%.2f' % (p, it[sel][:, pos == p].mean()) for p in range(pos.max() + 1)))
    print('   histogram of iterations at positions >= 2:', np.bincount(it[sel][:, pos >= 2].ravel())[:8])
    print('   mean iterations at positions >= 2 by alpha index decile:',
          ' '.join('%.2f' % it[sel][:, (pos >= 2) & (np.arange(100) // 10 == d)].mean() for d in range(10)))
