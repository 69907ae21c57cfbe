"""Newton iterations per piece of the default scan split, by entropy kind and position in the scan
(calibration data for the queue order of the persistent grid)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
split = int(sys.argv[1]) if len(sys.argv) > 1 else 14
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'],
                       device.default_opts(alpha_split=split), want_v=False, want_H=False)
ne = out['n_evals']
kinds = np.array(batch['kinds'])
bounds = [int(100 * s / split) for s in range(split + 1)]
print('piece: alpha range, evaluations per piece (mean, min, max) normal | plusminus')
for s in range(split):
    a0, a1 = bounds[s], bounds[s + 1]
    tn = ne[kinds == 0][:, a0:a1].sum(axis=1)
    tp = ne[kinds == 1][:, a0:a1].sum(axis=1)
    print('%2d: alpha[%2d:%3d] (%.3g .. %.3g)  normal %5.1f %3d %3d | plusminus %5.1f %3d %3d   first alpha: %4.1f | %4.1f' % (
        s, a0, a1, batch['alphas'][a0], batch['alphas'][a1 - 1], tn.mean(), tn.min(), tn.max(), tp.mean(), tp.min(), tp.max(),
        ne[kinds == 0][:, a0].mean(), ne[kinds == 1][:, a0].mean()))
