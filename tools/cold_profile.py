"""Cost of a cold start at every alpha of the scan (cfg4 batch): every alpha is its own piece (alpha_split = n_alpha),
so n_evals[scan][i] = evaluations from the default model to the minimiser of alpha_i.  Calibration data of the
piece planner (maxent_hip.hip: piece_cold)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(alpha_split=100),
                       want_v=False, want_H=False)
ne = out['n_evals']
kinds = np.array(batch['kinds'])
print('converged', int(out['converged'].sum()), 'of', out['converged'].size)
for k, name in ((0, 'normal'), (1, 'plusminus')):
    m = ne[kinds == k]
    print(name, 'cold start, evaluations by alpha index: mean')
    print(' '.join('%.0f' % x for x in m.mean(axis=0)))
    print(name, 'max over scans')
    print(' '.join('%d' % x for x in m.max(axis=0)))
m = ne[kinds == 0][:, 92:]
it = out['n_iter'][kinds == 0][:, 92:]
print('normal scans, alpha index 92..99: evaluations (Newton iterations) per scan')
for r in range(m.shape[0]):
    print('  scan %2d: ' % r + ' '.join('%3d(%3d)' % (a, b) for a, b in zip(m[r], it[r])))
