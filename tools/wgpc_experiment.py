"""experiment: one vs two workgroups per CU on a problem whose LDS footprint allows both
(n_omega = 180).  MAXENT_AMD_LIB selects the library build."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
n_omega = int(sys.argv[1]) if len(sys.argv) > 1 else 180
batch = bench.build_batch(16, 200, n_omega, 100, 0)
ctx = bench.stage(batch, 0)
out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], want_v=False)
ms = []
for _ in range(5):
    ctx.launch(); ctx.sync(); ms.append(ctx.last_kernel_ms())
print(os.environ.get('MAXENT_AMD_LIB', 'default'), 'n_omega', n_omega, 'kernel ms %.3f' % min(ms), ctx.last_launch_info(),
      'iters/solve %.3f' % out['n_iter'].mean(), 'converged', int(out['converged'].sum()))
