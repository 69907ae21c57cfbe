"""where the kernel time of a small batch goes: Newton iterations per piece (the persistent grid ends
with its slowest workgroup).  python tools/tail_diag.py [n_orb] [alpha_split]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
n_orb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
split = int(sys.argv[2]) if len(sys.argv) > 2 else 0
batch = bench.build_batch(n_orb, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
nc = len(batch['elems'])
for sp in ([split] if split else [0, 8, 4, 2, 1]):
    ctx.upload_chains(np.arange(nc), batch['alphas'], batch['v0'], device.default_opts(alpha_split=sp))
    ms = []
    for _ in range(4):
        ctx.launch(); ctx.sync(); ms.append(ctx.last_kernel_ms())
    out = ctx.fetch()
    it = out['n_iter'] + 0
    ev = out['n_evals']
    info = ctx.last_launch_info()
    print('n_orb %d alpha_split %d: kernel %.3f ms, %d workgroups x %d waves, iterations/solve %.2f, evals/solve %.2f, '
          'max iterations of one alpha %d, scans with the most evaluations: %s' %
          (n_orb, sp, min(ms[1:]), info['n_workgroups'], info['waves_per_chain'], it.mean(), ev.mean(), it.max(),
           np.sort(ev.sum(axis=1))[-4:]))
    worst = np.unravel_index(np.argmax(ev), ev.shape)
    print('   worst alpha-solve: element %s alpha index %d: %d iterations, %d evaluations' %
          (batch['elems'][worst[0]], worst[1], it[worst], ev[worst]))
