"""kernel time / iterations of the cfg4 batch against Bryan's step bound (mxe_opts.step_max)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
ref = None
for sm in (0.2, 0.3, 0.4, 0.5, 0.6, 0.2):
    ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(step_max=sm))
    ms = []
    for _ in range(40):
        ctx.launch(); ctx.sync(); ms.append(ctx.last_kernel_ms())
    out = ctx.fetch()
    if ref is None:
        ref = out['H']
    e = np.linalg.norm(out['H'] - ref, axis=-1) / np.linalg.norm(ref, axis=-1)
    print('step_max %.2f: kernel %.3f ms  iters/solve %.3f  evals/solve %.3f  max iters %d  converged %d  max rel L2 vs step_max 0.2: %.1e' % (
        sm, float(np.median(ms[10:])), out['n_iter'].mean(), out['n_evals'].mean(), out['n_iter'].max(), out['converged'].sum(), e.max()))
