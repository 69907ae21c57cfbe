"""throughput of back-to-back cfg4 batches: one context (one stream) against two or three contexts taking the batches in turn
(their kernels overlap where one's workgroups have finished and the other's can start)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
n = len(batch['elems'])
def make():
    c = bench.stage(batch, 0)
    c.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts())
    return c
ctxs = [make() for _ in range(3)]
for c in ctxs:
    for _ in range(3):
        c.launch(); c.select_launch(0)
    c.sync()
for n_ctx in (1, 2, 3, 1, 2):
    use = ctxs[:n_ctx]
    K = 600
    for c in use: c.sync()
    t0 = time.perf_counter()
    for k in range(K):
        c = use[k % n_ctx]
        c.launch(); c.select_launch(0)
    for c in use: c.sync()
    dt = time.perf_counter() - t0
    print('%d context(s): %.4f ms per step, %.2f M alpha-solves/s' % (n_ctx, 1e3 * dt / K, 25600 * K / dt / 1e6))
out = [c.fetch(want_v=False, want_H=False) for c in ctxs[:2]]
print('converged', [int(o['converged'].sum()) for o in out], 'audit', [float(np.nanmax(c.audit()['corr'])) for c in ctxs[:2]])
