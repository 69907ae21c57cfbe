"""throughput of back-to-back cfg4 batches: one context (one stream) against two or three contexts taking the batches in turn
(their kernels overlap where one's workgroups have finished and the other's can start)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
n = len(batch['elems'])
wgpc = int(sys.argv[4]) if len(sys.argv) > 4 else 0             # mxe_opts.wg_per_cu
hint = int(sys.argv[3]) if len(sys.argv) > 3 else 0             # mxe_opts.in_flight
split = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # mxe_opts.alpha_split: pieces per scan (0: the library's choice)
def make():
    c = bench.stage(batch, 0)
    c.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(alpha_split=split, in_flight=hint, wg_per_cu=wgpc))
    return c
n_max = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctxs = [make() for _ in range(n_max)]
for c in ctxs:
    for _ in range(3):
        c.launch(); c.select_launch(0)
    c.sync()
for n_ctx in ([1, 2, 3, 4, 2] if n_max == 4 else [c for c in (2, 3, 4, 6, 8, 12, 16) if c <= n_max]):
    use = ctxs[:n_ctx]
    K = 600
    for c in use: c.sync()
    t0 = time.perf_counter()
    for k in range(K):
        c = use[k % n_ctx]
        c.launch(); c.select_launch(0)
    for c in use: c.sync()
    dt = time.perf_counter() - t0
    print('alpha_split %d, in_flight hint %d (%s, %d workgroups), %d context(s): %.4f ms per step, %.2f M alpha-solves/s' % (split, hint, ctxs[0].last_launch_info()['kernel'], ctxs[0].last_launch_info()['n_workgroups'], n_ctx, 1e3 * dt / K, 25600 * K / dt / 1e6))
out = [c.fetch(want_v=False, want_H=False) for c in ctxs[:2]]
print('converged', [int(o['converged'].sum()) for o in out], 'audit', [float(np.nanmax(c.audit()['corr'])) for c in ctxs[:2]])
