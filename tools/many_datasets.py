#!/usr/bin/env python3
"""The BASELINE batch with a data set PER ELEMENT (per-element error bars, the input mode of ElementwiseMaxEnt.set_error with an
array per element or set_cov per element): every data set has a basis of its own on the device (V, V^T: 2 x 224 KB), so the
workgroups no longer stream ONE basis out of the L2.  Kernel time against the number of distinct data sets.
    python tools/many_datasets.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
K = batch['K']
n = len(batch['elems'])
for n_ds in (1, 2, 4, 16, 64, 256):
    ctx = device.DeviceContext(K.U, K.S, K.V, device=0)
    ds = [ctx.add_dataset(batch['err'] * (1.0 + 1e-3 * k)) for k in range(n_ds)]
    ctx.set_elements([ds[e % n_ds] for e in range(n)], [batch['Gmat'][batch['elems'][e]] for e in range(n)],
                     np.tile(batch['D'], (n, 1)), batch['kinds'])
    ctx.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'])
    ts = []
    for _ in range(6):
        ctx.launch(); ctx.sync(); ts.append(ctx.last_kernel_ms())
    left = ctx.finish()
    out = ctx.fetch(want_v=False, want_H=False)
    au = ctx.audit()['corr']
    info = ctx.last_launch_info()
    print('%3d data sets: kernel %.3f ms (min of 6; max %.3f) = %.1f M alpha-solves/s, %s, %d workgroups, converged %d / %d, left %d, audit max %.1e' % (
        n_ds, min(ts), max(ts), 25600 / min(ts) / 1e3, info['kernel'], info['n_workgroups'], int(out['converged'].sum()), out['converged'].size, left, np.nanmax(au)), flush=True)
    ctx.close()
