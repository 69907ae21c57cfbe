"""Kernel time of the one-chain-per-workgroup layout in binary64 and in the binary32 streaming
variant (mxe_opts.precision) on the cfg4 batch (256 alpha scans x 100 alpha), next to the default
lock-step binary64 kernel.  python tools/fp32_kernel_time.py [n_orb]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                    # noqa: E402
from maxent_amd import device                   # noqa: E402

n_orb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = bench.build_batch(n_orb, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
n_chain = len(batch['elems'])
ref = None
for label, kw in (('lock-step f64 (default)', dict()),
                  ('one chain per workgroup f64', dict(chains_per_wg=1)),
                  ('one chain per workgroup f32', dict(chains_per_wg=1, precision=device.PRECISION_F32))):
    ctx.upload_chains(np.arange(n_chain), batch['alphas'], batch['v0'], device.default_opts(**kw))
    ms = []
    for _ in range(6):
        ctx.launch()
        ctx.sync()
        ms.append(ctx.last_kernel_ms())
    out = ctx.fetch()
    if ref is None:
        ref = out['H']
    e = np.linalg.norm(out['H'] - ref, axis=-1) / np.linalg.norm(ref, axis=-1)
    info = ctx.last_launch_info()
    print('%-30s kernel %.3f ms (min of 5), %d waves/chain, %d workgroups, LDS %d B, iterations/alpha %.2f, '
          'converged %d / %d, max rel L2 of H vs default %.2e' %
          (label, min(ms[1:]), info['waves_per_chain'], info['n_workgroups'], info['lds_bytes'],
           out['n_iter'].mean(), int(out['converged'].sum()), out['converged'].size, e.max()))
