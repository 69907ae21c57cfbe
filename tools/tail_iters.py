"""evaluations per alpha along the scans of the BASELINE batch, by position in the alpha range, in the one-chain layout
(binary64 Gram matrix) and in the lock-step layout (binary16 Gram products): is the tail of a scan expensive because of the
path or because of the inexact Newton matrix?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
diag = [e for e, (i, j) in enumerate(batch['elems']) if i == j]
off = [e for e, (i, j) in enumerate(batch['elems']) if i != j][:32]
for name, sel in (('normal (16 diagonal scans)', diag), ('plusminus (32 scans)', off)):
    ctx = bench.stage(batch, 0, sel)
    for cpw in (1, 4):
        ctx.upload_chains(np.arange(len(sel), dtype=np.int32), batch['alphas'], batch['v0'][sel], device.default_opts(chains_per_wg=cpw, alpha_split=1))
        ctx.launch(); out = ctx.fetch(want_v=False, want_H=False)
        ev = out['n_evals'].astype(float)
        print(name, ctx.last_launch_info()['kernel'], 'evals per alpha: all %.2f | alpha index 10-89: %.2f | 90-93: %.2f | 94-99: %.2f | max single %d' % (
            ev[:, 1:].mean(), ev[:, 10:90].mean(), ev[:, 90:94].mean(), ev[:, 94:].mean(), ev[:, 1:].max()))
    ctx.close()
