"""evaluations per (scan, alpha) of the cfg4 batch under several cuts of the scans -- the cost data of tools/schedule_sim.py:
the first alpha of a piece shows what a cold start THERE costs, the others what a warm alpha costs.
    python tools/dump_piece_costs.py gpurun_out/piece_costs.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
out = {}
for split in (0, 1, 4, 10, 15, 20, 25, 33, 50):
    ctx = bench.stage(batch, 0)
    ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(alpha_split=split))
    ctx.launch(); ctx.sync()
    ms = []
    for _ in range(10):
        ctx.launch(); ctx.sync(); ms.append(ctx.last_kernel_ms())
    left = ctx.finish()
    o = ctx.fetch(want_v=False, want_H=False)
    info = ctx.last_launch_info()
    out['evals_split%d' % split] = o['n_evals'].astype(np.int16)
    out['ms_split%d' % split] = np.array([min(ms), left])
    print(split, info['kernel'], info['n_workgroups'], min(ms), left, o['n_evals'].mean(), flush=True)
    ctx.close()
out['kinds'] = np.array(batch['kinds'])
out['alphas'] = np.array(batch['alphas'])
np.savez_compressed(sys.argv[1], **out)
