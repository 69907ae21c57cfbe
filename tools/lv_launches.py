#!/usr/bin/env python3
"""launches of mxe::chain_kernel_lv for a counter run (tools/round_profile.sh): BASELINE config 2 and 3 in precision = F32, n launches each"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for n_orb in (1, 4):
    batch = bench.build_batch(n_orb, 200, 500, 100, 0)
    ctx = bench.stage(batch, 0)
    ne = len(batch['elems'])
    ctx.upload_chains(np.arange(ne, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(precision=device.PRECISION_F32))
    for _ in range(n):
        ctx.launch()
    ctx.sync()
    print(n_orb, ctx.last_launch_info()['kernel'], ctx.last_kernel_ms())
    ctx.close()
