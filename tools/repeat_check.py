#!/usr/bin/env python3
"""Does a launch repeat bit for bit?  n launches of the BASELINE batch (chain_kernel_mc<32, 2>, 512 workgroups taking pieces from a
queue), of its cut for four in flight (256 workgroups) and of BASELINE config 3 (one workgroup per CU, eight waves), every result
array of every launch against the first launch's.
    python tools/repeat_check.py [n]"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, n_orb, opts in (('cfg4', 16, {}), ('cfg4 cut for four in flight', 16, dict(in_flight=4)), ('cfg3', 4, {}), ('cfg3 binary32 (chain_kernel_lv)', 4, dict(precision=device.PRECISION_F32))):
    batch = bench.build_batch(n_orb, 200, 500, 100, 0)
    ctx = bench.stage(batch, 0)
    ne = len(batch['elems'])
    ctx.upload_chains(np.arange(ne, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(**opts))
    first, bad = None, 0
    for k in range(n):
        ctx.launch(); ctx.finish()
        out = ctx.fetch()
        sig = tuple(zlib.crc32(np.ascontiguousarray(out[f]).tobytes()) for f in ('v', 'H', 'chi2', 'S', 'Q', 'n_iter', 'n_evals', 'converged'))
        if first is None:
            first = sig
        elif sig != first:
            bad += 1
    print('%-34s %s, %d workgroups: %d launches, %d differ from the first' % (name, ctx.last_launch_info()['kernel'], ctx.last_launch_info()['n_workgroups'], n, bad), flush=True)
    ctx.close()
