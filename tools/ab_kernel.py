#!/usr/bin/env python3
"""A/B of library builds on ONE box: kernel time of the cfg4 batch (HIP events over launches back to back), each build in a
process of its own, alternating.   python tools/ab_kernel.py maxent_amd/lib/libmaxent_hip.so maxent_amd/lib/libmaxent_hip_x.so [--reps 3]
(a build that lacks newer entry points of include/maxent_hip.h is loaded without them)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import ctypes
    import numpy as np
    from maxent_amd import device
    lib = ctypes.CDLL(os.environ['MAXENT_AMD_LIB'])
    device.SYMBOLS[:] = [s for s in device.SYMBOLS if hasattr(lib, s[0])]
    import bench
    batch = bench.build_batch(16, 200, 500, 100, 0)
    ctx = bench.stage(batch, 0)
    ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
    for _ in range(20):
        ctx.launch()
    ctx.sync()
    out = []
    for _ in range(3):
        ctx.timing_mark()
        for _ in range(200):
            ctx.launch()
        out.append(ctx.ms_since_mark() / 200)
    print('%.4f %.4f %.4f  %s' % (out[0], out[1], out[2], ctx.last_launch_info()['kernel']))
    sys.exit(0)
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 3
for r in range(reps):
    for l in libs:
        env = dict(os.environ, MAXENT_AMD_LIB=os.path.abspath(l))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
        print('%-44s %s' % (os.path.basename(l), p.stdout.strip() or p.stderr.strip()[-300:]), flush=True)
