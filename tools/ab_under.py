#!/usr/bin/env python3
"""like tools/ab_kernel.py for the launches that do not fill the GPU: cfg2, cfg3 and the slowest shard of cfg4 / 8 per build"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import ctypes
    import numpy as np
    from maxent_amd import device, synthetic
    lib = ctypes.CDLL(os.environ['MAXENT_AMD_LIB'])
    device.SYMBOLS[:] = [s for s in device.SYMBOLS if hasattr(lib, s[0])]
    import bench
    out = []
    for name in ('cfg2', 'cfg3', 'shard8', 'cfg4'):
        if name == 'cfg2':
            batch = bench.build_batch(2, 200, 500, 100, 0)
            _, _, _, G1 = synthetic.single_G(200, 500)
            batch['Gmat'] = G1[None, None, :]
            batch['elems'], batch['kinds'], batch['v0'] = [(0, 0)], batch['kinds'][:1], batch['v0'][:1]
            which = [0]
        elif name == 'cfg3':
            batch = bench.build_batch(4, 200, 500, 100, 0); which = list(range(16))
        else:
            batch = bench.build_batch(16, 200, 500, 100, 0)
            which = [e for e in range(256) if e % 8 == 7] if name == 'shard8' else list(range(256))
        ctx = bench.stage(batch, 0, which)
        ctx.upload_chains(np.arange(len(which), dtype=np.int32), batch['alphas'], batch['v0'][which])
        for _ in range(10):
            ctx.launch()
        ctx.sync()
        ctx.timing_mark()
        for _ in range(100):
            ctx.launch()
        ms = ctx.ms_since_mark() / 100
        ctx.launch()
        res = ctx.fetch(want_v=False, want_H=False)
        aud = ctx.audit()['corr']
        out.append('%s %.4f (evals %d, audit %.1e)' % (name, ms, int(res['n_evals'].sum()), float(np.nanmax(aud))))
        ctx.close()
    if os.environ.get('AB_ALL_SHARDS'):
        # every rank's shard of cfg4 / 8 and cfg4 / 4: a job ends with its slowest rank (runaway tails show up here)
        batch = bench.build_batch(16, 200, 500, 100, 0)
        for N in (8, 4, 2):
            ts, worst, lefts = [], 0, []
            for r in range(N):
                which = [e for e in range(256) if e % N == r]
                ctx = bench.stage(batch, 0, which)
                ctx.upload_chains(np.arange(len(which), dtype=np.int32), batch['alphas'], batch['v0'][which])
                for _ in range(5):
                    ctx.launch()
                ctx.sync()
                ctx.timing_mark()
                for _ in range(40):
                    ctx.launch()
                ts.append(ctx.ms_since_mark() / 40)
                ctx.launch()
                left = ctx.finish()
                res = ctx.fetch(want_v=False, want_H=False)
                worst = max(worst, int(res['n_evals'].max()))
                lefts.append(left)
                assert res['converged'].all(), (N, r, left)
                ctx.close()
            out.append('N=%d slowest %.4f (all %s; most evals for one alpha %d; left to the finishing pass %s)' %
                       (N, max(ts), ' '.join('%.3f' % t for t in ts), worst, lefts))
    print('  '.join(out))
    sys.exit(0)
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 2
for r in range(reps):
    for l in libs:
        env = dict(os.environ, MAXENT_AMD_LIB=os.path.abspath(l))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
        print('%-34s %s' % (os.path.basename(l), p.stdout.strip() or p.stderr.strip()[-300:]), flush=True)
