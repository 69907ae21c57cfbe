#!/usr/bin/env python3
"""A/B of SCHEDULE variants of the cfg4 launch inside one process on one box: every variant is an environment setting read by
mxe_chains_upload (MXE_TAPER ...) and / or mxe_opts fields; kernel time by HIP events over launches back to back, the variants
taken in turn for several repetitions; evaluations per alpha, converged flags and the device audit of every variant.
    python tools/ab_env.py 'name:ENV=val,ENV2=val;opt=val' ...        (name 'base': nothing set)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from maxent_amd import device

variants = []
for spec in sys.argv[1:] or ['base:']:
    name, _, rest = spec.partition(':')
    envs, _, optstr = rest.partition(';')
    env = dict(kv.split('=') for kv in envs.split(',') if kv)
    opts = {k: (float(v) if ('.' in v or 'e' in v) else int(v)) for k, v in (kv.split('=') for kv in optstr.split(',') if kv)}
    variants.append((name, env, opts))
batch = bench.build_batch(16, 200, 500, 100, 0)
n_launch = int(os.environ.get('AB_LAUNCHES', '40'))
ctxs = {}
all_env = sorted(set(k for _, env, _ in variants for k in env))
for name, env, opts in variants:
    for k in all_env:
        os.environ.pop(k, None)
    os.environ.update(env)
    c = bench.stage(batch, 0)
    c.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(**opts))
    c.launch(); c.sync()
    left = c.finish()
    o = c.fetch(want_v=False, want_H=False)
    a = c.audit()['corr']
    info = c.last_launch_info()
    ctxs[name] = c
    print('%-22s %s wg %d: evals/alpha %.3f, converged %d / %d, left %d, audit max %.2e p99 %.2e' % (
        name, info['kernel'], info['n_workgroups'], o['n_evals'].mean(), int(o['converged'].sum()), o['converged'].size, left,
        np.nanmax(a), np.nanpercentile(a, 99)), flush=True)
for k in all_env:
    os.environ.pop(k, None)
res = {name: [] for name, _, _ in variants}
for rep in range(int(os.environ.get('AB_REPS', '5'))):
    for name, _, _ in variants:
        c = ctxs[name]
        for _ in range(5):
            c.launch()
        c.sync()
        c.timing_mark()
        for _ in range(n_launch):
            c.launch()
        res[name].append(c.ms_since_mark() / n_launch)
for name, _, _ in variants:
    k = sorted(res[name])
    print('%-22s kernel ms: min %.4f median %.4f max %.4f' % (name, k[0], k[len(k) // 2], k[-1]), flush=True)
