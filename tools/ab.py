#!/usr/bin/env python3
"""A/B of library builds on ONE box (boxes differ by a few per cent): kernel time of the bench workload, alternating.
    python tools/ab.py libA.so libB.so [...]      (paths relative to maxent_amd/lib)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
extra = os.environ.get('AB_ARGS', '').split()
res = {l: [] for l in libs}
for rep in range(int(os.environ.get('AB_REPS', '3'))):
    for l in libs:
        env = dict(os.environ, MAXENT_AMD_LIB=os.path.join(ROOT, 'maxent_amd', 'lib', l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-extras', '--steps', '300'] + extra,
                             env=env, capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[l].append((d['roofline']['kernel_ms'], d['ms_per_step']))
        print('%-28s rep %d: kernel %.4f ms, step %.4f ms, %s, evals/solve %.3f' % (l, rep, d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['evals_per_solve']), flush=True)
for l in libs:
    k = sorted(x[0] for x in res[l])
    print('%-28s kernel ms: min %.4f median %.4f' % (l, k[0], k[len(k) // 2]))
