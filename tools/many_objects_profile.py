"""Where the host time of jobs in flight goes: maxent_amd.run_many on four cfg4 objects, wall clock and cProfile.
    python tools/many_objects_profile.py [n_jobs] > gpurun_out/many_objects.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
import maxent_amd as mx                             # noqa: E402
from maxent_amd import synthetic                    # noqa: E402

n_jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = bench.build_batch(16, 200, 500, 100, 0)


def make(k=0):
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(batch['tau'], batch['Gmat'] * (1.0 + 1e-7 * k))
    ew.omega = batch['omega']
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
    ew.set_error(synthetic.SIGMA)
    return ew


res_one = make(0).run()
print(bench.many_objects_block(make, res_one, 16, 100, n_jobs=n_jobs))
jobs = [make(200 + k) for k in range(n_jobs)]
for _ in range(3):
    for ew in jobs:
        ew.maxent_result = None
    out = mx.run_many(jobs)
    del out
# phases of one pass, by hand
for ew in jobs:
    ew.maxent_result = None
t0 = time.perf_counter()
hs = []
marks = []
for ew in jobs:
    hs.append(ew.run_async())
    marks.append(time.perf_counter())
outs = []
for h in hs:
    outs.append(h.result())
    marks.append(time.perf_counter())
print('begin of each job (ms):', ['%.3f' % (1e3 * (b - a)) for a, b in zip([t0] + marks[:n_jobs - 1], marks[:n_jobs])])
print('end of each job (ms):  ', ['%.3f' % (1e3 * (b - a)) for a, b in zip(marks[n_jobs - 1:-1], marks[n_jobs:])])
print('kernel ms per job:', [ew.last_launches[-1]['kernel_ms'] for ew in jobs])
del outs, hs
pr = cProfile.Profile()
for rep in range(20):
    for ew in jobs:
        ew.maxent_result = None
    pr.enable()
    out = mx.run_many(jobs)
    pr.disable()
    del out
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue())
