"""run_many with ONLY its own contexts in the process (streams share hardware queues): 4 jobs, and 8 jobs with 4 in flight"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import maxent_amd as mx
from maxent_amd import synthetic
batch = bench.build_batch(16, 200, 500, 100, 0)
def make(k=0):
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(batch['tau'], batch['Gmat'] * (1.0 + 1e-7 * k))
    ew.omega = batch['omega']
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
    ew.set_error(synthetic.SIGMA)
    return ew
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
jobs = [make(k) for k in range(n)]
out = mx.run_many(jobs, in_flight=fl); del out
ts = []
for rep in range(12):
    for ew in jobs: ew.maxent_result = None
    t0 = time.perf_counter(); out = mx.run_many(jobs, in_flight=fl); ts.append(time.perf_counter() - t0); del out
from maxent_amd.batch_solver import BatchSolver
print('jobs %d in_flight %d: run_many best %.3f ms median %.3f ms -> %.2f M alpha-solves/s; contexts in the pool: %d; queues %s' % (
    n, fl, 1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2], 25600 * n / min(ts) / 1e6, len(BatchSolver._pooled), os.environ.get('GPU_MAX_HW_QUEUES')))
