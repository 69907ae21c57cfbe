"""wall time of the whole user-level job: ElementwiseMaxEnt on the cfg4 input (16x16 elements, 100 alpha)
-- a fresh object (kernel fill, SVD, contexts, two launches, records, analyzers) and the same object run
again (decomposition and contexts kept) -- with a cProfile summary of the host side of the warm run"""
import os, sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import maxent_amd as mx
n_orb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = bench.build_batch(n_orb, 200, 500, 100, 0)
def make(k=0):
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(batch['tau'], batch['Gmat'] * (1.0 + 1e-7 * k))      # (k: other data on the same grids)
    ew.omega = batch['omega']
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
    ew.set_error(1e-4)
    return ew
t0 = time.perf_counter(); ew = make(); res = ew.run(); t1 = time.perf_counter()
print('first job of the process (library load, device init): %.3f s' % (t1 - t0))
from maxent_amd.batch_solver import BatchSolver
for label, pool in (('contexts of its own', 0), ('contexts taken over from the object before', BatchSolver.POOL_SIZE)):
    keep, BatchSolver.POOL_SIZE = BatchSolver.POOL_SIZE, pool
    cold = []
    for k in range(4):
        res = None                      # (the result before is dropped, as a loop over data sets would)
        t0 = time.perf_counter(); ew = make(k + 1); tm = time.perf_counter(); res = ew.run(); t1 = time.perf_counter()
        cold.append((tm - t0, t1 - tm))
    BatchSolver.POOL_SIZE = keep
    print('fresh object on new data, %s: set-up %.1f ms + run() %.1f ms (best of 3 after the first; run() includes the SVD of the kernel and the staging)'
          % (label, 1e3 * min(c[0] for c in cold[1:]), 1e3 * min(c[1] for c in cold[1:])))
warm = []
for _ in range(5):
    ew.maxent_result = res = None       # (a result that is still held claims its H: it would be fetched first)
    t0 = time.perf_counter(); res = ew.run(); warm.append(time.perf_counter() - t0)
print('same object again: run() %.1f ms (best of 5) for %d elements x 100 alpha = %.0f alpha-solves/s as the caller sees them; kernel launches %s ms'
      % (1e3 * min(warm), n_orb * n_orb, n_orb * n_orb * 100 / min(warm), [round(l['kernel_ms'], 2) for l in ew.last_launches[-1:]]))
t0 = time.perf_counter(); A = np.asarray(res.A); t1 = time.perf_counter()
print('first look at result.A (fetches all H: %.0f MB): %.1f ms; A_out %s' % (A.nbytes / 1e6, 1e3 * (t1 - t0), res.A_out.shape))
ew.maxent_result = None
pr = cProfile.Profile(); pr.enable(); res = ew.run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:4200])
