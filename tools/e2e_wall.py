"""wall time of ElementwiseMaxEnt.run() on the cfg4 input by stage: timers wrapped around the calls of the host path (the
wrappers themselves cost ~2 ms of the total)"""
import os, sys, time, functools, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import maxent_amd as mx
from maxent_amd import device, batch_solver, maxent_loop, maxent_result, elementwise_maxent, analyzers
acc = collections.OrderedDict()
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or (getattr(obj, '__name__', str(obj)) + '.' + name)
    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)
for n in ('set_elements', 'upload_chains', 'launch', 'fetch', 'fetch_rows', 'add_dataset', 'clear_datasets', 'logdet', 'sync'):
    if hasattr(device.DeviceContext, n): wrap(device.DeviceContext, n, 'ctx.' + n)
wrap(batch_solver.BatchSolver, 'solve'); wrap(batch_solver.BatchSolver, '_stage'); wrap(batch_solver.BatchSolver, 'rows'); wrap(batch_solver.BatchSolver, 'materialize_pending')
wrap(maxent_result.MaxEntResult, 'analyze_batch'); wrap(maxent_result.MaxEntResult, 'add_element_results')
wrap(maxent_loop.MaxEntLoop, 'make_record'); wrap(maxent_loop.MaxEntLoop, 'make_spec')
wrap(elementwise_maxent.ElementwiseMaxEnt, '_run_batch'); wrap(elementwise_maxent.ElementwiseMaxEnt, '_load_element'); wrap(elementwise_maxent.ElementwiseMaxEnt, 'prepare_maxent_result')
for n in ('_prepare_batch', '_solve_batches', '_finish_batch', '_offdiag_jobs'):
    if hasattr(elementwise_maxent.ElementwiseMaxEnt, n): wrap(elementwise_maxent.ElementwiseMaxEnt, n)
for n in ('make_records', 'spec_like', 'note_minimizer_state'):
    if hasattr(maxent_loop.MaxEntLoop, n) and not isinstance(maxent_loop.MaxEntLoop.__dict__.get(n), staticmethod): wrap(maxent_loop.MaxEntLoop, n)
wrap(maxent_result.MaxEntResult, 'add_batch_results')
for n in ('finish', 'select3_launch', 'select3_fetch'):
    if hasattr(device.DeviceContext, n): wrap(device.DeviceContext, n, 'ctx.' + n)
for cls in (analyzers.LineFitAnalyzer, analyzers.Chi2CurvatureAnalyzer, analyzers.EntropyAnalyzer):
    if hasattr(cls, 'pick_many'): wrap(cls, 'pick_many', cls.__name__ + '.pick_many')
for cls in (analyzers.LineFitAnalyzer, analyzers.Chi2CurvatureAnalyzer, analyzers.EntropyAnalyzer):
    if hasattr(cls, 'analyze_many'): wrap(cls, 'analyze_many', cls.__name__ + '.analyze_many')
batch = bench.build_batch(16, 200, 500, 100, 0)
ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
ew.set_verbosity(mx.VerbosityFlags.Quiet)
ew.set_G_tau_data(batch['tau'], batch['Gmat'])
ew.omega = batch['omega']
ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
ew.set_error(1e-4)
for _ in range(3):
    ew.maxent_result = res = None; res = ew.run()
acc.clear()
N = 5
tt = 0
for _ in range(N):
    ew.maxent_result = res = None
    t0 = time.perf_counter(); res = ew.run(); tt += time.perf_counter() - t0
print('run() %.2f ms' % (1e3 * tt / N))
for k, v in acc.items():
    print('  %-45s %.2f ms' % (k, 1e3 * v / N))
