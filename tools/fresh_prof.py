import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from maxent_amd import synthetic
import maxent_amd as mx
batch = bench.build_batch(16, 200, 500, 100, 0)
def make():
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(batch['tau'], batch['Gmat'])
    ew.omega = batch['omega']
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
    ew.set_error(synthetic.SIGMA)
    return ew
ew = make(); ew.run()      # process warm-up
ew = make()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); ew.run(); dt = time.perf_counter() - t0; pr.disable()
print('fresh object run(): %.1f ms' % (1e3 * dt))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:5000])
