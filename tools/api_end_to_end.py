"""wall time of the whole user-level job: ElementwiseMaxEnt on the cfg4 input (16x16 elements, 100 alpha),
with a cProfile summary of the host side"""
import os, sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import maxent_amd as mx
n_orb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = bench.build_batch(n_orb, 200, 500, 100, 0)
def job():
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(batch['tau'], batch['Gmat'])
    ew.omega = batch['omega']
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
    ew.set_error(1e-4)
    return ew, ew.run()
t0 = time.perf_counter(); ew, res = job(); t1 = time.perf_counter()
print('first run (includes library load): %.2f s' % (t1 - t0))
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); ew, res = job(); t1 = time.perf_counter()
pr.disable()
print('second run: %.2f s for %d elements x 100 alpha; kernel launches: %s' % (
    t1 - t0, n_orb * n_orb, [round(l['kernel_ms'], 2) for l in ew.last_launches]))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
