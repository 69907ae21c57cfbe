import os, sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
import maxent_amd as mx
batch = bench.build_batch(16, 200, 500, 100, 0)
ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
ew.set_verbosity(mx.VerbosityFlags.Quiet)
ew.set_G_tau_data(batch['tau'], batch['Gmat'])
ew.omega = batch['omega']
ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
ew.set_error(1e-4)
res = ew.run(); res = None; ew.maxent_result = None
res = ew.run(); res = None; ew.maxent_result = None
ts = []
for _ in range(5):
    ew.maxent_result = res = None
    t0 = time.perf_counter(); res = ew.run(); ts.append(time.perf_counter() - t0)
print('warm run ms', [round(1e3 * t, 2) for t in ts])
ew.maxent_result = res = None
pr = cProfile.Profile(); pr.enable(); res = ew.run(); pr.disable()
st = pstats.Stats(pr).stats          # (pstats prints milliseconds as 0.000: microseconds here)
rows = sorted(((tt, ct, nc, '%s:%d(%s)' % (os.path.basename(f), l, n)) for (f, l, n), (cc, nc, tt, ct, _) in st.items()), reverse=True)
print('profiled run: %.0f us in total (profiler overhead included)' % (1e6 * sum(r[0] for r in rows)))
print('%9s %9s %7s  function' % ('self us', 'cum us', 'calls'))
for tt, ct, nc, name in rows[:int(os.environ.get('E2E_PROF_ROWS', '60'))]:
    print('%9.0f %9.0f %7d  %s' % (1e6 * tt, 1e6 * ct, nc, name))
