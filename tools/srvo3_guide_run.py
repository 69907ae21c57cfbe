"""The reference's SrVO3 guide (doc/guide/srvo3/srvo3_run.py) on the device: G(tau) with error column (the data of
tests/golden/kat_srvo3.npz), 400 omega points, 50 alpha from 1e-4 to 100, probability, then the same with preblur
b = 0.1; prints what the guide prints (the alpha index the analyzers pick) and the time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maxent_amd as mx
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'kat_srvo3.npz'))
tm = mx.TauMaxEnt(probability='normal')
tm.set_verbosity(mx.VerbosityFlags.Quiet)
tm.set_G_tau_data(g['tau'], g['G'])
tm.set_error(g['err'])
tm.omega = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=400)
tm.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-4, alpha_max=100, n_points=50)
t0 = time.perf_counter(); res = tm.run(); t1 = time.perf_counter()
print('run: %.3f s, n_s %d, converged %d / %d, iterations per alpha %s' % (t1 - t0, len(tm.K.S), int(np.sum(res.converged)), res.converged.size,
                                                                        list(np.asarray(res.n_iter).astype(int))))
for name in ('Chi2CurvatureAnalyzer', 'LineFitAnalyzer', 'ClassicAnalyzer'):
    print(name, res.analyzer_results[name]['alpha_index'])
b = 0.1
tm.A_of_H = mx.PreblurA_of_H(b=b, omega=tm.omega)
K_orig = tm.K
tm.K = mx.PreblurKernel(K=K_orig, b=b)
t0 = time.perf_counter(); res_pb = tm.run(); t1 = time.perf_counter()
print('preblur run: %.3f s, converged %d / %d' % (t1 - t0, int(np.sum(res_pb.converged)), res_pb.converged.size))
print('Preblur LineFit:', res_pb.analyzer_results['LineFitAnalyzer']['alpha_index'])
w = np.asarray(tm.omega)
print('norm of A_out (LineFit, Bryan, preblur LineFit): %.4f %.4f %.4f' % (
    np.trapezoid(res.analyzer_results['LineFitAnalyzer']['A_out'], w), np.trapezoid(res.analyzer_results['BryanAnalyzer']['A_out'], w),
    np.trapezoid(res_pb.analyzer_results['LineFitAnalyzer']['A_out'], w)))
