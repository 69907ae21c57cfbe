#!/usr/bin/env python3
"""More than 64 singular values above the reference's threshold (many data points: 1 000 imaginary times give 79): the device's
lock-step kernels hold 64 directions, the one-chain kernel with its 128 x 128 Newton matrix is 15-80 x slower.  The host layer
keeps 64 directions when the others cannot be told from zero in the job (maxent_amd.batch_solver.directions_to_keep);
MAXENT_AMD_ALL_DIRECTIONS=1 switches that off.  run() both ways, time and difference.
    python tools/many_data_points.py [n_orb] [n_tau]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maxent_amd as mx
from maxent_amd import synthetic
n_orb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_tau = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, 500)


def run():
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(tau, Gmat)
    ew.omega = omega
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
    ew.set_error(synthetic.SIGMA)
    ew.run()
    t0 = time.perf_counter()
    ew.maxent_result = None
    r = ew.run()
    dt = time.perf_counter() - t0
    return np.array(r.A), np.array(r.chi2), np.array(r.v) if hasattr(r, 'v') else None, dt, ew.last_launches[-1]


out = {}
for name, env in (('64 directions kept', None), ('all directions', '1')):
    if env:
        os.environ['MAXENT_AMD_ALL_DIRECTIONS'] = env
    out[name] = run()
    A, chi2, v, dt, info = out[name]
    print('%-20s run() %.2f ms, kernel %s %.3f ms, converged all: %s' % (name, 1e3 * dt, info['kernel'], info['kernel_ms'], bool(np.all(np.isfinite(A)))))
a, b = out['64 directions kept'], out['all directions']
d = np.linalg.norm(a[0] - b[0], axis=-1) / np.linalg.norm(b[0], axis=-1)
print('n_orb %d n_tau %d: A differs by at most %.2e (relative L2 per alpha), chi2 by %.2e; v shapes %s %s, largest |v| in the directions not kept (all-directions run): %.2e' % (
    n_orb, n_tau, d.max(), np.max(np.abs(a[1] - b[1]) / np.abs(b[1])), a[2].shape if a[2] is not None else None, b[2].shape if b[2] is not None else None,
    np.max(np.abs(b[2][..., 64:])) if b[2] is not None and b[2].shape[-1] > 64 else float('nan')))
