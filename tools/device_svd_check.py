"""Device kernel staging (mxe_kernel_svd) against the host numpy path: K fill, preblur product,
truncated SVD; timing of a b-scan.  python tools/device_svd_check.py [n_tau n_omega]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import maxent_amd as mx                           # noqa: E402
from maxent_amd import device, synthetic          # noqa: E402

n_tau, n_w = (int(x) for x in sys.argv[1:3]) if len(sys.argv) >= 3 else (200, 500)
tau, omega = synthetic.grids(n_tau, n_w)
K = mx.TauKernel(tau=tau, omega=omega, beta=synthetic.BETA)
bs = [0.0, 0.05, 0.1, 0.2]
device.kernel_svd(tau, np.asarray(omega), omega.delta, synthetic.BETA, [0.0])      # warm-up (module load)
res = device.kernel_svd(tau, np.asarray(omega), omega.delta, synthetic.BETA, bs, want_K=True)
for b, r in zip(bs, res):
    Kh = np.array(K.K) if b <= 0 else np.array(mx.PreblurKernel(K=K, b=b).K)
    t0 = time.perf_counter()
    Ul, Sl, Vhl = np.linalg.svd(Kh, full_matrices=False)
    t_host = time.perf_counter() - t0
    ns_l = int((Sl >= 1e-14).sum())
    U, S, V = r['U'], r['S'], r['V']
    ns = len(S)
    k = min(ns, ns_l)
    print('b=%.2f  n_s device %d / lapack %d  qr_rank %d  sweeps %d | max|K_dev-K_host| %.2e  max|dS| %.2e  '
          'recon %.2e  orthU %.2e  orthV %.2e | host np.linalg.svd %.1f ms' %
          (b, ns, ns_l, r['qr_rank'], r['sweeps'], np.abs(r['K'] - Kh).max(), np.abs(S[:k] - Sl[:k]).max(),
           np.abs((U * S) @ V.T - Kh).max(), np.abs(U.T @ U - np.eye(ns)).max(),
           np.abs(V.T @ V - np.eye(ns)).max(), 1e3 * t_host))
print('device time of the 4-item batch: %.2f ms' % res[0]['ms'])
for nb in (1, 8, 32):
    t0 = time.perf_counter()
    r = device.kernel_svd(tau, np.asarray(omega), omega.delta, synthetic.BETA, np.linspace(0.02, 0.3, nb))
    print('b-scan of %2d widths: device %.2f ms, wall incl. alloc + copies %.1f ms' %
          (nb, r[0]['ms'], 1e3 * (time.perf_counter() - t0)))
