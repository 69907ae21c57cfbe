"""BASELINE config 5: off-diagonal PlusMinusEntropy + preblur on an 8x8 matrix G(tau),
fp32 vs fp64 tolerance sweep.

Runs the same ElementwiseMaxEnt job (n_tau = 200, n_omega = 500, 100 alpha, preblur b = 0.1 on the
off-diagonal worker, hermiticity on: 8 diagonal + 28 off-diagonal alpha scans) twice on the
device -- binary64 and the binary32 streaming variant (LevenbergMinimizer(precision='f32')) -- and
reports, per alpha, the relative L2 distance of A(omega) between the two, the largest tolerance
class (1e-3 / 1e-4 / 1e-5 / 1e-6) every element passes, iteration counts and kernel times.

    python tools/cfg5_tolerance_sweep.py [n_tau n_omega n_alpha] > profiles/r01_f_cfg5_fp32_sweep.txt
"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__file__), '..'))
import maxent_amd as mx                     # noqa: E402
from maxent_amd import synthetic            # noqa: E402

CLASSES = (1e-6, 1e-5, 1e-4, 1e-3, 1e-2)


def run(precision, tau, omega, K, Gmat, n_alpha, b):
    ew = mx.ElementwiseMaxEnt(use_hermiticity=True,
                              minimizer=mx.LevenbergMinimizer(precision=precision))
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(tau, Gmat)
    ew.omega = omega
    ew.alpha_mesh = synthetic.alpha_mesh(n_alpha)
    ew.set_error(synthetic.SIGMA)
    off = ew.maxent_offdiagonal
    off.A_of_H = mx.PreblurA_of_H(b=b, omega=off.omega)
    off.K = mx.PreblurKernel(K=off.K, b=b)
    t0 = time.perf_counter()
    res = ew.run()
    cold = time.perf_counter() - t0
    warm = []
    for _ in range(3):                      # the same object again: kernel decompositions and contexts kept
        del res
        t0 = time.perf_counter()
        res = ew.run()
        warm.append(time.perf_counter() - t0)
    return res, (cold, min(warm))


def klass(e):
    for c in CLASSES:
        if e <= c:
            return c
    return float('inf')


def main():
    n_tau, n_w, n_alpha = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (200, 500, 100)
    b = 0.1
    tau, omega, K, Gmat, _ = synthetic.matrix_G(8, n_tau, n_w)
    r64, t64 = run('f64', tau, omega, K, Gmat, n_alpha, b)
    r32, t32 = run('f32', tau, omega, K, Gmat, n_alpha, b)
    iu = np.triu_indices(8)
    A64, A32 = r64.A[iu], r32.A[iu]                       # [36][n_alpha][n_w]
    err = np.linalg.norm(A32 - A64, axis=-1) / np.linalg.norm(A64, axis=-1)
    diag = iu[0] == iu[1]
    print('# cfg5: 8x8 G(tau), n_tau=%d n_omega=%d n_alpha=%d, off-diagonals PlusMinusEntropy + preblur b=%.2f' %
          (n_tau, n_w, n_alpha, b))
    print('# rel. L2 of A(omega): binary32 streaming variant vs binary64, max over elements, per alpha')
    print('# wall (host + device, whole ElementwiseMaxEnt.run), first run of a fresh object (SVD of K and of the '
          'preblurred K): f64 %.3f s, f32 %.3f s; the same object again (best of 3): f64 %.3f s, f32 %.3f s' %
          (t64[0], t32[0], t64[1], t32[1]))
    for name, r in (('f64', r64), ('f32', r32)):
        it = r.n_iter[iu]
        print('# %s: converged %d / %d, Newton iterations per alpha-solve %.2f (diag %.2f, offdiag %.2f)' %
              (name, int(np.nansum(r.converged[iu])), it.size, it.mean(), it[diag].mean(), it[~diag].mean()))
    print('# alpha_scaled  max_err_diag(normal)  max_err_offdiag(plusminus+preblur)  class')
    alpha = np.asarray(r64.alpha).reshape(-1)[:n_alpha]
    for ia in range(n_alpha):
        ed, eo = err[diag, ia].max(), err[~diag, ia].max()
        print('%12.5e  %10.3e  %10.3e  %g' % (alpha[ia], ed, eo, klass(max(ed, eo))))
    print('# overall: diag max %.3e (class %g), offdiag max %.3e (class %g)' %
          (err[diag].max(), klass(err[diag].max()), err[~diag].max(), klass(err[~diag].max())))
    for c in CLASSES:
        print('# alpha-solves within %g: %.1f %%' % (c, 100.0 * np.mean(err <= c)))
    chi = np.abs(r32.chi2[iu] / r64.chi2[iu] - 1).max()
    print('# max rel. difference of chi2: %.3e' % chi)


if __name__ == '__main__':
    main()
