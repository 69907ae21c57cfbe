#!/usr/bin/env python3
"""VERDICT r03 item 5 (i), measured on the numpy model of the kernel's iteration (oracle/sform.py: solve_alpha -- test
infrastructure, this script is a measurement aid, not product code): "chord rounds" -- the second and later Newton
iterations of a warm alpha reuse the Gram matrix (and so the eliminated system) of its FIRST iteration: no Gram tiles, no
operand split, no MFMA burst in those rounds.  What it costs in rounds: evaluations per alpha over a warm scan of the BASELINE
grids (cfg2's normal-entropy scan and an off-diagonal, plus-minus scan of cfg3), tol_h = 1e-9, for
  newton   : every iteration with its own Gram matrix (the kernel today), stopping estimate on
  chord    : iterations >= 2 of an alpha with the matrix of its first; the estimate of the next correction is then
             (kappa + theta) r with kappa = the relative change of the weights since the matrix was formed (sum of |du|).
A round of the lock-step kernel costs the same whether its solve is a chord step or not unless ALL four slots of the
workgroup skip the Gram part, so rounds are what count."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sform as SF
from maxent_amd import synthetic, hostprep, device
import bench


def chain(basis, el, alphas, v0, chord, tol=1e-9, theta=1e-5):
    c = basis.c
    sumD = el.D.sum() * (1.0 if el.entropy == 'normal' else 2.0)
    v = np.array(v0, float)
    ev = SF.evaluate(basis, el, alphas[0], v)
    evals = []
    for a in alphas:
        n = 0
        W0 = None
        drift = 0.0
        for it in range(60):
            rhs = ev['rho'] + a * v / c
            if chord and W0 is not None:
                W = W0
            else:
                W = SF.gram(basis, ev['w'])
                W0, drift = W, 0.0
            B = c[:, None] * W * c[None, :]
            z = np.linalg.solve(B + a * np.eye(len(c)), rhs)
            delta = c * z
            # Bryan's bound by shortening, as the kernel does
            nrm = float(delta @ (SF.gram(basis, ev['w']) @ delta))
            if nrm > 0.2 * sumD:
                delta *= np.sqrt(0.2 * sumD / nrm)
            evt = SF.evaluate(basis, el, a, v - delta)
            n += 1
            du = basis.V @ delta
            relH = np.linalg.norm(ev['w'] * du) / np.linalg.norm(ev['H'])
            drift += float(np.max(np.abs(du)))
            kappa = np.expm1(drift) if (chord and it > 0) else np.expm1(float(np.max(np.abs(du))))
            v, ev = v - delta, evt
            if min(relH, (kappa + theta) * relH) < tol:
                break
        evals.append(n)
    return np.array(evals), ev


def problem(kind):
    if kind == 'normal':
        tau, omega, K, G = synthetic.single_G(200, 500)
    else:
        b = bench.build_batch(4, 200, 500, 100, 0)
        tau, omega, K, G = b['tau'], b['omega'], b['K'], b['Gmat'][0, 1]
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(len(tau))
    return tau, omega, K, G, D, err


if __name__ == '__main__':
    for kind in ('normal', 'plusminus'):
        tau, omega, K, G, D, err = problem(kind)
        basis = SF.Basis(K.U, K.S, K.V, err)
        el = SF.Element(basis, G, D, kind)
        alphas = np.array(synthetic.alpha_mesh(100)) * len(tau)
        v0 = basis.from_v(hostprep.initial_v(K.V, D, omega.delta, device.ENTROPY_NORMAL if kind == 'normal' else device.ENTROPY_PLUSMINUS))
        for name, chord in (('newton', False), ('chord', True)):
            ev, last = chain(basis, el, alphas, v0, chord)
            print('%-10s %-7s evaluations per alpha: mean %.2f (first alpha %d, the other 99: mean %.2f, max %d)  histogram %s' %
                  (kind, name, ev.mean(), ev[0], ev[1:].mean(), ev[1:].max(), np.bincount(ev[1:]).tolist()), flush=True)
