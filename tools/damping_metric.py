#!/usr/bin/env python3
"""Which damping for the alphas that crawl?  A numpy model of the one-chain kernel's iteration on one scan of a stress case
(warm chain down the mesh, the device's acceptance rules), with the damping added
  (a) as the device adds it:         (c W c + (alpha + mu) I) z = rhs                    [whitened basis, delta = c z]
  (b) as the reference adds it:      (W M W + alpha W + mu I) delta = -W g                [levenberg_minimizer.py:155-243 on
      maxent_cost_function.py's d = W g, dd = W M W + alpha W: damping relative to W, heavy where the spectrum vanishes]
and the iterations / evaluations each needs per alpha.   python tools/damping_metric.py [case] [element] [seed]      (CPU only)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('OMP_NUM_THREADS', '1')
import stress
from maxent_amd import device

case = int(sys.argv[1]) if len(sys.argv) > 1 else 43
elem = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
c = [x for x in stress.cases(case + 1, seed) if x['case'] == case][0]
tau, omega, K, Gmat, D, err, alphas, elems, kinds, v0s = stress.inputs(c)
i, j = elems[elem]
kind = kinds[elem]
U, S, V = np.array(K.U), np.array(K.S), np.array(K.V)
ns = len(S)
cvec = S / err[0]
ghat = U.T @ (Gmat[i, j] / err[0])
cperp = float(np.sum((Gmat[i, j] / err[0]) ** 2) - np.sum(ghat ** 2))
sumD = (2.0 if kind == device.ENTROPY_PLUSMINUS else 1.0) * D.sum()
step_lim = 0.2 * sumD


def evaluate(v, alpha):
    u = V @ v
    if kind == device.ENTROPY_NORMAL:
        H = D * np.exp(u); w = H
        Sent = np.sum(H - D - H * u)
    else:
        ep, em = D * np.exp(u), D * np.exp(-u)
        H = ep - em; w = ep + em
        Sent = np.sum(ep - D - ep * u) + np.sum(em - D + em * u)
    h = V.T @ H
    rho = cvec * h - ghat
    chi2 = float(rho @ rho) + cperp
    return dict(H=H, w=w, S=Sent, rho=rho, chi2=chi2, Q=0.5 * chi2 - alpha * Sent)


def solve_alpha(v, alpha, mode, maxiter=1500, tol=1e-9):
    st = evaluate(v, alpha)
    nev, mu_hint = 1, 0.0
    for it in range(maxiter):
        W = V.T @ (st['w'][:, None] * V)
        g = cvec * st['rho'] + alpha * v                      # grad Q = W g
        mu = 0.0
        accepted = False
        while True:
            if mode == 'device':
                B = cvec[:, None] * W * cvec[None, :] + (alpha + mu) * np.eye(ns)
                z = np.linalg.solve(B, -(st['rho'] + alpha * v / cvec))
                delta = cvec * z
            else:
                wscale = 1.0                                   # mu in the units of (W M W + alpha W)
                J = W @ (cvec[:, None] ** 2 * W) + alpha * W + mu * wscale * np.eye(ns)
                delta = np.linalg.solve(J, -(W @ g))
            nrm = float(delta @ W @ delta)
            scaled = False
            d = delta
            ok = np.isfinite(nrm)
            if ok and nrm > step_lim:
                if mu == 0.0:
                    d = delta * np.sqrt(step_lim / nrm); scaled = True
                else:
                    ok = False
            if ok:
                tr = evaluate(v + d, alpha); nev += 1
                if not np.isfinite(tr['Q']):
                    ok = False
                elif (mu > 0.0 or scaled) and tr['Q'] > st['Q'] + 1e-12 * abs(st['Q']):
                    ok = False
                elif tr['Q'] > 1e6 * (abs(st['Q']) + 1.0):
                    ok = False
            if ok:
                accepted = True
                break
            if mode == 'device':
                mu = max(1e-3 * alpha, mu_hint / 4.0) if mu == 0.0 else mu * 4.0
                if mu > 1e20 * alpha: break
            else:
                base = 1e-3 * alpha * float(np.max(np.diag(W)))            # the same first damping, in the units of W
                mu = max(base, mu_hint / 4.0) if mu == 0.0 else mu * 4.0
                if mu > 1e30: break
        if not accepted:
            return v, it, nev, False, mu
        relH = np.linalg.norm(tr['H'] - st['H']) / np.linalg.norm(st['H'])
        # the correction a full Newton step would have been: solve undamped and compare (the model's stopping test)
        v = v + d
        st = tr
        mu_hint = mu
        if mu == 0.0 and not scaled and relH < tol:
            return v, it + 1, nev, True, mu
        if mu > 0.0 or scaled:
            # is the undamped correction small already?
            Wn = V.T @ (st['w'][:, None] * V)
            Bn = cvec[:, None] * Wn * cvec[None, :] + alpha * np.eye(ns)
            zn = np.linalg.solve(Bn, -(st['rho'] + alpha * v / cvec))
            un = V @ (cvec * zn)
            corr = np.linalg.norm(st['w'] * un) / np.linalg.norm(st['H'])
            if corr < tol:
                return v, it + 1, nev, True, mu
    return v, maxiter, nev, False, mu_hint


print('case %d element %s kind %s: n_tau %d n_omega %d n_s %d sigma %.1e, %d alphas %.2e .. %.2e' % (
    case, (i, j), 'normal' if kind == device.ENTROPY_NORMAL else 'plusminus', len(tau), len(D), ns, err[0], len(alphas), alphas[0], alphas[-1]))
for mode in ('device', 'reference'):
    v = v0s[elem].copy()
    tot_it = tot_ev = 0
    rows = []
    for ia, alpha in enumerate(alphas):
        v, it, nev, conv, mu = solve_alpha(v, alpha, mode)
        tot_it += it; tot_ev += nev
        rows.append('%d:%d%s' % (ia, it, '' if conv else '!'))
    print('%-10s iterations %6d evaluations %6d | per alpha (index:iterations, ! = not converged): %s' % (mode, tot_it, tot_ev, ' '.join(rows)))
