"""Whitened singular-space ("S-form") restatement + numpy model of the HIP kernel.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference evaluates Q_alpha(v) = 1/2 chi2(H(v)) - alpha S(H(v)) with the
full kernel K (functions.py:358-365) and dense n_omega x n_omega second
derivatives (maxent_cost_function.py:120-165).  With K = U S V^T (kernels.py:
53-64) everything the minimiser needs lives in the n_s-dimensional singular
space (notation):

    u = V v,  H = D e^u  (normal)   |  H+- = D e^{+-u}, H = H+ - H-  (plusminus)
    w = H                           |  w = H+ + H-
    h = V^T H,   W = V^T diag(w) V
    g = M h - b + alpha v,  M = S U^T diag(1/err^2) U S,  b = S U^T (G/err^2)
    MaxEntCostFunction:  d = W g,  dd = W M W + alpha W
    BryanCostFunction :  d = g,    dd = M W

This module rotates the singular basis once more on the host so that M is
diagonal ("whitened basis"):  C = diag(1/err) U S = Uhat diag(c) Q^T (thin
SVD, n_tau x n_s) and V' = V Q.  Then, with v' = Q^T v (u = V v = V' v'),

    chi2 = sum_k (c_k h'_k - ghat_k)^2 + c_perp ,  ghat = Uhat^T (G/err),
    c_perp = || (I - Uhat Uhat^T) G/err ||^2                (a constant)
    g' = c * rho + alpha v' ,  rho = c h' - ghat ,  M' = diag(c^2)

i.e. chi2 is a sum of squares in the rotated data space (no cancellation) and
no U mat-vec is left in the inner loop.  The minimiser of Q is unchanged
(span V' = span V), so H, A, chi2, S, Q at the optimum equal the reference's.

The damped Newton step the kernel takes is Bryan's: with a = alpha + mu,

    (M' W + a I) delta = g'   <=>   (c W c + a I) z = rho + alpha v'/c ,
    delta = c * z      (symmetric positive definite, Cholesky)

which is the reference's MaxEntCostFunction step (W M W + alpha W + damping)
delta = W g with the damping taken in the entropy metric (mu W instead of
the reference's mu I, levenberg_minimizer.py:185-192).  mu is steered by the
gain ratio of actual to predicted reduction instead of the reference's
multiplicative scan; the fixed point (g' = 0) is the same.
"""

import numpy as np


class Basis(object):
    """Whitened singular basis for one (K, err) pair."""

    def __init__(self, U, S, V, err):
        n_tau = U.shape[0]
        err = np.asarray(err, dtype=float) * np.ones(n_tau)
        self.err = err
        C = (U * S[np.newaxis, :]) / err[:, np.newaxis]
        if np.all(err == err[0]):
            # U has orthonormal columns: C is already Uhat diag(c)
            self.Q = None
            self.c = S / err[0]
            self.Uhat = U
            self.V = np.ascontiguousarray(V)
        else:
            Uh, c, Qt = np.linalg.svd(C, full_matrices=False)
            self.Q = Qt.T
            self.c = c
            self.Uhat = Uh
            self.V = np.ascontiguousarray(np.dot(V, self.Q))
        # guard against exactly vanishing weights (rank-deficient rotated data)
        self.c = np.maximum(self.c, 1e-150 * np.max(self.c))
        self.n_s = len(self.c)
        self.n_omega = V.shape[0]

    def to_v(self, vprime):
        """v in the caller's (reference) singular basis."""
        return vprime if self.Q is None else np.dot(self.Q, vprime)

    def from_v(self, v):
        return v if self.Q is None else np.dot(self.Q.T, v)


class Element(object):
    """Per matrix-element data in the whitened basis."""

    def __init__(self, basis, G, D, entropy='normal'):
        Gt = np.asarray(G, dtype=float) / basis.err
        self.ghat = np.dot(basis.Uhat.T, Gt)
        resid = Gt - np.dot(basis.Uhat, self.ghat)
        self.c_perp = float(np.dot(resid, resid))
        self.D = np.asarray(D, dtype=float)
        self.entropy = entropy


def evaluate(basis, el, alpha, v):
    """One cost evaluation pass (SURVEY 8d 'unit of work')."""
    u = np.dot(basis.V, v)
    with np.errstate(all='ignore'):
        if el.entropy == 'normal':
            H = el.D * np.exp(u)
            w = H
            S = np.sum(H - el.D - H * u)
        else:
            Hp = el.D * np.exp(u)
            Hm = el.D * np.exp(-u)
            H = Hp - Hm
            w = Hp + Hm
            S = np.sum((Hp - el.D - Hp * u) + (Hm - el.D + Hm * u))
        h = np.dot(basis.V.T, H)
        rho = basis.c * h - el.ghat
        chi2 = float(np.dot(rho, rho)) + el.c_perp
        Q = 0.5 * chi2 - alpha * S
    return dict(u=u, H=H, w=w, h=h, rho=rho, chi2=chi2, S=S, Q=Q)


def gram(basis, w):
    return np.dot(basis.V.T * w[np.newaxis, :], basis.V)


class KernelOptions(object):
    def __init__(self, maxiter=1000, tol_d=1e-4, tol_relQ=1e-16,
                 tol_pred=1e-17, mu_factor0=0.0, nu=2.0, max_reject=60,
                 miniter=0):
        self.maxiter = maxiter
        self.tol_d = tol_d            # max |W g| < tol_d  (MaxDerivative)
        self.tol_relQ = tol_relQ      # |Q0-Q1|/|Q1| < tol_relQ
        self.tol_pred = tol_pred      # predicted decrease < tol_pred*|Q|:
        #                               the Newton step is below round-off
        self.mu_factor0 = mu_factor0  # initial mu / alpha
        self.nu = nu
        self.max_reject = max_reject
        self.miniter = miniter


def solve_alpha(basis, el, alpha, v, opts, stats=None):
    """numpy model of one alpha-solve of the HIP kernel (same control flow)."""
    ev = evaluate(basis, el, alpha, v)
    n_evals = 1
    n_chol = 0
    mu = opts.mu_factor0 * alpha
    nu = opts.nu
    Q0 = np.nan
    converged = False
    c = basis.c
    it = -1
    for it in range(opts.maxiter):
        g = c * ev['rho'] + alpha * v
        W = gram(basis, ev['w'])
        d = np.dot(W, g)
        maxd = np.max(np.abs(d))
        conv = maxd < opts.tol_d
        if not np.isnan(Q0):
            conv = conv or abs(abs(Q0 - ev['Q']) / ev['Q']) < opts.tol_relQ
        if conv and it >= opts.miniter:
            converged = True
            break
        rhs = ev['rho'] + alpha * v / c
        B = c[:, None] * W * c[None, :]
        accepted = False
        for rej in range(opts.max_reject):
            a = alpha + mu
            A = B + a * np.eye(len(c))
            try:
                L = np.linalg.cholesky(A)
                n_chol += 1
            except np.linalg.LinAlgError:
                mu = max(mu * nu, alpha * 1e-3)
                nu *= 2
                continue
            z = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            delta = c * z
            vt = v - delta
            evt = evaluate(basis, el, alpha, vt)
            n_evals += 1
            Wd = np.dot(W, delta)
            pred = 0.5 * (np.dot(g, Wd) + mu * np.dot(delta, Wd))
            actual = ev['Q'] - evt['Q']
            if pred <= opts.tol_pred * abs(ev['Q']):
                # stalled at round-off: the current point is the answer
                converged = True
                break
            ok = np.isfinite(evt['Q']) and (actual > 0 or
                                            abs(actual) <= 1e-15 * abs(ev['Q']))
            if ok:
                rho_gain = actual / pred if pred > 0 else 1.0
                # Nielsen's update
                mu = mu * max(1.0 / 3.0, 1.0 - (2.0 * rho_gain - 1.0) ** 3)
                if mu < 1e-8 * alpha:
                    mu = 0.0
                nu = opts.nu
                Q0 = ev['Q']
                v = vt
                ev = evt
                accepted = True
                break
            mu = max(mu * nu, alpha * 1e-3)
            nu *= 2
        if converged or not accepted:
            break
    if stats is not None:
        stats.append((it + 1, n_evals, n_chol))
    return v, ev, it + 1, converged


def alpha_chain(basis, el, alphas_scaled, v0, opts=None, stats=None):
    """Warm-started alpha scan (maxent_loop.py:241-245) with the kernel model.

    ``v0`` in the whitened basis.  Returns dict(H, chi2, S, Q, v, n_iter,
    converged) with v in the whitened basis.
    """
    if opts is None:
        opts = KernelOptions()
    X = len(alphas_scaled)
    out = dict(v=np.empty((X, basis.n_s)), H=np.empty((X, basis.n_omega)),
               chi2=np.empty(X), S=np.empty(X), Q=np.empty(X),
               n_iter=np.zeros(X, dtype=int),
               converged=np.zeros(X, dtype=bool))
    v = np.array(v0, dtype=float)
    for ia, a in enumerate(alphas_scaled):
        v, ev, n_iter, conv = solve_alpha(basis, el, a, v, opts, stats)
        out['v'][ia] = v
        out['H'][ia] = ev['H']
        out['chi2'][ia] = ev['chi2']
        out['S'][ia] = ev['S']
        out['Q'][ia] = ev['Q']
        out['n_iter'][ia] = n_iter
        out['converged'][ia] = conv
    return out
