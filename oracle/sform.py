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
the reference's mu I, levenberg_minimizer.py:185-192).  mu starts at 0 (pure Newton) and is raised only until
Bryan's step bound  delta^T W delta <= 0.2 sum(D)  holds and the trial point
is finite, instead of the reference's multiplicative scan for a monotone
decrease of Q; the fixed point (g' = 0) is the same.
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
    """mirror of ``mxe_opts`` (include/maxent_hip.h)."""

    def __init__(self, maxiter=1000, miniter=0, tol_h=1e-9, tol_d=0.0,
                 tol_relq=0.0, step_max=0.2, mu_first=1e-3, mu_grow=4.0,
                 mu_max=1e20, stop_estimate=True, decouple_tol=1e-5):
        self.maxiter = maxiter
        self.miniter = miniter
        self.tol_h = tol_h            # |w o V delta| / |H| < tol_h
        self.tol_d = tol_d            # max |W g| < tol_d  (MaxDerivative)
        self.tol_relq = tol_relq      # |Q0-Q1|/|Q1| < tol_relq
        self.step_max = step_max      # delta^T W delta <= step_max * sum(D)
        self.mu_first = mu_first
        self.mu_grow = mu_grow
        self.mu_max = mu_max
        self.stop_estimate = stop_estimate   # tol_h also on (expm1(max|du|) + decouple_tol) * relH
        self.decouple_tol = decouple_tol     # only enters the estimate here (the model solves the full block)


def solve_alpha(basis, el, alpha, v, ev, opts, stats=None):
    """numpy model of one alpha-solve of the HIP kernel (same control flow as
    maxent_amd/csrc/mxe_kernel.hip.h).  ``ev``: evaluation at ``v`` (it does
    not depend on alpha, so the chain carries it over)."""
    c = basis.c
    sumD = el.D.sum() * (1.0 if el.entropy == 'normal' else 2.0)
    n_evals = 0
    n_chol = 0
    converged = False
    Qprev = np.nan
    Q = 0.5 * ev['chi2'] - alpha * ev['S']
    n_iter = 0
    for it in range(opts.maxiter):
        g = c * ev['rho'] + alpha * v
        rhs = ev['rho'] + alpha * v / c
        W = gram(basis, ev['w'])
        stop = False
        if opts.tol_d > 0 and np.max(np.abs(np.dot(W, g))) < opts.tol_d:
            stop = True
        if opts.tol_relq > 0 and it > 0 and \
                abs(abs(Qprev - Q) / Q) < opts.tol_relq:
            stop = True
        if stop and it >= opts.miniter:
            converged = True
            break
        B = c[:, None] * W * c[None, :]
        mu = 0.0
        accepted = False
        while True:
            good = True
            try:
                L = np.linalg.cholesky(B + (alpha + mu) * np.eye(len(c)))
                n_chol += 1
            except np.linalg.LinAlgError:
                good = False
            if good:
                z = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
                delta = c * z
                Wd = np.dot(W, delta)
                if not (np.dot(delta, Wd) <= opts.step_max * sumD):
                    good = False
            if good:
                evt = evaluate(basis, el, alpha, v - delta)
                n_evals += 1
                Qt = 0.5 * evt['chi2'] - alpha * evt['S']
                if not np.isfinite(Qt):
                    good = False
                elif mu > 0.0 and Qt > Q:
                    good = False      # a damped step must not make Q worse
            if good:
                accepted = True
                break
            mu = opts.mu_first * alpha if mu == 0.0 else mu * opts.mu_grow
            if not (mu <= opts.mu_max * alpha):
                break
        if not accepted:
            break
        du = np.dot(basis.V, delta)
        dH = ev['w'] * du
        relH = np.linalg.norm(dH) / np.linalg.norm(ev['H'])
        relH_next = relH
        if opts.stop_estimate and mu == 0.0:
            relH_next = (np.expm1(np.max(np.abs(du))) + opts.decouple_tol) * relH
        v = v - delta
        ev = evt
        Qprev = Q
        Q = 0.5 * ev['chi2'] - alpha * ev['S']
        n_iter += 1
        if opts.tol_h > 0 and min(relH, relH_next) < opts.tol_h \
                and n_iter > opts.miniter:
            converged = True
            break
    if stats is not None:
        stats.append((n_iter, n_evals, n_chol))
    return v, ev, n_iter, converged


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
    ev = evaluate(basis, el, alphas_scaled[0], v)
    for ia, a in enumerate(alphas_scaled):
        v, ev, n_iter, conv = solve_alpha(basis, el, a, v, ev, opts, stats)
        out['v'][ia] = v
        out['H'][ia] = ev['H']
        out['chi2'][ia] = ev['chi2']
        out['S'][ia] = ev['S']
        out['Q'][ia] = 0.5 * ev['chi2'] - a * ev['S']
        out['n_iter'][ia] = n_iter
        out['converged'][ia] = conv
    return out
