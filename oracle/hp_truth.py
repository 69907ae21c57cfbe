"""Extended-precision fixed point of the reference's cost function.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The minimiser of Q_alpha over span(V) is unique (Q is strictly convex in H).
The reference's own Levenberg iteration stops, by its two stopping rules
(convergence_methods.py:81-122), up to ~2e-5 (relative L2 in H) away from it
at the smallest alphas -- also when asked for ``MaxDerivative(1e-9)``: its
Newton system (W M W + alpha W) is too ill-conditioned in binary64 to go
further (measured; see DESIGN.md "Parity").  To have a golden answer that is
good to far better than the 1e-6 parity gate, this module polishes a
reference result with Newton steps carried out in x87 extended precision
(``np.longdouble``, 64-bit mantissa) on the gradient of the reference's
cost function written with the *full* kernel K exactly as the reference does
(functions.py:362-365, 512-514; maxent_cost_function.py:85-118):

    g(v) = V^T [ K^T ((K H - G)/err^2) + alpha (log H+ - log D) ],
    log(H+/D) = V v  exactly for H = D e^{Vv} (normal) and for H+ (plusminus)

The Newton matrix is the exact Jacobian of g, M W + alpha I with
M = V^T K^T diag(1/err^2) K V, also in extended precision.
"""

import numpy as np

LD = np.longdouble
TOL = 1e-12            # relative Newton correction at which a polish counts as converged
CAP = 2.0              # largest rise of u in one guarded step
MAX_GUARDED = 400      # iterations of the guarded (line-search) form


def _lu_solve(A, b):
    """Gaussian elimination with partial pivoting in extended precision."""
    A = A.copy()
    b = b.copy()
    n = len(b)
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            b[[k, p]] = b[[p, k]]
        f = A[k + 1:, k] / A[k, k]
        A[k + 1:, k:] -= f[:, None] * A[k, k:][None, :]
        b[k + 1:] -= f * b[k]
    x = np.zeros(n, dtype=LD)
    for i in range(n - 1, -1, -1):
        x[i] = (b[i] - np.dot(A[i, i + 1:], x[i + 1:])) / A[i, i]
    return x


def polish(K, G, err, D, V, S, alpha, v0, entropy='normal', iters=5,
           history=None, info=None):
    """Newton-polish ``v0`` (in the basis V, V^T V = 1) in extended precision.

    ``S``: unused (kept for call compatibility).  Returns
    ``(v, H)`` rounded to binary64.

    ``info`` (a dict, optional) receives ``converged`` (the last Newton
    correction, ``|w * V dv| / |H|``, is below ``TOL``), ``corr`` (that
    correction), ``iterations`` and ``damped``.  The plain iteration -- ``iters``
    full steps, what every committed fixture was made with -- runs first and its
    result is returned unchanged when it converged.  Where it does not (a start
    whose gradient has a large component along a direction in which H is ~0: the
    full step lifts u there by hundreds and exp overflows -- element (15, 15) of
    BASELINE config 4 at the smallest alpha), the iteration is repeated from
    ``v0`` with the step scaled so that u rises by at most ``CAP`` anywhere
    (full steps must also lower |g|), until the correction is below ``TOL``.
    """
    err = np.asarray(err, dtype=float) * np.ones(len(G))
    K_ = K.astype(LD)
    G_ = np.asarray(G).astype(LD)
    e_ = err.astype(LD)
    D_ = np.asarray(D).astype(LD)
    V_ = V.astype(LD)
    a = LD(alpha)
    # exact singular-space curvature of chi2/2:  M = V^T K^T diag(1/err^2) K V
    KV = K_ @ V_
    M = (KV / e_[:, None] ** 2).T @ KV
    eye = np.eye(V_.shape[1], dtype=LD)

    def state(v):
        u = V_ @ v
        if entropy == 'normal':
            H = D_ * np.exp(u)
            w = H
        else:
            Hp = D_ * np.exp(u)
            Hm = D_ * np.exp(-u)
            H = Hp - Hm
            w = Hp + Hm
        return H, w

    def grad(v, H):
        r = (K_ @ H - G_) / e_ ** 2
        return V_.T @ (K_.T @ r) + a * v

    def newton(v, H, w, g):
        W = (V_.T * w) @ V_
        # exact Newton step:  d g / d v = M W + alpha I  (row-equilibrated)
        J = M @ W + a * eye
        sc = 1 / np.max(np.abs(J), axis=1)
        dl = _lu_solve(J * sc[:, None], g * sc)
        return dl, float(np.linalg.norm(w * (V_ @ dl)) / np.linalg.norm(H))

    def report(converged, corr, n, damped):
        if info is not None:
            info.update(converged=bool(converged), corr=float(corr), iterations=n, damped=damped)

    # --- the plain iteration (bit for bit what made the fixtures) ---
    v = np.asarray(v0).astype(LD)
    corr = np.inf
    with np.errstate(all='ignore'):
        for it in range(iters):
            H, w = state(v)
            g = grad(v, H)
            dl, corr = newton(v, H, w, g)
            if history is not None:
                history.append((float(np.abs(g).max()), corr))
            v = v - dl
        H, _ = state(v)
    if np.all(np.isfinite(H)) and np.all(np.isfinite(v)) and corr < TOL:
        report(True, corr, iters, False)
        return v.astype(float), H.astype(float)

    # --- guarded iteration: the same steps, capped in max(du) ---
    v = np.asarray(v0).astype(LD)
    H, w = state(v)
    g = grad(v, H)
    gn = np.linalg.norm(g)
    corr, n = np.inf, 0
    with np.errstate(all='ignore'):
        for n in range(1, MAX_GUARDED + 1):
            dl, corr = newton(v, H, w, g)
            # H grows by at most e^CAP per step anywhere: the exponential is what the linear model misses
            du = V_ @ dl
            rise = float(np.max(-du if entropy == 'normal' else np.abs(du)))
            t = LD(1) if rise <= CAP else LD(CAP / rise)
            while True:
                vt = v - t * dl
                Ht, wt = state(vt)
                gt = grad(vt, Ht)
                gtn = np.linalg.norm(gt)
                if np.isfinite(gtn) and (gtn < gn or t < 1):
                    break
                t = t / 2
                if t < LD(2) ** -60:
                    report(False, corr, n, True)
                    return vt.astype(float), np.full(len(D_), np.nan)
            if history is not None:
                history.append((float(np.abs(g).max()), corr, float(t)))
            v, H, w, g, gn = vt, Ht, wt, gt, gtn
            if t == 1 and corr < TOL:
                break
    report(corr < TOL, corr, n, True)
    return v.astype(float), H.astype(float)
