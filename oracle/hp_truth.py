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

The step is preconditioned with the singular-space Hessian, which only
affects the speed of convergence, not the fixed point.
"""

import numpy as np

LD = np.longdouble


def _chol_solve(A, b):
    n = len(b)
    L = np.zeros((n, n), dtype=LD)
    for j in range(n):
        s = A[j, j] - np.dot(L[j, :j], L[j, :j])
        L[j, j] = np.sqrt(s)
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - np.dot(L[i, :j], L[j, :j])) / L[j, j]
    y = np.zeros(n, dtype=LD)
    for i in range(n):
        y[i] = (b[i] - np.dot(L[i, :i], y[:i])) / L[i, i]
    x = np.zeros(n, dtype=LD)
    for i in range(n - 1, -1, -1):
        x[i] = (y[i] - np.dot(L[i + 1:, i], x[i + 1:])) / L[i, i]
    return x


def polish(K, G, err, D, V, S, alpha, v0, entropy='normal', iters=5,
           history=None):
    """Newton-polish ``v0`` (in the basis V, V^T V = 1) in extended precision.

    ``S``: singular values of K (only used to precondition).  Returns
    ``(v, H)`` rounded to binary64.
    """
    err = np.asarray(err, dtype=float) * np.ones(len(G))
    K_ = K.astype(LD)
    G_ = np.asarray(G).astype(LD)
    e_ = err.astype(LD)
    D_ = np.asarray(D).astype(LD)
    V_ = V.astype(LD)
    v = np.asarray(v0).astype(LD)
    a = LD(alpha)
    # preconditioner scale: c_k ~ S_k / typical error
    c_ = (np.asarray(S) / np.exp(np.mean(np.log(err)))).astype(LD)
    c_ = np.maximum(c_, LD(1e-150) * c_.max())

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

    for it in range(iters):
        H, w = state(v)
        r = (K_ @ H - G_) / e_ ** 2
        g = V_.T @ (K_.T @ r) + a * v
        W = (V_.T * w) @ V_
        A = (c_[:, None] * W * c_[None, :]) + a * np.eye(len(c_), dtype=LD)
        dl = c_ * _chol_solve(A, g / c_)
        if history is not None:
            history.append((float(np.abs(g).max()),
                            float(np.linalg.norm(w * (V_ @ dl)) /
                                  np.linalg.norm(H))))
        v = v - dl
    H, _ = state(v)
    return v.astype(float), H.astype(float)
