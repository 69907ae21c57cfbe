"""Step-faithful numpy restatement of the reference's alpha-scan path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  This file restates, in plain
numpy and in a functional style of its own, what TRIQS/maxent v1.2.0 computes
on the path  TauMaxEnt.run() -> MaxEntLoop.run() -> LevenbergMinimizer
.minimize() -> MaxEntCostFunction / BryanCostFunction -> functions.py.
It deliberately keeps the reference's arithmetic (full-K mat-vecs, dense
n_omega x n_omega second derivatives, LAPACK ``solve``) so that iteration
counts and iterates agree with the reference to round-off; it is therefore
also the "reference CPU path" that bench.py times as ``cpu_baseline``
(kind "port").

All ``file:line`` citations are relative to /root/reference/python/.
"""

import time

import numpy as np


# --------------------------------------------------------------------------
#  grids and inputs
# --------------------------------------------------------------------------

def hyperbolic_omega_mesh(omega_min=-10.0, omega_max=10.0, n_points=100):
    """omega_meshes.py:215-222 (HyperbolicOmegaMesh)."""
    u = np.linspace(-1, 1, n_points)
    w = np.sign(u) * (np.sqrt(1 + u ** 2) - 1)
    return omega_min + (omega_max - omega_min) * (w - w[0]) / (w[-1] - w[0])


def linear_omega_mesh(omega_min=-10.0, omega_max=10.0, n_points=100):
    """omega_meshes.py:86-88 (LinearOmegaMesh)."""
    return np.linspace(omega_min, omega_max, n_points)


def lorentzian_omega_mesh(omega_min=-10.0, omega_max=10.0, n_points=100,
                          cut=0.01):
    """omega_meshes.py:132-144 (LorentzianOmegaMesh)."""
    u = np.linspace(0, 1, n_points + 1)
    temp = np.tan(np.pi * (u * (1. - 2 * cut) + cut - 0.5))
    t = (temp - temp[0]) / (temp[-1] - temp[0])
    w = omega_min + (omega_max - omega_min) * t
    w = (w[:-1] + w[1:]) / 2.0
    w = (w - w[0]) / (w[-1] - w[0]) * (omega_max - omega_min) + omega_min
    return w


def omega_delta(omega):
    """omega_meshes.py:54-62 (BaseOmegaMesh.delta, trapezoid weights)."""
    delta = np.empty(len(omega))
    delta[1:-1] = (omega[2:] - omega[:-2]) / 2.0
    delta[0] = (omega[1] - omega[0]) / 2.0
    delta[-1] = (omega[-1] - omega[-2]) / 2.0
    return delta


def log_alpha_mesh(alpha_min=0.0001, alpha_max=20, n_points=20):
    """alpha_meshes.py:81-85 (LogAlphaMesh; descending)."""
    return np.logspace(np.log10(alpha_min), np.log10(alpha_max),
                       n_points)[::-1].copy()


def flat_default_model(omega):
    """default_models.py:61-63 (FlatDefaultModel.D; includes delta)."""
    delta = omega_delta(omega)
    return np.ones(omega.shape) / np.sum(delta) * delta


def data_default_model(default, omega):
    """default_models.py:88-93 (DataDefaultModel on the same grid)."""
    return np.asarray(default) * omega_delta(omega)


def tau_kernel(tau, omega, beta=None):
    """kernels.py:244-271 (TauKernel._fill_values). Returns (K, K_delta)."""
    tau = np.asarray(tau, dtype=float)
    omega = np.asarray(omega, dtype=float)
    if beta is None:
        beta = tau[-1]
    oomega, ttau = np.meshgrid(omega, tau)
    L = oomega >= 0.0
    iL = np.where(L)
    nL = np.where(np.logical_not(L))
    K = np.empty(oomega.shape)
    K[iL] = -np.exp(-oomega[iL] * ttau[iL]) / \
        (np.exp(-beta * oomega[iL]) + 1.0)
    K[nL] = -np.exp(oomega[nL] * (beta - ttau[nL])) / \
        (1.0 + np.exp(beta * oomega[nL]))
    K_delta = np.einsum('ij,j->ij', K, omega_delta(omega))
    return K, K_delta


def get_preblur(omega, b):
    """preblur.py:31-58 (Gaussian blur matrix, rows then columns normalised)."""
    omega = np.asarray(omega, dtype=float)
    delta = omega_delta(omega)
    w1, w2 = np.meshgrid(omega, omega)
    B = np.exp(-(w1 - w2) ** 2 / 2.0 / b ** 2) / np.sqrt(2.0 * np.pi * b ** 2)
    renorm = np.dot(delta, B)
    B = B / renorm[:, np.newaxis]
    renorm = np.dot(B, delta)
    B = B / renorm[np.newaxis, :]
    return B


def preblur_kernel(K, omega, b):
    """kernels.py:384-393 (PreblurKernel._fill_values): K' = K (diag(dw) B)."""
    B = get_preblur(omega, b)
    return np.dot(K, np.einsum('ij,i->ij', B, omega_delta(omega))), B


def svd_reduce(K, threshold=1.e-14):
    """kernels.py:53-64 + 101-122 (svd, reduce_singular_space).

    Returns U (n_tau x n_s), S (n_s), V (n_omega x n_s) with S >= threshold
    (absolute threshold).
    """
    U, S, Vh = np.linalg.svd(K, full_matrices=False)
    V = Vh.transpose()
    L = np.where(S >= threshold)[0]
    return U[:, L], S[L], V[:, L]


def safelog(A):
    """functions.py:53-56 (clamps |x|<=1e-100 IN PLACE, then log)."""
    A[np.where(np.abs(A) <= 1.e-100)] = 1.e-100
    return np.log(A)


# --------------------------------------------------------------------------
#  the cost function, H-form, exactly as the reference evaluates it
# --------------------------------------------------------------------------

class Problem(object):
    """Inputs of one alpha scan (one matrix element).

    K      n_tau x n_omega kernel (possibly rotated / pre-blurred)
    U,S,V  truncated SVD of K (svd_reduce)
    G,err  data and its (diagonal) error
    D      default model including delta-omega
    entropy 'normal' | 'plusminus'
    form   'maxent' (MaxEntCostFunction, d_dv=False, dA_projection=2)
           | 'bryan' (BryanCostFunction; normal entropy only)
    """

    def __init__(self, K, U, S, V, G, err, D, entropy='normal',
                 form='maxent', chi2_factor=1.0):
        self.K = K
        self.U, self.S, self.V = U, S, V
        self.G = np.asarray(G, dtype=float)
        self.err = np.asarray(err, dtype=float) * np.ones(len(self.G))
        self.D = np.asarray(D, dtype=float)
        self.entropy = entropy
        self.form = form
        self.chi2_factor = chi2_factor
        assert not (form == 'bryan' and entropy != 'normal')
        # functions.py:372-377 (NormalChi2.parameter_change): constant d2
        self.d2 = 2 * np.einsum('il,ik,i->kl', np.conjugate(K), K,
                                1. / self.err ** 2)
        self.n_evals = 0


def H_of_v(p, v):
    """functions.py:739-741 (normal) / 778-781 (plusminus)."""
    if p.entropy == 'normal':
        return p.D * np.exp(np.dot(p.V, v))
    return p.D * (np.exp(np.dot(p.V, v)) - np.exp(-np.dot(p.V, v)))


def dH_dv(p, v):
    """functions.py:743-746 (normal) / 783-786 (plusminus)."""
    if p.entropy == 'normal':
        return p.D[:, np.newaxis] * p.V * \
            np.exp(np.dot(p.V, v))[:, np.newaxis]
    return p.D[:, np.newaxis] * p.V * (
        np.exp(np.dot(p.V, v))[:, np.newaxis] +
        np.exp(-np.dot(p.V, v))[:, np.newaxis])


def H_of_v_inv(p, A):
    """functions.py:753-755 (normal) / 793-796 (plusminus)."""
    if p.entropy == 'normal':
        return np.dot(p.V.transpose(), safelog(A / p.D))
    return np.dot(p.V.transpose().conjugate(), safelog(
        (A + np.sqrt(A ** 2 + 4 * p.D ** 2)) / (2 * p.D)))


def chi2_f(p, H):
    """functions.py:358-360 (NormalChi2.f; Python ``sum``)."""
    return sum(np.abs(np.dot(p.K, H) - p.G) ** 2 / p.err ** 2)


def chi2_d(p, H):
    """functions.py:362-365 (NormalChi2.d)."""
    return np.dot(2 * (np.dot(p.K, H) - p.G) / p.err ** 2, np.conjugate(p.K))


def _normal_S_f(D, A):
    """functions.py:508-510 (NormalEntropy.f)."""
    return np.sum((A - D - A * safelog(A / D)))


def _A_plus(p, A):
    """functions.py:544-546."""
    return (np.sqrt(A ** 2.0 + 4.0 * p.D ** 2) + A) / 2.0


def _A_minus(p, A):
    """functions.py:548-550."""
    return (np.sqrt(A ** 2.0 + 4.0 * p.D ** 2) - A) / 2.0


def S_f(p, H):
    """functions.py:508-510 (normal) / 552-555 (plusminus)."""
    if p.entropy == 'normal':
        return _normal_S_f(p.D, H)
    return _normal_S_f(p.D, _A_plus(p, H)) + _normal_S_f(p.D, _A_minus(p, H))


def S_d(p, H):
    """functions.py:512-514 (normal) / 557-559 (plusminus)."""
    if p.entropy == 'normal':
        return -(safelog(H.copy()) - safelog(p.D.copy()))
    Ap = _A_plus(p, H)
    return -(safelog(Ap) - safelog(p.D.copy()))


def S_dd(p, H):
    """functions.py:516-520 (normal) / 561-564 (plusminus)."""
    if p.entropy == 'normal':
        A = H.copy()
    else:
        A = _A_plus(p, H) + _A_minus(p, H)
    A[np.where(np.abs(A) <= 1.e-100)] = 1.e-100
    return -np.diag(1.0 / A)


def Q_f(p, alpha, v):
    """maxent_cost_function.py:68-83 == bryan_cost_function.py:57-72."""
    p.n_evals += 1
    H = H_of_v(p, v)
    return 0.5 * chi2_f(p, H) * p.chi2_factor - alpha * S_f(p, H)


def Q_d(p, alpha, v):
    """maxent_cost_function.py:85-118 (dA_projection=2) /
    bryan_cost_function.py:84-102."""
    H = H_of_v(p, v)
    if p.form == 'bryan':
        dchi2 = 2 * (np.dot(p.K, H) - p.G) / p.err ** 2
        ret = p.S * np.dot(p.U.conjugate().transpose(),
                           0.5 * dchi2 * p.chi2_factor)
        return -(-ret - alpha * v)
    dQ_dH = 0.5 * chi2_d(p, H) * p.chi2_factor - alpha * S_d(p, H)
    T = dH_dv(p, v)
    return np.dot(T.transpose(), dQ_dH)


def Q_dd(p, alpha, v):
    """maxent_cost_function.py:120-165 (dA_projection=2) /
    bryan_cost_function.py:114-128."""
    H = H_of_v(p, v)
    if p.form == 'bryan':
        ret = np.dot(p.V.conjugate().transpose(), p.d2)
        ret = np.einsum('ij,j,jk->ik', ret, H, p.V)
        return 0.5 * ret * p.chi2_factor
    ddQ = 0.5 * p.d2 * p.chi2_factor - alpha * S_dd(p, H)
    T = dH_dv(p, v)
    return np.dot(T.transpose(), np.dot(ddQ, T))


# --------------------------------------------------------------------------
#  the minimiser
# --------------------------------------------------------------------------

class LevenbergOptions(object):
    """levenberg_minimizer.py:92-101 defaults; convergence :103-106."""

    def __init__(self, maxiter=1000, miniter=0, mu0=1.e-18, nu=1.3,
                 max_mu=1.e20, max_derivative=1.e-4, rel_function_change=1.e-16):
        self.maxiter = maxiter
        self.miniter = miniter
        self.mu0 = mu0
        self.nu = nu
        self.max_mu = max_mu
        # OrConvergenceMethod(MaxDerivative(1e-4), RelativeFunctionChange(1e-16))
        # set one of them to None to drop it (e.g. the tight goldens use
        # MaxDerivativeConvergenceMethod(1e-7) alone)
        self.max_derivative = max_derivative
        self.rel_function_change = rel_function_change


def _converged(opts, f, Q0, Q1):
    """convergence_methods.py:64-122."""
    is_conv = False
    if opts.max_derivative is not None:
        is_conv = is_conv or bool(np.max(np.abs(f)) < opts.max_derivative)
    if opts.rel_function_change is not None:
        with np.errstate(all='ignore'):
            conv = np.abs(np.abs(Q0 - Q1) / Q1)
        is_conv = is_conv or bool(conv < opts.rel_function_change)
    return is_conv


def levenberg_minimize(p, alpha, v0, opts):
    """levenberg_minimizer.py:123-248, statement by statement.

    Returns (v, n_iter, converged).  ``v0`` is updated IN PLACE like the
    reference does (:239).
    """
    converged = False
    mu = opts.mu0
    v = v0
    Q1 = Q_f(p, alpha, v)
    Q0 = np.nan
    nu = opts.nu
    i = -1
    for i in range(opts.maxiter):
        f = Q_d(p, alpha, v)
        J = Q_dd(p, alpha, v)
        converged = _converged(opts, f, Q0, Q1)
        if converged and i >= opts.miniter:
            break
        Id = np.eye(len(J))
        Q0 = Q1
        dv = np.linalg.solve(J + mu * Id, f)
        old = np.seterr(all='ignore')
        Q1 = Q_f(p, alpha, v - dv)
        while (Q1 > Q0 or np.isnan(Q1)) and mu < opts.max_mu:
            mu *= nu
            dv = np.linalg.solve(J + mu * Id, f)
            Q1 = Q_f(p, alpha, v - dv)
        dv2 = np.linalg.solve(J + nu * mu * Id, f)
        Q2 = Q_f(p, alpha, v - dv2)
        if Q2 < Q1:
            nuf = nu
            mu *= nu
            Q2 = Q1
            dvnew = dv2
        else:
            nuf = 1.0 / nu
            mu /= nuf
            dvnew = dv
        Q1 = np.inf
        while (Q2 < Q1 and mu < opts.max_mu
               and mu > nu * np.finfo(float).eps):
            Q1 = Q2
            dv = dvnew
            mu *= nuf
            dvnew = np.linalg.solve(J + mu * Id, f)
            Q2 = Q_f(p, alpha, v - dvnew)
        np.seterr(**old)
        v -= dv
        Q1 = Q_f(p, alpha, v)
    return v, i + 1, converged


# --------------------------------------------------------------------------
#  the alpha loop
# --------------------------------------------------------------------------

def initial_v(p, delta, A_init=None):
    """maxent_loop.py:196-203: v0 = H_of_v.inv(D.D * omega.delta).

    (D already contains delta; the reference multiplies by it again.)
    """
    right_side = (p.D if A_init is None else np.asarray(A_init)) * delta
    return H_of_v_inv(p, right_side.copy())


def alpha_loop(p, delta, alpha_mesh, opts=None, scale_alpha='Ndata',
               A_of_H=None, A_init=None, G_threshold=1.e-10, timing=None):
    """maxent_loop.py:144-302 (the solver part; no analyzers).

    ``A_of_H``: None -> IdentityA_of_H (A = H/delta, functions.py:947-952) or
    a blur matrix B -> PreblurA_of_H (A = B H, functions.py:999-1001).

    Returns a dict of arrays in MaxEntResult layout for a scalar run
    (maxent_result.py:835-967): alpha (X,) [= alpha*scale], v (X,n_s),
    H, A (X,n_omega), chi2, S, Q (X,), n_iter (X,), converged (X,).
    Returns None when max|G| < G_threshold (maxent_loop.py:174-179).
    """
    if opts is None:
        opts = LevenbergOptions()
    if np.max(np.abs(p.G)) < G_threshold:
        return None
    if scale_alpha is None:
        scale = 1.0
    elif isinstance(scale_alpha, str):
        assert scale_alpha.lower() == 'ndata'
        scale = len(p.G)
    else:
        scale = scale_alpha
    v = initial_v(p, delta, A_init)
    X = len(alpha_mesh)
    out = dict(alpha=np.empty(X), v=np.empty((X, len(v))),
               H=np.empty((X, len(p.D))), A=np.empty((X, len(p.D))),
               chi2=np.empty(X), S=np.empty(X), Q=np.empty(X),
               n_iter=np.zeros(X, dtype=int),
               converged=np.zeros(X, dtype=bool),
               n_evals=np.zeros(X, dtype=int))
    t0 = time.perf_counter()
    for ia, alpha in enumerate(alpha_mesh):
        a = alpha * scale
        ne0 = p.n_evals
        v, n_iter, conv = levenberg_minimize(p, a, v, opts)
        H = H_of_v(p, v)
        out['alpha'][ia] = a
        out['v'][ia] = v          # the CORRECT per-alpha v
        out['H'][ia] = H
        out['A'][ia] = H / delta if A_of_H is None else np.dot(A_of_H, H)
        out['chi2'][ia] = chi2_f(p, H)
        out['S'][ia] = S_f(p, H)
        out['Q'][ia] = 0.5 * out['chi2'][ia] * p.chi2_factor - a * out['S'][ia]
        out['n_iter'][ia] = n_iter
        out['converged'][ia] = conv
        out['n_evals'][ia] = p.n_evals - ne0
    if timing is not None:
        timing.append(time.perf_counter() - t0)
    return out


def log_probability(p, alpha, v):
    """probabilities.py:76-85 with the default measure / norm / prior."""
    H = H_of_v(p, v)
    ddQ = 0.5 * p.d2 * p.chi2_factor - alpha * S_dd(p, H)
    _, lp = np.linalg.slogdet(ddQ)
    lp = -0.5 * lp
    lp += 1 / 2.0 * np.linalg.slogdet(-S_dd(p, H))[1]
    lp += (len(H) / 2.0) * np.log(alpha)
    lp -= 0.5 * chi2_f(p, H) * p.chi2_factor - alpha * S_f(p, H)
    lp += -np.log(alpha)
    return lp


# --------------------------------------------------------------------------
#  convenience: build a Problem the way TauMaxEnt does
# --------------------------------------------------------------------------

def make_tau_problem(tau, omega, G, err, beta=None, D=None, entropy='normal',
                     form='maxent', preblur_b=None, threshold=1.e-14,
                     usv=None):
    """tau_maxent.py:56-66,181-251 + maxent_loop.py:184.

    ``usv``: optional (U,S,V) to use instead of this machine's LAPACK SVD
    (V is machine dependent for tiny singular values; reference
    test/python/maxent_result.py:26-35).
    Returns (Problem, delta, B or None, K_delta).
    """
    omega = np.asarray(omega, dtype=float)
    K, K_delta = tau_kernel(tau, omega, beta)
    B = None
    if preblur_b is not None:
        K, B = preblur_kernel(K, omega, preblur_b)
    if usv is None:
        U, S, V = svd_reduce(K, threshold)
    else:
        U, S, V = usv
    if D is None:
        D = flat_default_model(omega)
    p = Problem(K, U, S, V, G, err, D, entropy=entropy, form=form)
    return p, omega_delta(omega), B, K_delta
