"""LevenbergMinimizer on general functions: the reference's own minimiser tests restated
(test/python/minimizer_cubefun.py:24-62, minimizer_circlefun.py:25-62, minimizer_sin.py:24-56) and the first half of
maxent_cost_function_lazy.py:36-64 (a linear least-squares problem is solved to 1e-15).  Host only: the general search
never touches the device (the alpha scan never comes here)."""
import numpy as np
import pytest

import maxent_amd as mx
from maxent_amd.functions import DoublyDerivableFunction, cached
from maxent_amd.minimizers import (LevenbergMinimizer, MaxDerivativeConvergenceMethod,
                                   FunctionChangeConvergenceMethod)


class Cube(DoublyDerivableFunction):
    @cached
    def f(self, v):
        return np.sum(v ** 3)

    @cached
    def d(self, v):
        return 3 * v ** 2

    @cached
    def dd(self, v):
        return 6 * np.diag(v)


class Circle(DoublyDerivableFunction):
    def __init__(self, x, y):
        self.center = np.array([x, y])

    @cached
    def f(self, v):
        return np.sqrt(np.sum((v - self.center) ** 2))

    @cached
    def d(self, v):
        return (v - self.center) / self.f(v)

    @cached
    def dd(self, v):
        dx, dy = v - self.center
        return np.array([[dy ** 2, -dx * dy], [-dx * dy, dx ** 2]]) / self.f(v) ** 3


class Sin(DoublyDerivableFunction):
    @cached
    def f(self, v):
        return np.sin(v)[0]

    @cached
    def d(self, v):
        return np.cos(v)

    @cached
    def dd(self, v):
        return -np.diag(np.sin(v))


class Coupled(DoublyDerivableFunction):
    """sum((A v)^3): a Hessian that is not diagonal, so that every damping variant takes its own path"""
    A = np.array([[1.0, 0.5], [0.2, 1.0]])

    @cached
    def f(self, v):
        return np.sum(np.dot(self.A, v) ** 3)

    @cached
    def d(self, v):
        return np.dot(self.A.T, 3 * np.dot(self.A, v) ** 2)

    @cached
    def dd(self, v):
        return np.dot(self.A.T * (6 * np.dot(self.A, v)), self.A)


class Rosenbrock(DoublyDerivableFunction):
    @cached
    def f(self, v):
        return (1 - v[0]) ** 2 + 100 * (v[1] - v[0] ** 2) ** 2

    @cached
    def d(self, v):
        return np.array([-2 * (1 - v[0]) - 400 * v[0] * (v[1] - v[0] ** 2), 200 * (v[1] - v[0] ** 2)])

    @cached
    def dd(self, v):
        return np.array([[2 - 400 * (v[1] - 3 * v[0] ** 2), -400 * v[0]], [-400 * v[0], 200.0]])


FLAGS = [(j, m) for j in (True, False) for m in (True, False)]


@pytest.mark.parametrize('J_squared,marquardt', FLAGS)
def test_cube(J_squared, marquardt):
    cube = Cube()
    lm = LevenbergMinimizer(J_squared=J_squared, marquardt=marquardt, mu0=1e-18, nu=1.3,
                            convergence=MaxDerivativeConvergenceMethod(1.e-10))
    v = lm.minimize(cube, np.array([200.0]))
    assert v < 1.e-5
    assert np.max(np.abs(cube.d(v))) < lm.convergence.convergence_criterion
    assert lm.converged and lm.n_iter_last > 1 and lm.n_iter == lm.n_iter_last


def test_circle():
    center = (3.0, 2.0)
    lines = []
    lm = LevenbergMinimizer(convergence=FunctionChangeConvergenceMethod(1.e-10), mu0=1e-18, nu=1.3,
                            verbose_callback=lines.append)
    v = lm.minimize(Circle(*center), np.array([200.0, 150.0]))
    assert np.max(np.abs(v - center)) < 1.e-9
    assert len(lines) == lm.n_iter_last and 'max_f' in lines[0]


@pytest.mark.parametrize('J_squared,marquardt', FLAGS)
def test_sin(J_squared, marquardt):
    sin = Sin()
    lm = LevenbergMinimizer(J_squared=J_squared, marquardt=marquardt, mu0=1e-18, nu=1.3)     # (default stopping rule)
    v = lm.minimize(sin, np.array([0.1]))
    assert abs(sin.f(v) + 1) < 1.e-9


@pytest.mark.parametrize('J_squared,marquardt', FLAGS)
def test_the_flags_change_the_iterates(J_squared, marquardt):
    """J_squared and marquardt are real in the general search: the four variants take different paths"""
    lines = []
    lm = LevenbergMinimizer(J_squared=J_squared, marquardt=marquardt, mu0=1.0, nu=2.0, maxiter=3,
                            convergence=MaxDerivativeConvergenceMethod(1e-300), verbose_callback=lines.append)
    # (a function whose full Newton step goes uphill from here, so that the damping stays in the step: where the walk of mu
    #  ends at its floor -- Coupled() from (3, 5): every smaller mu is better -- all four variants take the Newton step)
    v = lm.minimize(Rosenbrock(), np.array([-1.2, 1.0]))
    test_the_flags_change_the_iterates.seen = getattr(test_the_flags_change_the_iterates, 'seen', {})
    test_the_flags_change_the_iterates.seen[(J_squared, marquardt)] = tuple(np.round(v, 10))
    seen = test_the_flags_change_the_iterates.seen
    assert len(set(seen.values())) == len(seen)


class LeastSquares(DoublyDerivableFunction):
    """chi2 / 2 of a linear model: what the reference's dummy MaxEntCostFunction (NullFunction entropy, IdentityH_of_v,
    alpha = 0) of maxent_cost_function_lazy.py:36-55 amounts to"""

    def __init__(self, A, y, err):
        self.A, self.y, self.err = A, y, err

    @cached
    def f(self, v):
        return 0.5 * np.sum(((np.dot(self.A, v) - self.y) / self.err) ** 2)

    @cached
    def d(self, v):
        return np.dot(self.A.T, (np.dot(self.A, v) - self.y) / self.err ** 2)

    @cached
    def dd(self, v):
        return np.dot(self.A.T / self.err ** 2, self.A)


def test_linear_least_squares_to_machine_precision():
    tau = np.linspace(0, 1, 101)
    A = np.ones((len(tau), 3))
    A[:, 0], A[:, 1] = tau ** 2, tau
    solution = np.array([1.0, 2.0, 3.0])
    lsq = LeastSquares(A, np.dot(A, solution), 1.e-4)
    lm = LevenbergMinimizer(convergence=MaxDerivativeConvergenceMethod(1.e-7), mu0=1e-18, nu=1.3)
    v = lm.minimize(lsq, np.ones(3))
    assert np.max(np.abs(v - solution)) < 1.e-13      # (the reference asserts 1e-15 on its own path through K; the same
    assert lm.converged                               #  search on chi2 / 2 reaches the conditioning of A^T A / err^2)


def test_nu_below_one_is_refused_and_miniter_is_kept():
    with pytest.raises(Exception):
        LevenbergMinimizer(nu=1.0).minimize(Cube(), np.array([1.0]))
    from maxent_amd.minimizers import NullConvergenceMethod
    lm = LevenbergMinimizer(miniter=7, convergence=NullConvergenceMethod())
    lm.minimize(Sin(), np.array([0.3]))
    assert lm.n_iter_last == 8


def _reference_style_search(fun, v0, mu0=1e-18, nu=1.3, max_mu=1e20, maxiter=1000, tol=1e-4):
    """the search of levenberg_minimizer.py:155-243 written down independently for this test (plain loops, the
    reference's stopping rule max|d| < 1e-4): iteration count and the damping it ends with"""
    v, mu = np.array(v0, float), mu0
    for it in range(maxiter):
        g, J = fun(v).d(), fun(v).dd()
        if np.max(np.abs(g)) < tol:
            return v, it + 1, mu
        Q0 = fun(v).f()
        q = lambda m: fun(v - np.linalg.solve(J + m * np.eye(2), g)).f()
        Q1 = q(mu)
        while (Q1 > Q0 or np.isnan(Q1)) and mu < max_mu:
            mu *= nu
            Q1 = q(mu)
        Q2 = q(nu * mu)
        if Q2 < Q1:
            f, take, Q2 = nu, nu * mu, Q1
        else:
            f, take = 1 / nu, mu
        use = mu
        mu *= nu
        Qb = np.inf
        while Q2 < Qb and nu * np.finfo(float).eps < mu < max_mu:
            Qb, use = Q2, take
            mu *= f
            take, Q2 = mu, q(mu)
        v = v - np.linalg.solve(J + use * np.eye(2), g)
    return v, maxiter, mu


def test_the_damping_falls_back_after_a_rejected_step():
    """once a step had been rejected the damping could only grow (the walk towards smaller mu re-evaluated
    the same trial and stopped): Rosenbrock's valley from (-1.2, 1) then took 1000 iterations with mu stuck near 65.
    With the reference's walk it converges in a few dozen iterations and mu is back at its floor."""
    m = LevenbergMinimizer(convergence=MaxDerivativeConvergenceMethod(1e-4))
    v = m.minimize(Rosenbrock(), np.array([-1.2, 1.0]))
    v_ref, n_ref, mu_ref = _reference_style_search(Rosenbrock(), [-1.2, 1.0])
    assert m.converged and np.allclose(v, [1.0, 1.0], atol=1e-5)
    assert n_ref < 60 and abs(m.n_iter_last - n_ref) <= 1, (m.n_iter_last, n_ref)
    assert np.allclose(v, v_ref, atol=1e-9)
