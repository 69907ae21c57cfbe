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
    v = lm.minimize(Coupled(), np.array([3.0, 5.0]))
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
