"""Meshes and default models against the reference's own expected files and assertions (CPU).

Restated from reference test/python/omega_meshes.py, alpha_meshes.py, default_models.py; the ``.ref`` files under
tests/golden/ are the reference test suite's data files (copied by tests/golden/make_golden.py: logtaker_case)."""
import copy
import os

import numpy as np
import pytest

import maxent_amd as mx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _ref(name):
    with open(os.path.join(GOLD, name)) as f:
        return f.read()


def _listing(name, mesh):
    return name + '\n' + ''.join('%.8f\n' % w for w in mesh) + '-' * 80 + '\n'


def _old_delta(v):
    # the formula the reference's test keeps for comparison (omega_meshes.py:28-46)
    v = np.asarray(v)
    d = np.empty(len(v))
    d[1:-1] = (v[2:] - v[:-2]) / 2.0
    d[0] = (v[1] - v[0]) / 2.0
    d[-1] = (v[-1] - v[-2]) / 2.0
    return d


def test_omega_meshes_match_the_reference_listing():
    out = ''
    rng = np.random.RandomState(0)
    for cls in (mx.LinearOmegaMesh, mx.LorentzianOmegaMesh, mx.LorentzianSmallerOmegaMesh, mx.HyperbolicOmegaMesh):
        m = cls(omega_min=-10, omega_max=10, n_points=10)
        out += _listing(cls.__name__, m)
        assert np.max(np.abs(_old_delta(m) - m.delta)) < 1.e-15
        func = rng.rand(len(m))
        assert abs(np.trapezoid(func, np.asarray(m)) - np.sum(func * m.delta)) < 1.e-14
        for other in (m.copy(), copy.deepcopy(m)):
            assert np.all(other == m)
            assert (other.omega_min, other.omega_max, other.n_points) == (m.omega_min, m.omega_max, m.n_points)
    assert out == _ref('omega_meshes.ref')


def test_alpha_meshes_match_the_reference_listing():
    out = ''
    for cls in (mx.LinearAlphaMesh, mx.LogAlphaMesh):
        out += _listing(cls.__name__, cls(alpha_min=1.e-4, alpha_max=1.e2, n_points=10))
        with pytest.raises(Exception):
            cls(alpha_min=2, alpha_max=1)
        with pytest.raises(Exception):
            cls(alpha_min=-1, alpha_max=2)
    out += _listing('DataAlphaMesh', mx.DataAlphaMesh(np.linspace(1.e-3, 100, 10)))
    with pytest.raises(Exception):
        mx.DataAlphaMesh(np.linspace(-1, 10))
    assert out == _ref('alpha_meshes.ref')


def test_default_models_follow_the_reference_assertions():
    # reference test/python/default_models.py:25-62
    w = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=100)
    D1 = mx.FlatDefaultModel(omega=w)
    assert np.all(np.abs(D1.D / w.delta - 1.0 / 20.0) < 1.e-15)
    assert np.sum(D1.D) - 1.0 < 2.e-15
    D2 = mx.DataDefaultModel(D1.D / w.delta, w)
    assert np.all(D1.D == D2.D)
    w_lin = mx.LinearOmegaMesh(omega_min=-10, omega_max=10, n_points=50)
    D3 = mx.DataDefaultModel(D1.D / w.delta, w, w_lin)
    assert len(D3) == 50
    assert np.all(np.abs(D3.D / w_lin.delta - 1.0 / 20.0) < 1.e-15)
    w_2p = mx.DataOmegaMesh([-10, 10])
    D4 = mx.DataDefaultModel([-10, 10], w_2p, w_lin)
    assert np.all(np.abs(D4.D / w_lin.delta - np.linspace(-10, 10, 50)) < 1.e-15)
    D5 = mx.DataDefaultModel([-9, 11], w_2p, w)
    assert abs(np.sum(D5.D) - 20.0) < 1.e-13          # the trapezoidal rule is exact for a linear function
    D5.omega = w_2p
    assert len(D5.D) == len(w)                          # nothing changes before parameter_change()
    D5.parameter_change()
    assert len(D5.D) == 2 and abs(np.sum(D5.D) - 20.0) < 1.e-13


def test_tau_kernel_follows_the_reference_assertions():
    # reference test/python/tau_kernel.py:26-85
    def old_kernel(tau, omega, beta):
        oomega, ttau = np.meshgrid(omega, tau)
        K = np.empty(oomega.shape)
        L = oomega >= 0.0
        iL, nL = np.where(L), np.where(np.logical_not(L))
        K[iL] = -np.exp(-oomega[iL] * ttau[iL]) / (np.exp(-beta * oomega[iL]) + 1.0)
        K[nL] = -np.exp(oomega[nL] * (beta - ttau[nL])) / (1.0 + np.exp(beta * oomega[nL]))
        return K
    rng = np.random.RandomState(12)
    tau = 10 * rng.rand(10)
    omega = mx.DataOmegaMesh(rng.rand(20))
    beta = 10.0
    K1 = mx.TauKernel(tau=tau, omega=omega, beta=beta)
    K2 = old_kernel(tau, np.asarray(omega), beta)
    assert np.max(np.abs(K1.K - K2)) < 1.e-15
    assert np.max(np.abs(K1.K - np.dot(K1.U, np.dot(np.diag(K1.S), K1.V.transpose())))) < 1.e-13
    L1 = len(K1.S)
    K1.reduce_singular_space(np.median(K1.S))
    assert len(K1.S) == L1 // 2
    K3 = mx.TauKernel(tau=tau, omega=omega, beta=beta)
    K3.omega = omega[::2]
    assert np.max(np.abs(K2 - K3.K)) < 1.e-15            # nothing changes before parameter_change()
    K3.parameter_change()
    assert K3.K.shape == (10, 10)
    assert np.max(np.abs(K3.K - old_kernel(tau, np.asarray(omega)[::2], beta))) < 1.e-15


def test_user_written_mesh_and_default_model_as_in_the_customization_guide():
    """reference doc/guide/customization.rst: a mesh class with its own __init__ on top of BaseOmegaMesh, a default
    model that fills ``_D`` in ``_fill_values``, both handed to TauMaxEnt"""
    from maxent_amd.omega_meshes import BaseOmegaMesh
    from maxent_amd.default_models import BaseDefaultModel

    class MyLogOmegaMesh(BaseOmegaMesh):
        def __init__(self, order_min=-5, order_max=1, n_points=100):
            super(MyLogOmegaMesh, self).__init__(omega_min=-10 ** order_max, omega_max=10 ** order_max, n_points=n_points)
            if n_points % 2 != 0:
                raise Exception('MyLogOmegaMesh needs an even number of n_points.')
            mesh_p = -np.logspace(order_min, order_max, n_points // 2)
            self[:] = np.append(mesh_p[::-1], -mesh_p)

    class MyGaussianDefaultModel(BaseDefaultModel):
        def __init__(self, omega, sigma=0.5):
            super(MyGaussianDefaultModel, self).__init__(omega)
            self.sigma = sigma
            self._fill_values()

        def _fill_values(self):
            self._D = 1.0 / np.sqrt(2.0 * np.pi * self.sigma ** 2) * np.exp(-self.omega ** 2 / (2.0 * self.sigma ** 2))

    m = MyLogOmegaMesh(order_max=1, n_points=400)
    assert m.shape == (400,) and m[0] == -10.0 and m[-1] == 10.0 and (m.omega_min, m.omega_max, m.n_points) == (-10, 10, 400)
    assert np.all(np.diff(np.asarray(m)) > 0) and m.delta.shape == (400,)
    assert isinstance(m[::2], MyLogOmegaMesh) and m[::2].omega_max == 10
    back = copy.deepcopy(m)                    # (a class defined inside a function cannot be pickled by name)
    assert isinstance(back, MyLogOmegaMesh) and np.array_equal(back, m) and back.n_points == 400
    with pytest.raises(Exception):
        MyLogOmegaMesh(n_points=3)
    tm = mx.TauMaxEnt()
    tm.omega = m
    tm.D = MyGaussianDefaultModel(tm.omega)
    assert tm.D.D.shape == (400,) and tm.maxent_loop.D is tm.D
    tm.D.sigma = 1.0
    before = tm.D.D.copy()
    tm.D.parameter_change()
    assert not np.array_equal(before, tm.D.D)
