"""The cost-function contract on the device: ``Q(v)``, ``f / d / dd`` in every
mode, the component functions, and a user-supplied minimiser running on them.

Restates the reference's own tests
    test/python/maxent_cost_function_d.py:27-58   (finite differences, all dA_projection modes)
    test/python/plus_minus_entropy.py:46-59       (closed form of the plus-minus entropy, 1e-14)
against fixtures made from the imported reference (tests/golden/make_golden.py: derivs_case,
plusminus_entropy_case).  Everything below goes through ``mxe_eval_batch`` / ``mxe_entropy``.
"""

import os

import numpy as np
import pytest

import maxent_amd as mx
from maxent_amd import device

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


@pytest.fixture(scope='module')
def case():
    z = np.load(os.path.join(GOLD, 'derivs.npz'))
    omega = mx.DataOmegaMesh(z['omega'])
    assert np.array_equal(omega.delta, z['delta'])
    K = mx.TauKernel(tau=z['tau'], omega=omega, beta=float(z['beta']))
    # the derivatives are written in the basis of the singular vectors: use the reference's own
    # decomposition (the vectors that belong to its 1e-17 singular values are not reproducible)
    K._U, K._S, K._V = z['U'], z['S'], z['V']
    D = mx.DataDefaultModel(z['D'] / omega.delta, omega)
    assert np.allclose(D.D, z['D'], rtol=1e-15)
    return z, omega, K, D


def make_Q(z, K, D, entropy='normal', **kw):
    chi2 = mx.NormalChi2(K=K, G=z['G'], err=z['err'])
    if entropy == 'normal':
        Q = mx.MaxEntCostFunction(chi2=chi2, S=mx.NormalEntropy(D=D), H_of_v=mx.NormalH_of_v(D=D, K=K), **kw)
    else:
        Q = mx.MaxEntCostFunction(chi2=chi2, S=mx.PlusMinusEntropy(D=D), H_of_v=mx.PlusMinusH_of_v(D=D, K=K), **kw)
    Q.set_alpha(float(z['alpha']))
    return Q


def test_every_mode_of_the_cost_function_matches_the_reference(case):
    z, omega, K, D = case
    v = z['v']
    Q = make_Q(z, K, D)
    for d_dv in (False, True):
        for proj in range(3):
            Q.d_dv, Q.dA_projection = d_dv, proj
            tag = 'n_ddv%d_p%d_' % (int(d_dv), proj)
            assert abs(Q.f(v) - z[tag + 'f']) <= 1e-12 * abs(z[tag + 'f'])
            assert rel(Q.d(v), z[tag + 'd']) < 1e-10, (tag, rel(Q.d(v), z[tag + 'd']))
            assert rel(Q.dd(v), z[tag + 'dd']) < 1e-10, (tag, rel(Q.dd(v), z[tag + 'dd']))
    # chi2_factor
    Qe = make_Q(z, K, D, chi2_factor=float(z['chi2_factor']))
    assert abs(Qe.f(v) - z['n_eta_f']) <= 1e-12 * abs(z['n_eta_f'])
    assert rel(Qe.d(v), z['n_eta_d']) < 1e-10 and rel(Qe.dd(v), z['n_eta_dd']) < 1e-10
    # Bryan's form
    Qb = mx.BryanCostFunction()
    Qb.chi2 = mx.NormalChi2(K=K, G=z['G'], err=z['err'])
    Qb.set_D(D)
    Qb.H_of_v.set_K(K)
    Qb.set_alpha(float(z['alpha']))
    assert abs(Qb.f(v) - z['b_f']) <= 1e-12 * abs(z['b_f'])
    assert rel(Qb.d(v), z['b_d']) < 1e-10 and rel(Qb.dd(v), z['b_dd']) < 1e-10
    with pytest.raises(NotImplementedError):
        Qb.H_of_v = mx.PlusMinusH_of_v(D=D, K=K)
    # plus-minus entropy pair
    Qp = make_Q(z, K, D, entropy='plusminus')
    for d_dv in (False, True):
        Qp.d_dv = d_dv
        tag = 'pm_ddv%d_p2_' % int(d_dv)
        assert abs(Qp.f(z['v_pm']) - z[tag + 'f']) <= 1e-12 * abs(z[tag + 'f'])
        assert rel(Qp.d(z['v_pm']), z[tag + 'd']) < 1e-10
        assert rel(Qp.dd(z['v_pm']), z[tag + 'dd']) < 1e-10


def test_pinned_point_exposes_the_component_functions(case):
    z, omega, K, D = case
    for pre, ent, v in (('n_', 'normal', z['v']), ('pm_', 'plusminus', z['v_pm'])):
        Q = make_Q(z, K, D, entropy=ent)
        b = Q(v)
        assert b.f() == Q.f(v)
        H = b.H_of_v.f()
        assert rel(H, z[pre + 'H']) < 1e-13
        assert rel(b.H_of_v.d(), z[pre + 'dH_dv']) < 1e-13
        assert abs(b.chi2.f() - z[pre + 'chi2']) <= 1e-12 * z[pre + 'chi2']
        assert abs(b.S.f() - z[pre + 'S']) <= 1e-12 * abs(z[pre + 'S'])
        assert rel(b.S.d(), z[pre + 'dS_dH']) < 1e-12
        assert rel(np.diag(b.S.dd()), z[pre + 'ddS_diag']) < 1e-12
        assert rel(b.chi2.d(H), z[pre + 'dchi2_dH']) < 1e-9
        assert rel(b.A_of_H.f(), H / omega.delta) < 1e-15
        # the blocks on their own, as functions of a hidden image
        assert abs(Q.chi2.f(H) - z[pre + 'chi2']) <= 1e-11 * z[pre + 'chi2']
        assert abs(Q.S.f(H) - z[pre + 'S']) <= 1e-12 * abs(z[pre + 'S'])
        assert rel(Q.S.d(H), z[pre + 'dS_dH']) < 1e-12
        assert rel(Q.H_of_v.f(v), z[pre + 'H']) < 1e-13
        assert rel(Q.H_of_v.inv(H), z[pre + 'v_of_H']) < 1e-9


def test_finite_differences_like_the_reference_test(case):
    """maxent_cost_function_d.py:51-58 with the same calls"""
    z, omega, K, D = case
    # the reference test's own random numbers (maxent_cost_function_d.py:25,35,49-50): noise, v, random_A
    rng = np.random.RandomState(658436166)
    rng.randn(len(z['G']))
    rng.rand(len(z['S']))
    random_A = rng.rand(len(omega))
    Q = make_Q(z, K, D)
    assert Q.chi2.check_derivatives(random_A, Q.chi2.f(random_A), prec=1.e-8)
    assert Q.S.check_derivatives(random_A, prec=1.e-5)
    v = z['v'][:]
    for Q.d_dv in [True]:
        for Q.dA_projection in range(3):
            assert Q.check_derivatives(v, Q.f(v), prec=1.e-8)
    # (Bryan's d is W^-1 times the gradient of f, dA_projection = 1 likewise: no finite-difference twin;
    #  they are compared with the reference's values in the test above)


def test_plus_minus_entropy_closed_form():
    """plus_minus_entropy.py:46-59: value, gradient and curvature against the closed form, 1e-14"""
    z = np.load(os.path.join(GOLD, 'plusminus_entropy.npz'))
    w = mx.LinearOmegaMesh(-10, 10, 101)
    D = mx.DataDefaultModel(0 * w + 0.9, w)
    assert np.array_equal(D.D, z['D'])
    A, Dd = z['A'], z['D']
    r = np.sqrt(A ** 2 + 4 * Dd ** 2)
    Ap, Am = (r + A) / 2, (r - A) / 2
    closed_f = np.sum(r - 2 * Dd - Ap * np.log(Ap / Dd) - Am * np.log(Am / Dd))
    closed_d = -(A / (2 * r) + 0.5) * np.log(Ap / Dd) - (A / (2 * r) - 0.5) * np.log(Am / Dd)
    closed_dd = -1.0 / r
    S = mx.PlusMinusEntropy(D=D)(A)
    # a sum of 101 terms of size ~1: the device's reduction tree and numpy's differ by a few ulp of |S|
    assert abs(S.f() - closed_f) < 5e-14 and abs(S.f() - z['f']) < 5e-14
    assert np.max(np.abs(S.d() - closed_d)) < 1e-14 and np.max(np.abs(S.d() - z['d'])) < 1e-14
    assert np.max(np.abs(np.diag(S.dd()) - closed_dd)) < 1e-14
    assert np.max(np.abs(np.diag(S.dd()) - z['dd_diag'])) < 1e-14
    Sn = mx.NormalEntropy(D=D)(A)
    assert abs(Sn.f() - z['normal_f']) < 5e-14
    assert np.max(np.abs(Sn.d() - z['normal_d'])) < 1e-14
    assert np.max(np.abs(np.diag(Sn.dd()) - z['normal_dd_diag'])) < 1e-13


class PlainNewton(mx.Minimizer):
    """a user's minimiser: damped Newton on Q.d / Q.dd, nothing of the device solver in it"""

    def __init__(self):
        self.n_iter_last, self.converged = 0, False

    def minimize(self, function, v0):
        v = np.array(v0, dtype=float)
        self.converged = False
        for it in range(200):
            g, J = function.d(v), function.dd(v)
            step = np.linalg.solve(J, g)
            t, f0 = 1.0, function.f(v)
            while t > 1e-6 and not function.f(v - t * step) <= f0:
                t *= 0.5
            v = v - t * step
            if np.linalg.norm(t * step) < 1e-10 * max(1.0, np.linalg.norm(v)):
                self.converged = True
                break
        self.n_iter_last = it + 1
        return v


def test_user_supplied_minimizer_runs_on_the_device_cost_function():
    from maxent_amd import synthetic
    tau, omega, K, G = synthetic.single_G(40, 80)
    runs = {}
    for name, minimizer in (('device', None), ('user', PlainNewton())):
        # dA_projection = 1: d = g and dd = M W + alpha 1 is its Jacobian, a well-conditioned Newton system
        tm = mx.TauMaxEnt(cost_function=mx.MaxEntCostFunction(dA_projection=1),
                          **({} if minimizer is None else dict(minimizer=minimizer)))
        tm.set_verbosity(mx.VerbosityFlags.Quiet)
        tm.omega = omega
        tm.set_G_tau_data(tau, G)
        tm.set_error(synthetic.SIGMA)
        tm.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-1, alpha_max=1e3, n_points=5)
        runs[name] = tm.run()
    e = np.linalg.norm(runs['user'].H - runs['device'].H, axis=1) / np.linalg.norm(runs['device'].H, axis=1)
    assert e.max() < 1e-6, e
    np.testing.assert_allclose(runs['user'].chi2, runs['device'].chi2, rtol=1e-6)
    assert runs['user'].analyzer_results['LineFitAnalyzer']['alpha_index'] == \
        runs['device'].analyzer_results['LineFitAnalyzer']['alpha_index']


def test_chi2_factor_reaches_the_solver(case):
    """Q = eta chi2 / 2 - alpha S: the scan with chi2_factor = eta equals the scan at alpha / eta, Q scaled"""
    from maxent_amd import synthetic
    tau, omega, K, G = synthetic.single_G(40, 80)
    out = {}
    for eta in (1.0, 2.5):
        tm = mx.TauMaxEnt(cost_function=mx.MaxEntCostFunction(chi2_factor=eta))
        tm.set_verbosity(mx.VerbosityFlags.Quiet)
        tm.omega = omega
        tm.set_G_tau_data(tau, G)
        tm.set_error(synthetic.SIGMA)
        tm.alpha_mesh = mx.DataAlphaMesh(np.array([1.0, 10.0, 100.0]) * eta)
        out[eta] = tm.run()
    e = np.linalg.norm(out[2.5].H - out[1.0].H, axis=1) / np.linalg.norm(out[1.0].H, axis=1)
    assert e.max() < 1e-8
    np.testing.assert_allclose(out[2.5].Q, 2.5 * out[1.0].Q, rtol=1e-9)


def test_chi2_and_entropy_do_not_depend_on_the_omega_grid():
    """reference test/python/functions_different_grid.py:27-60: the same A(omega) on a hyperbolic and on a linear
    mesh of 1000 points gives the same chi2 and S (device evaluation at a caller-supplied H, n_omega = 1000)"""
    beta = 40
    tau = np.linspace(0, beta, 100)
    omega = mx.LinearOmegaMesh(omega_min=-10, omega_max=10, n_points=100)
    K = mx.TauKernel(tau=tau, omega=omega, beta=beta)
    A = np.exp(-np.asarray(omega) ** 2)
    A /= np.trapezoid(A, np.asarray(omega))
    G = np.dot(K.K_delta, A)
    err = np.ones(len(G))
    grids = [mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=1000),
             mx.LinearOmegaMesh(omega_min=-10, omega_max=10, n_points=1000)]
    for test_A in (lambda w: w / w, lambda w: np.exp(-w ** 2)):
        c2, s = [], []
        for om in grids:
            Kg = mx.TauKernel(tau=tau, omega=om, beta=beta)
            Ag = test_A(np.asarray(om))
            Ag /= np.trapezoid(Ag, np.asarray(om))
            D = mx.FlatDefaultModel(omega=om)
            chi2 = mx.NormalChi2(K=Kg, G=G, err=err)
            S = mx.NormalEntropy(D=D)
            c2.append(chi2.f(Ag * om.delta))
            s.append(S.f(Ag * om.delta))
            # and against plain numpy on the full kernel
            assert abs(c2[-1] - np.sum((np.dot(Kg.K, Ag * om.delta) - G) ** 2)) < 1e-9 * max(1.0, c2[-1])
        assert abs(c2[1] - c2[0]) < 1.e-4, 'chi2 not equal'
        assert abs(s[1] - s[0]) < 1.e-4, 'S not equal'


def test_result_filled_alpha_by_alpha_has_the_reference_shapes():
    """reference test/python/matrix_maxent_result.py:26-140: MaxEntResult.add_result(Q(v)) -- one alpha at a time,
    with and without matrix structure -- and the shapes of everything that comes out"""
    rng = np.random.RandomState(658436166)
    beta = 40
    tau = np.linspace(0, beta, 100)
    omega = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=100)
    K = mx.TauKernel(tau=tau, omega=omega, beta=beta)
    K.reduce_singular_space()
    A = np.exp(-np.asarray(omega) ** 2)
    A /= np.trapezoid(A, np.asarray(omega))
    G = np.dot(K.K, A) + 1.e-4 * rng.randn(len(tau))
    err = 1.e-4 * np.ones(len(G))
    D = mx.FlatDefaultModel(omega=omega)
    Q = mx.MaxEntCostFunction(chi2=mx.NormalChi2(K=K, G=G, err=err), S=mx.NormalEntropy(D=D),
                              H_of_v=mx.NormalH_of_v(D=D, K=K), A_of_H=mx.IdentityA_of_H(omega=D.omega))
    Q.set_alpha(0.1)
    ns = len(K.S)
    v1, v2 = rng.rand(ns), rng.rand(ns)

    mr = mx.MaxEntResult()
    mr.add_result(Q(v1))
    with pytest.raises(AssertionError):
        mr.add_result(Q(v1), matrix_element=(1, 1))
    assert mr.alpha == [0.1] and mr.alpha.shape == (1,) and mr._n_alphas == 1
    for name in ('chi2', 'S', 'Q'):
        assert getattr(mr, name).shape == (1,)
    assert mr.A.shape == (1, 100) and mr.v.shape == (1, ns) and mr.run_times.shape == (1,)
    at = Q(v1)
    assert mr.chi2[0] == at.chi2.f() and mr.Q[0] == at.f()
    np.testing.assert_array_equal(mr.A[0], at.H_of_v.f() / omega.delta)

    mr = mx.MaxEntResult(matrix_structure=(2, 2))
    with pytest.raises(AssertionError):
        mr.add_result(Q(v1))
    assert mr._get_empty(fill_with=lambda: 1) == [[1, 1], [1, 1]]
    mr.add_result(Q(v1), matrix_element=(1, 1))
    mr.add_result(Q(v2), matrix_element=(1, 1))
    assert np.all(mr.alpha == 0.1) and mr.alpha.shape == (2,)
    assert np.all(mr._n_alphas == [[0, 0], [0, 2]])
    for name in ('chi2', 'S', 'Q'):
        assert getattr(mr, name).shape == (2, 2, 2)
    assert mr.A.shape == (2, 2, 2, 100) and mr.v.shape == (2, 2, 2, ns)
    assert mr.run_times.shape == (2, 2, 2) and mr.omega.shape == (100,)

    mr = mx.MaxEntResult(matrix_structure=(2, 2))
    for elem, v in (((0, 0), v1), ((1, 1), v2), ((0, 1), v1), ((1, 0), v2)):
        mr.add_result(Q(v), matrix_element=elem)
    assert mr.alpha == [0.1]
    for name in ('chi2', 'S', 'Q'):
        assert getattr(mr, name).shape == (2, 2, 1)
    assert mr.A.shape == (2, 2, 1, 100) and mr.v.shape == (2, 2, 1, ns) and mr.omega.shape == (100,)
    assert mr.run_times.shape == (2, 2, 1) and mr.run_time_total.shape == (2, 2)
    assert mr.chi2[0, 1, 0] == mr.chi2[0, 0, 0] and mr.chi2[1, 0, 0] == mr.chi2[1, 1, 0]


def test_pinned_cost_function_evaluates_once(case):
    """reference test/python/maxent_cost_function_lazy.py:66-113 restated: the reference traces its Python calls and asserts
    that a cost function pinned to v -- ``me1 = me(v)`` -- evaluates every component function ONCE however many of
    ``d / dH / ddH / dd / f / chi2.f / H_of_v.f / A_of_H.f`` are asked for, while the same method called twice on the
    unpinned object evaluates twice.  Here the evaluations are launches of the device evaluation kernel
    (``DeviceContext.eval_batch``), counted."""
    z, omega, K, D = case
    v = z['v']
    Q = make_Q(z, K, D)
    calls = []
    orig = device.DeviceContext.eval_batch

    def counting(self, *a, **k):
        calls.append(k.get('want', a[5] if len(a) > 5 else None))
        return orig(self, *a, **k)
    device.DeviceContext.eval_batch = counting
    try:
        Q.dH(v)
        n_first = len(calls)
        Q.dH(v)
        assert n_first >= 1 and len(calls) == 2 * n_first        # unpinned: every call evaluates (first block of the reference test)
        calls.clear()
        me1 = Q(v)
        n_pin = len(calls)
        assert n_pin == 1                                        # pinning evaluates H (eagerly, like the reference)
        for rep in range(2):
            me1.d(); me1.dH(); me1.ddH(); me1.dd(); me1.f()
            me1.chi2.f(); me1.H_of_v.f(); me1.A_of_H.f(); me1.S.f(); me1.S.d(); me1.H_of_v.d()
            if rep == 0:
                n_block = len(calls)
        assert len(calls) == n_block                             # the second round of the block: nothing evaluated again
        assert n_block <= 2                                      # one launch serves f, d, dH, ddH, dd and the components
    finally:
        device.DeviceContext.eval_batch = orig
