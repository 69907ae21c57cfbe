"""CPU tests: the oracle against the committed golden vectors.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py
from the REAL reference (/root/reference, imported in the build container).
These tests pin oracle/ref_numpy.py (step-faithful port) to the reference's
outputs, and oracle/sform.py (numpy model of the HIP kernel) to the
extended-precision truth.
"""
import os

import numpy as np
import pytest

from oracle import ref_numpy as R, sform as SF, hp_truth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    with np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


def rel_l2(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


def problem_from(g, K=None):
    if K is None:
        K, _ = R.tau_kernel(g['tau'], g['omega'], float(g['beta']) if 'beta' in g else None)
        if 'B' in g:
            K = np.dot(K, g['B'] * g['delta'][:, np.newaxis])
    return R.Problem(K, g['U'], g['S'], g['V'], g['G'], g['err'], g['D'],
                     entropy=str(g['entropy']), form=str(g['form']))


SINGLE = ['cfg1_normal', 'cfg1_bryan', 'cfg1_plusminus', 'cfg1_tauerr', 'cfg5_preblur_pm']


@pytest.mark.parametrize('name', SINGLE)
def test_port_reproduces_reference_exactly(name):
    """identical per-alpha iteration counts; chi2/S/Q/H to 1e-12."""
    g = load(name)
    p = problem_from(g)
    n_tau = len(g['tau'])
    out = R.alpha_loop(p, g['delta'], g['alpha'] / n_tau,
                       A_of_H=g['B'] if 'B' in g else None)
    assert list(out['n_iter']) == list(g['n_iter_ref'])
    assert list(out['converged']) == list(g['converged_ref'])
    for k in ('chi2', 'S', 'Q'):
        np.testing.assert_allclose(out[k], g[k + '_ref'], rtol=1e-12)
    rows = g['rows']
    np.testing.assert_allclose(out['H'][rows], g['H_ref'], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(out['A'][rows], g['A_ref'], rtol=1e-10, atol=1e-300)


def test_port_kernel_fill_and_meshes():
    g = load('cfg1_normal')
    w = R.hyperbolic_omega_mesh(-10, 10, 200)
    np.testing.assert_array_equal(w, g['omega'])
    np.testing.assert_array_equal(R.omega_delta(w), g['delta'])
    np.testing.assert_array_equal(R.flat_default_model(w), g['D'])
    np.testing.assert_allclose(R.log_alpha_mesh(1e-2, 1e4, 20) * 100, g['alpha'], rtol=1e-15)
    K, _ = R.tau_kernel(g['tau'], w, 40.0)
    # reference test/python/tau_kernel.py:66-69: U S V^T reconstructs K
    assert np.max(np.abs(np.dot(g['U'] * g['S'], g['V'].T) - K)) < 1e-13


def test_port_known_answer_log_probability():
    """reference test/python/tau_maxent.py:134-135."""
    g = load('kat_tau_maxent')
    K, _ = R.tau_kernel(g['tau'], g['omega'], None)
    p = problem_from(g, K)
    out = R.alpha_loop(p, g['delta'], g['alpha'] / len(g['tau']))
    assert list(out['n_iter']) == list(g['n_iter_ref'])
    lp = [R.log_probability(p, a, v) for a, v in zip(out['alpha'], out['v'])]
    np.testing.assert_almost_equal(lp, g['probability_kat'], 6)
    np.testing.assert_allclose(lp, g['probability_ref'], rtol=1e-10)


def test_port_huge_alpha_reproduces_default_model():
    """reference test/python/huge_alpha.py:49-50."""
    g = load('kat_huge_alpha')
    K, _ = R.tau_kernel(g['tau'], g['omega'], None)
    p = problem_from(g, K)
    out = R.alpha_loop(p, g['delta'], g['alpha'] / len(g['tau']))
    assert np.max(out['H'] - g['D']) < 1e-6
    np.testing.assert_allclose(out['H'], g['H_ref'], rtol=1e-10)


def test_port_srvo3_bryan_matches_alps():
    """reference test/python/srvo3_mesh_and_ALPS.py:100 (A equals ALPS maxspec
    to 2 decimals)."""
    g = load('kat_srvo3')
    w = R.lorentzian_omega_mesh(-15, 15, 500)
    np.testing.assert_allclose(w, g['omega'], rtol=0, atol=1e-13)
    K, _ = R.tau_kernel(g['tau'], g['omega'], None)
    p = problem_from(g, K)
    out = R.alpha_loop(p, g['delta'], g['alpha'] / len(g['tau']))
    assert list(out['n_iter']) == list(g['n_iter_ref'])
    ms = g['alps_maxspec']
    assert np.max(np.abs(out['A'][1] - np.interp(g['omega'], ms[:, 0], ms[:, 1]))) < 1e-2
    np.testing.assert_allclose(out['A'], g['A_ref'], rtol=1e-9, atol=1e-300)


def test_port_cov_rotated_problem():
    g = load('cov')
    p = R.Problem(g['K_rot'], g['U_rot'], g['S'], g['V'], g['G_rot'], g['err_rot'], g['D'])
    out = R.alpha_loop(p, g['delta'], g['alpha'] / len(g['G_rot']), scale_alpha=len(g['G_rot']))
    assert list(out['n_iter']) == list(g['n_iter_ref'])
    np.testing.assert_allclose(out['chi2'], g['chi2_ref'], rtol=1e-12)


@pytest.mark.parametrize('name', SINGLE + ['cfg2_normal', 'kat_tau_maxent', 'kat_srvo3'])
def test_reference_is_within_its_own_tolerance_of_truth(name):
    """documents the reference's accuracy floor: default tolerances leave H up
    to ~3e-5 (rel. L2) from the extended-precision fixed point."""
    g = load(name)
    e = rel_l2(g['H_ref'] if g['H_ref'].shape == g['H_truth'].shape else g['H_ref'][g['rows']],
               g['H_truth'])
    assert e.max() < 1e-4
    assert np.median(e) < 1e-6


@pytest.mark.parametrize('name', SINGLE + ['cfg2_normal', 'kat_tau_maxent', 'kat_srvo3'])
def test_kernel_model_hits_truth(name):
    """numpy model of the HIP kernel (whitened S-form, Bryan-bounded Newton)
    reaches the extended-precision truth to far better than the 1e-6 gate."""
    g = load(name)
    ent = str(g['entropy'])
    basis = SF.Basis(g['U'], g['S'], g['V'], g['err'])
    el = SF.Element(basis, g['G'], g['D'], ent)
    pr = R.Problem(np.zeros((len(g['G']), len(g['D']))), g['U'], g['S'], g['V'], g['G'],
                   g['err'], g['D'], entropy=ent)
    v0 = basis.from_v(R.initial_v(pr, g['delta']))
    out = SF.alpha_chain(basis, el, g['alpha'], v0)
    assert out['converged'].all()
    rows = g['rows']
    e = rel_l2(out['H'][rows], g['H_truth'])
    assert e.max() < 1e-8, e.max()
    # and therefore as close to the reference as the reference is to the truth
    Href = g['H_ref'] if g['H_ref'].shape == g['H_truth'].shape else g['H_ref'][rows]
    assert rel_l2(out['H'][rows], Href).max() < 1e-4
    np.testing.assert_allclose(out['chi2'], g['chi2_ref'], rtol=2e-5)


def test_cov_kernel_model():
    g = load('cov')
    basis = SF.Basis(g['U_rot'], g['S'], g['V'], g['err_rot'])
    el = SF.Element(basis, g['G_rot'], g['D'])
    pr = R.Problem(g['K_rot'], g['U_rot'], g['S'], g['V'], g['G_rot'], g['err_rot'], g['D'])
    v0 = basis.from_v(R.initial_v(pr, g['delta']))
    out = SF.alpha_chain(basis, el, g['alpha'], v0)
    assert out['converged'].all()
    assert rel_l2(out['H'], g['H_truth']).max() < 1e-8


def test_sform_identities_match_port():
    """chi2, S, Q, gradient and Hessian of the whitened S-form equal the
    reference formulas (<= 3e-15 rel)."""
    g = load('cfg1_tauerr')
    p = problem_from(g)
    basis = SF.Basis(g['U'], g['S'], g['V'], g['err'])
    el = SF.Element(basis, g['G'], g['D'])
    rng = np.random.RandomState(5)
    v = 0.1 * rng.randn(len(g['S']))
    a = 37.0
    ev = SF.evaluate(basis, el, a, basis.from_v(v))
    H = R.H_of_v(p, v)
    assert abs(ev['chi2'] - R.chi2_f(p, H)) / R.chi2_f(p, H) < 1e-12
    assert abs(ev['S'] - R.S_f(p, H)) < 1e-12 * abs(R.S_f(p, H)) + 1e-14
    assert abs(ev['Q'] - R.Q_f(p, a, v)) / abs(R.Q_f(p, a, v)) < 1e-12
    gq = basis.c * ev['rho'] + a * basis.from_v(v)
    W = SF.gram(basis, ev['w'])
    d_ref = R.Q_d(p, a, v.copy())
    d_s = basis.to_v(W @ gq)
    assert np.max(np.abs(d_s - d_ref)) / np.max(np.abs(d_ref)) < 1e-9
    J = W @ np.diag(basis.c ** 2) @ W + a * W
    J_ref = R.Q_dd(p, a, v.copy())
    Jr = basis.Q @ J @ basis.Q.T
    assert np.max(np.abs(Jr - J_ref)) / np.max(np.abs(J_ref)) < 1e-9


def test_the_reference_cannot_supply_a_1e_6_golden_but_its_fixed_point_can():
    """SURVEY 8(c) asked for goldens of the reference under ``MaxDerivativeConvergenceMethod(1e-7)``
    (tests/golden/make_golden.py: tight_case, the real reference, maxiter 20000).  What that run shows, as data:
    at the smallest alphas of BASELINE cfg2 the tight run is still ~1e-5 from the fixed point (two alphas do not even
    reach the criterion in 20 000 iterations), no closer than the default run, and the reference's OWN binary64 Newton
    correction there is ~1e-6 -- while at H_truth (the reference's optimum polished in extended precision,
    oracle/hp_truth.py) that correction is < 1e-10 at every alpha.  H_truth is the reference's fixed point; the 1e-6
    parity gate is asserted against it."""
    z = np.load(os.path.join(GOLD, 'tight_ref.npz'))
    g = np.load(os.path.join(GOLD, 'cfg2_normal.npz'))
    # the same default run as the cfg2 fixture, bit for bit, and the same truth
    assert np.array_equal(z['cfg2_H_ref'], g['H_ref']) and np.array_equal(z['cfg2_n_iter_ref'], g['n_iter_ref'])
    assert np.array_equal(z['cfg2_H_truth'], g['H_truth'])

    def rel(a, b):
        return np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
    for name, worst_default, worst_tight in (('cfg2', 9.2e-6, 8.1e-6), ('pm', 1.3e-8, 1.9e-9)):
        e_ref, e_tight = rel(z[name + '_H_ref'], z[name + '_H_truth']), rel(z[name + '_H_tight_ref'], z[name + '_H_truth'])
        assert 0.5 * worst_default < e_ref.max() < 2 * worst_default
        assert 0.5 * worst_tight < e_tight.max() < 2 * worst_tight
        corr = z[name + '_ref_newton_corr']                   # rows: at H_ref, at H_tight_ref, at H_truth
        assert corr[2].max() < 1e-10
    corr = z['cfg2_ref_newton_corr']
    assert corr[2].max() < 1e-10 < 1e-7 < corr[1].max()       # truth: a fixed point; the tight run: not to 1e-6
    assert rel(z['cfg2_H_tight_ref'], z['cfg2_H_truth']).max() > 1e-6
    assert int(z['cfg2_n_iter_tight_ref'].max()) == 20000 and not z['cfg2_converged_tight_ref'].all()
    # where the reference does converge tightly -- the large alphas -- it lands on the truth
    assert rel(z['cfg2_H_tight_ref'], z['cfg2_H_truth'])[:7].max() < 1e-11


def test_hard_inputs_reference_versus_device_as_recorded():
    """VERDICT r02 item 7, as data: tools/stress.py's 100 random cases on the device, and for every case in which the
    device left an alpha unconverged or spent more than 300 evaluations on one, the scan of its worst element solved by
    the oracle port of the reference's algorithm with the reference's defaults (tools/stress_reference.py, build
    container; profiles/r03_b_stress_reference.txt).  On those scans the reference leaves more alphas unconverged than
    the device (error bars far below the noise of the data, a few alphas over many decades: both give up at 1000
    iterations)."""
    z = np.load(os.path.join(GOLD, 'stress_reference.npz'))
    ref = sum(int((z['case%d_ref_converged' % c] == 0).sum()) for c in z['cases'])
    dev = sum(int((z['case%d_dev_converged' % c] == 0).sum()) for c in z['cases'])
    assert len(z['cases']) >= 20 and dev <= ref
    for c in z['cases']:
        assert z['case%d_ref_n_iter' % c].max() <= 1000
