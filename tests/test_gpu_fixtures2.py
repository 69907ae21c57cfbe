"""Reference fixtures of the two input modes that had no test: complex matrix elements (use_complex) and
one covariance matrix per element (tests/golden/make_golden.py: complex_elementwise_case,
elementwise_cov_case; reference python/elementwise_maxent.py:203-219, 236-241, 266, 502-515), and BASELINE
cfg5 at its full size."""
import os

import numpy as np
import pytest

import maxent_amd as mx
from maxent_amd import device, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope='module')
def _audit_every_launch():
    """every launch of the drivers is audited on the device (BatchSolver.solve: info['audit_max']) -- for THIS module only
    (it used to be set at import and leaked into every later test of the process: ADVICE r04)"""
    mp = pytest.MonkeyPatch()
    mp.setenv('MAXENT_AMD_AUDIT', '1')
    yield
    mp.undo()


GATE = 1e-6
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel_l2(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


def test_complex_elements_match_the_reference():
    g = np.load(os.path.join(GOLD, 'complex_elementwise.npz'))
    for herm in (True, False):
        ew = mx.ElementwiseMaxEnt(use_hermiticity=herm, use_complex=True)
        ew.set_verbosity(mx.VerbosityFlags.Quiet)
        ew.set_G_tau_data(g['tau'], g['G_tau'])
        ew.omega = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=60)
        ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=0.05, alpha_max=500, n_points=6)
        ew.set_error(float(g['noise']))
        res = ew.run()
        tag = 'herm%d_' % int(herm)
        assert res.H.shape == g[tag + 'H'].shape == (2, 2, 2, 6, 60)
        assert res.A_out.shape == (2, 2, 60) and np.iscomplexobj(res.A_out)
        # imaginary part of a diagonal element: not calculated, zero in A_out
        assert sorted(tuple(z) for z in res.zero_elements) == sorted(tuple(z) for z in g[tag + 'zero_elements'])
        ref_H, mask = g[tag + 'H'], ~np.isnan(g[tag + 'H'])
        assert np.array_equal(np.isnan(res.H), ~mask)
        # the gate: real and imaginary parts of every element against the fixed point of the reference's own iterates
        # (make_golden.py: truth_of), and the device's audit of every launch
        Ht = g[tag + 'H_truth']
        solved = np.isfinite(Ht).all(axis=-1)
        e = rel_l2(np.asarray(res.H)[solved], Ht[solved])
        assert solved.sum() == (24 if herm else 36) and e.max() < GATE, e.max()
        assert all(info['audit_max'] < GATE for info in ew.last_launches) and len(ew.last_launches) >= 1
        # against the reference's outputs: its own stopping slack (3e-5 of H at small alpha)
        for idx in np.ndindex(2, 2, 2):
            if mask[idx].all():
                assert rel_l2(res.H[idx], ref_H[idx]).max() < 2e-4, idx
                np.testing.assert_allclose(res.chi2[idx], g[tag + 'chi2'][idx], rtol=1e-4)
        np.testing.assert_allclose(res.alpha, g[tag + 'alpha'], rtol=1e-13)
        assert np.max(np.abs(res.A_out - g[tag + 'A_out'])) < 2e-4 * np.max(np.abs(g[tag + 'A_out']))
        if herm:
            # (1, 0) is the conjugate of (0, 1)
            np.testing.assert_array_equal(res.A_out[1, 0], np.conj(res.A_out[0, 1]))


@pytest.mark.parametrize('blur', [False, True])
def test_one_covariance_per_element_matches_the_reference(blur):
    g = np.load(os.path.join(GOLD, 'elementwise_cov.npz'))
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(g['tau'], g['G_tau'])
    ew.omega = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=60)
    if blur:
        ew.maxent_offdiagonal.A_of_H = mx.PreblurA_of_H(b=0.3, omega=ew.omega)
        ew.maxent_offdiagonal.K = mx.PreblurKernel(K=ew.maxent_offdiagonal.K, b=0.3)
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=0.05, alpha_max=500, n_points=6)
    ew.set_cov(g['cov'])
    res = ew.run()
    tag = 'blur%d_' % int(blur)
    assert res.A.shape == g[tag + 'A'].shape == (2, 2, 6, 60)
    np.testing.assert_allclose(res.alpha, g[tag + 'alpha'], rtol=1e-13)
    # the gate: rotated data sets (one per element) against the fixed point of the reference's own iterates, device audit
    e = rel_l2(np.asarray(res.H), g[tag + 'H_truth'])
    assert np.all(np.isfinite(e)) and e.max() < GATE, e.max()
    assert all(info['audit_max'] < GATE for info in ew.last_launches) and len(ew.last_launches) >= 1
    for i in range(2):
        for j in range(2):
            assert rel_l2(res.A[i, j], g[tag + 'A'][i, j]).max() < 2e-4, (i, j)       # (the reference's raw output: its slack)
            np.testing.assert_allclose(res.chi2[i, j], g[tag + 'chi2'][i, j], rtol=1e-4)
    assert rel_l2(res.A_out, g[tag + 'A_out']).max() < 2e-4
    # one decomposition serves every element: the rotated kernels share S and V
    K = ew.maxent_offdiagonal.K
    assert K.rotation is not None and K.V.shape[0] == 60


def test_one_covariance_for_all_elements_matches_the_reference_and_is_one_data_set():
    """ElementwiseMaxEnt.set_cov((T, T)) (elementwise_maxent.py:502-515): the reference's results, from ONE
    eigendecomposition and ONE rotated data set on the device (the same matrix object is recognised by
    TauMaxEnt.set_cov; a data set per element would also switch off the lock-step layout)"""
    g = np.load(os.path.join(GOLD, 'elementwise_shared_cov.npz'))
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(g['tau'], g['G_tau'])
    ew.omega = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=60)
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=0.05, alpha_max=500, n_points=6)
    ew.set_cov(g['cov'])
    res = ew.run()
    assert res.A.shape == g['A'].shape == (2, 2, 6, 60)
    e = rel_l2(np.asarray(res.H), g['H_truth'])
    assert np.all(np.isfinite(e)) and e.max() < GATE, e.max()
    assert all(info['audit_max'] < GATE for info in ew.last_launches)
    for i in range(2):
        for j in range(2):
            assert rel_l2(res.A[i, j], g['A'][i, j]).max() < 2e-4, (i, j)
            np.testing.assert_allclose(res.chi2[i, j], g['chi2'][i, j], rtol=1e-4)
    assert rel_l2(res.A_out, g['A_out']).max() < 2e-4
    assert [info['n_datasets'] for info in ew.last_launches] == [[1], [1]]


def test_cfg5_full_size_fp32_against_fp64_and_the_reference_port():
    """BASELINE cfg5 as written: 8 x 8 G(tau), n_tau = 200, n_omega = 500, 100 alpha, off-diagonals with
    PlusMinusEntropy + preblur (b = 0.1); binary32 streaming variant against binary64 (tolerance classes), and
    a sample of elements of the binary64 run against the oracle port of the reference."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import anchor
    from oracle import ref_numpy as R
    n_orb, n_tau, n_omega, n_alpha = 8, 200, 500, 100
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_omega)
    runs = {}
    for prec in ('f64', 'f32'):
        ew = mx.ElementwiseMaxEnt(use_hermiticity=True, minimizer=mx.LevenbergMinimizer(precision=prec))
        ew.set_verbosity(mx.VerbosityFlags.Quiet)
        ew.set_G_tau_data(tau, Gmat)
        ew.omega = omega
        ew.maxent_offdiagonal.A_of_H = mx.PreblurA_of_H(b=0.1, omega=ew.omega)
        ew.maxent_offdiagonal.K = mx.PreblurKernel(K=ew.maxent_offdiagonal.K, b=0.1)
        ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=n_alpha)
        ew.set_error(synthetic.SIGMA)
        runs[prec] = (ew, ew.run())
    r64, r32 = runs['f64'][1], runs['f32'][1]
    assert r64.A.shape == (8, 8, 100, 500) and not np.any(np.isnan(r64.A))
    assert bool(np.all(r64.converged[np.triu_indices(8)]))
    e = rel_l2(r32.A, r64.A)
    off = ~np.eye(8, dtype=bool)
    # tolerance classes of the binary32 variant: off-diagonal 1e-6, diagonal 1e-4
    assert e[off].max() < 1e-5 and np.mean(e[off] < 1e-6) > 0.95
    assert e[~off].max() < 2e-4
    # the binary64 run against the reference's algorithm on a diagonal and a preblurred off-diagonal element
    D = np.array(runs['f64'][0].maxent_diagonal.D.D)
    Kd = runs['f64'][0].maxent_diagonal.K
    p = R.Problem(np.array(Kd.K), Kd.U, Kd.S, Kd.V, Gmat[2, 2], synthetic.SIGMA * np.ones(n_tau), D)
    alphas = np.asarray(r64.alpha)
    truth, ref = anchor.truth_rows(p, omega.delta, alphas, n_tau, (0, 50, 99), 'normal')
    for ia in (0, 50, 99):
        assert np.linalg.norm(r64.H[2, 2, ia] - truth[ia]) / np.linalg.norm(truth[ia]) < 1e-6
    Ko = runs['f64'][0].maxent_offdiagonal.K
    po = R.Problem(np.array(Ko.K), Ko.U, Ko.S, Ko.V, Gmat[1, 5], synthetic.SIGMA * np.ones(n_tau), D, entropy='plusminus')
    truth, ref = anchor.truth_rows(po, omega.delta, alphas, n_tau, (0, 50, 99), 'plusminus')
    for ia in (0, 50, 99):
        assert np.linalg.norm(r64.H[1, 5, ia] - truth[ia]) / np.linalg.norm(truth[ia]) < 1e-6
    # the spectral function of the blurred element is B H
    B = runs['f64'][0].maxent_offdiagonal.A_of_H.matrix()
    np.testing.assert_allclose(r64.A[1, 5, 50], B @ r64.H[1, 5, 50], rtol=1e-12, atol=1e-14)


def test_bryan_cost_function_run_with_an_error_per_tau():
    """reference test/python/bryan_cost_function.py:32-58 restated: TauMaxEnt(cost_function='bryan') on the semicircular
    G(tau) of the reference's test data with an error bar proportional to G (set through ``set_err``, the loop's
    forwarded setter), five alphas from 0.01 to 2000.  The reference only runs it; here the scan is also compared with
    the oracle port of the reference's Bryan iteration on the same input (its stopping slack) and every problem with
    the device audit."""
    from oracle import ref_numpy as R
    z = np.load(os.path.join(GOLD, 'kat_tau_maxent.npz'))
    table = z['G_clean_file']
    tau, G = table[:, 0].copy(), table[:, 1].copy()
    rng = np.random.RandomState(298347923 % (2 ** 31))
    err = -1.e-3 * G
    G = G + err * rng.randn(len(err))
    tm = mx.TauMaxEnt(cost_function='bryan')
    tm.set_verbosity(mx.VerbosityFlags.Quiet)
    tm.set_G_tau_data(tau, G)
    tm.set_err(err)
    tm.alpha_mesh = mx.LogAlphaMesh(alpha_min=0.01, alpha_max=2000, n_points=5)
    res = tm.run()
    assert res.A.shape == (5, len(tm.omega)) and np.all(np.isfinite(res.A))
    K = tm.K
    p = R.Problem(np.array(K.K), K.U, K.S, K.V, G, err, np.array(tm.D.D), form='bryan')
    ref = R.alpha_loop(p, tm.omega.delta, np.array(tm.alpha_mesh))
    np.testing.assert_allclose(res.alpha, ref['alpha'], rtol=1e-13)
    assert rel_l2(np.asarray(res.H), ref['H']).max() < 5e-5
    np.testing.assert_allclose(res.chi2, ref['chi2'], rtol=1e-4)
    # (the analyzers of a single scan run on the host: the same index as the oracle's curve gives)
    from maxent_amd.analyzers import fit_piecewise
    with np.errstate(all='ignore'):
        assert res.analyzer_results['LineFitAnalyzer']['alpha_index'] == \
            fit_piecewise(np.log(ref['alpha']), np.log(np.asarray(res.chi2)))[0]
