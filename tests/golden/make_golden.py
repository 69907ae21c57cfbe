#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (reads /root/reference, which does not exist
on the GPU box).  It
  1. assembles an importable ``triqs_maxent`` package in a temp dir exactly as
     the reference's CMake would with USE_TRIQS=OFF (SURVEY.md 8c recipe),
  2. runs the reference on seeded inputs and on the data files of its own
     tests (g_tau_semicircular.dat, elementwise_g_tau.npz, srvo3_*.dat),
  3. checks that oracle/ref_numpy.py reproduces the reference bit for bit
     (identical per-alpha iteration counts; H, chi2, S, Q identical),
  4. polishes the reference optimum in extended precision (oracle/hp_truth.py)
     to obtain a golden H that is good to ~1e-12,
  5. writes small .npz fixtures (inputs, U/S/V, reference outputs, truth).

Usage:  python tests/golden/make_golden.py [case_function ...]     (no argument: every fixture)
"""

import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
os.environ['MPLBACKEND'] = 'Agg'


def import_reference():
    tmp = tempfile.mkdtemp(prefix='maxent_ref_')
    pkg = os.path.join(tmp, 'triqs_maxent')
    shutil.copytree(os.path.join(REF, 'python'), pkg)
    subprocess.check_call(['chmod', '-R', 'u+w', pkg])
    os.remove(os.path.join(pkg, 'CMakeLists.txt'))
    for src, subs in (('triqs_support.py.in', {'@TRIQS_V2@': 'OFF', '@TRIQS_V1@': 'OFF',
                                               '@USE_TRIQS@': 'OFF'}),
                      ('version.py.in', {'@MAXENT_VERSION@': '1.2.0', '@TRIQS_GIT_HASH@': '',
                                         '@MAXENT_GIT_HASH@': ''})):
        text = open(os.path.join(pkg, src)).read()
        for k, v in subs.items():
            text = text.replace(k, v)
        open(os.path.join(pkg, src[:-3]), 'w').write(text)
        os.remove(os.path.join(pkg, src))
    with open(os.path.join(tmp, 'decorator.py'), 'w') as f:
        f.write('import functools\n'
                'def decorate(func, caller):\n'
                '    @functools.wraps(func)\n'
                '    def wrapper(*a, **k):\n'
                '        return caller(func, *a, **k)\n'
                '    return wrapper\n')
    sys.path.insert(0, tmp)
    import triqs_maxent
    return triqs_maxent, tmp


tm_mod, TMP = import_reference()
from triqs_maxent import *                                    # noqa: E402,F401,F403
from triqs_maxent.minimizers.convergence_methods import *     # noqa: E402,F401,F403
from oracle import ref_numpy as R, hp_truth                   # noqa: E402

TESTDATA = os.path.join(REF, 'test', 'python')


def record_v(tm):
    """record a copy of v after every alpha (MaxEntResult.v aliases)."""
    vs, its, conv = [], [], []
    orig = tm.minimizer.minimize

    def wrapped(f, v):
        r = orig(f, v)
        vs.append(np.array(r, copy=True))
        its.append(tm.minimizer.n_iter_last)
        conv.append(bool(tm.minimizer.converged))
        return r
    tm.minimizer.minimize = wrapped
    return vs, its, conv


def truth_rows(p, alphas_scaled, vs, entropy, rows):
    out = np.empty((len(rows), len(p.D)))
    for n, ia in enumerate(rows):
        _, H = hp_truth.polish(p.K, p.G, p.err, p.D, p.V, p.S, alphas_scaled[ia],
                               vs[ia], entropy, iters=5)
        out[n] = H
    return out


def check_port(p, delta, mesh, res, its, scale_alpha='Ndata', A_of_H=None, opts=None):
    out = R.alpha_loop(p, delta, np.array(mesh), opts=opts, scale_alpha=scale_alpha,
                       A_of_H=A_of_H)
    assert list(out['n_iter']) == list(its), 'oracle port: iteration counts differ'
    for k in ('H', 'A', 'chi2', 'S', 'Q', 'alpha'):
        a, b = np.asarray(getattr(res, k)), out[k]
        assert np.allclose(a, b, rtol=1e-12, atol=0), 'oracle port differs in ' + k
    return out


def single_case(name, n_tau, n_w, n_alpha, cf, rows, off=False, err=1e-4, preblur_b=None,
                tau_err=False):
    beta = 40.0
    rng = np.random.RandomState(1234)
    tau = np.linspace(0, beta, n_tau)
    omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=n_w)
    K = TauKernel(tau=tau, omega=omega, beta=beta)
    if off:
        A = 0.3 * (np.exp(-(omega - 1.0) ** 2 / (2 * 0.5 ** 2)) -
                   np.exp(-(omega + 1.5) ** 2 / (2 * 0.8 ** 2)))
    else:
        A = 0.6 * np.exp(-(omega - 1.0) ** 2 / (2 * 0.5 ** 2)) + \
            0.4 * np.exp(-(omega + 1.5) ** 2 / (2 * 0.8 ** 2))
        A /= np.trapezoid(A, omega)
    G = K.K_delta @ np.array(A) + 1e-4 * rng.randn(n_tau)
    tm = TauMaxEnt(cost_function=cf)
    tm.set_verbosity(VerbosityFlags.Quiet)
    tm.omega = omega
    tm.set_G_tau_data(tau, G)
    if tau_err:
        errv = err * (1 + 0.5 * np.sin(np.arange(n_tau)))
        tm.set_error(errv)
    else:
        errv = err * np.ones(n_tau)
        tm.set_error(err)
    B = None
    if preblur_b is not None:
        tm.A_of_H = PreblurA_of_H(b=preblur_b, omega=tm.omega)
        tm.K = PreblurKernel(K=tm.K, b=preblur_b)
        B = np.array(tm.A_of_H._B)
    tm.alpha_mesh = LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=n_alpha)
    vs, its, conv = record_v(tm)
    res = tm.run()
    ent = 'plusminus' if cf == 'plusminus' else 'normal'
    form = 'bryan' if cf == 'bryan' else 'maxent'
    Kmat = np.array(tm.K.K)
    p = R.Problem(Kmat, tm.K.U, tm.K.S, tm.K.V, G, errv, np.array(tm.D.D), entropy=ent, form=form)
    check_port(p, omega.delta, tm.alpha_mesh, res, its, A_of_H=B)
    alphas_scaled = np.array(res.alpha)
    Htruth = truth_rows(p, alphas_scaled, vs, ent, rows)
    e = np.linalg.norm(np.array(res.H)[rows] - Htruth, axis=1) / np.linalg.norm(Htruth, axis=1)
    print('%-28s n_s=%d iters=%d  ref-vs-truth max %.2e' % (name, len(tm.K.S), sum(its), e.max()))
    d = dict(tau=tau, omega=np.array(omega), delta=omega.delta, beta=beta, G=G, err=errv,
             D=np.array(tm.D.D), alpha=alphas_scaled, U=tm.K.U, S=tm.K.S, V=tm.K.V,
             entropy=ent, form=form, rows=np.array(rows),
             H_ref=np.array(res.H)[rows], A_ref=np.array(res.A)[rows],
             chi2_ref=np.array(res.chi2), S_ref=np.array(res.S), Q_ref=np.array(res.Q),
             n_iter_ref=np.array(its), converged_ref=np.array(conv),
             v_ref=np.array(vs)[rows], H_truth=Htruth,
             linefit_alpha_index=res.analyzer_results['LineFitAnalyzer']['alpha_index'],
             chi2curv_alpha_index=res.analyzer_results['Chi2CurvatureAnalyzer']['alpha_index'],
             A_out_linefit=res.analyzer_results['LineFitAnalyzer']['A_out'],
             A_out_chi2curv=res.analyzer_results['Chi2CurvatureAnalyzer']['A_out'],
             A_out_entropy=res.analyzer_results['EntropyAnalyzer']['A_out'])
    if B is not None:
        d['B'] = B
        d['preblur_b'] = preblur_b
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)


def kat_tau_maxent():
    """reference test/python/tau_maxent.py:34-47,134-135 (known answer:
    5 log-probabilities to 6 decimals)."""
    np.random.seed(9)
    tm = TauMaxEnt(probability='normal')
    tm.set_verbosity(VerbosityFlags.Quiet)
    tm.set_G_tau_file(os.path.join(TESTDATA, 'g_tau_semicircular.dat'))
    tm.set_G_tau_data(tm.tau, tm.G + 1.e-3 * np.random.randn(len(tm.G)))
    tm.alpha_mesh = LogAlphaMesh(alpha_min=0.08, n_points=5)
    tm.omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=200)
    tm.set_error(1.e-3)
    vs, its, conv = record_v(tm)
    res = tm.run()
    kat = [-8476.52812836, -2343.02752796, -704.28318351, -280.26627323, -175.30592555]
    np.testing.assert_almost_equal(res.probability, kat, 6)
    G = np.array(tm.G)
    p = R.Problem(np.array(tm.K.K), tm.K.U, tm.K.S, tm.K.V, G, np.array(tm.err),
                  np.array(tm.D.D))
    out = check_port(p, tm.omega.delta, tm.alpha_mesh, res, its)
    lp = [R.log_probability(p, a, v) for a, v in zip(out['alpha'], out['v'])]
    np.testing.assert_almost_equal(lp, kat, 6)
    rows = list(range(5))
    Htruth = truth_rows(p, np.array(res.alpha), vs, 'normal', rows)
    print('%-28s n_s=%d iters=%d' % ('kat_tau_maxent', len(tm.K.S), sum(its)))
    np.savez_compressed(os.path.join(HERE, 'kat_tau_maxent.npz'),
                        tau=np.array(tm.tau), omega=np.array(tm.omega), delta=tm.omega.delta,
                        G=G, err=np.array(tm.err), D=np.array(tm.D.D), alpha=np.array(res.alpha),
                        U=tm.K.U, S=tm.K.S, V=tm.K.V, entropy='normal', form='maxent',
                        rows=np.array(rows), H_ref=np.array(res.H), A_ref=np.array(res.A),
                        chi2_ref=np.array(res.chi2), S_ref=np.array(res.S), Q_ref=np.array(res.Q),
                        n_iter_ref=np.array(its), converged_ref=np.array(conv),
                        v_ref=np.array(vs), H_truth=Htruth,
                        probability_ref=np.array(res.probability), probability_kat=np.array(kat),
                        bryan_A_out=np.array(res.analyzer_results['BryanAnalyzer']['A_out']),
                        classic_A_out=np.array(res.analyzer_results['ClassicAnalyzer']['A_out']),
                        classic_alpha_index=int(res.analyzer_results['ClassicAnalyzer']['alpha_index']),
                        G_clean_file=np.loadtxt(os.path.join(TESTDATA, 'g_tau_semicircular.dat')))


def kat_huge_alpha():
    """reference test/python/huge_alpha.py:30-50: alpha -> 1e10 gives H -> D."""
    np.random.seed(77)
    tm = TauMaxEnt()
    tm.set_verbosity(VerbosityFlags.Quiet)
    tm.set_G_tau_file(os.path.join(TESTDATA, 'g_tau_semicircular.dat'))
    tm.set_G_tau_data(tm.tau, tm.G + 1.e-4 * np.random.randn(len(tm.G)))
    tm.alpha_mesh = LogAlphaMesh(alpha_min=1e10 - 1, alpha_max=1e10, n_points=5)
    tm.set_error(5.e-4)
    tm.reduce_singular_space = 1.e-16
    vs, its, conv = record_v(tm)
    res = tm.run()
    assert np.max(res.H - tm.D.D) < 1e-6
    print('%-28s n_s=%d iters=%d' % ('kat_huge_alpha', len(tm.K.S), sum(its)))
    np.savez_compressed(os.path.join(HERE, 'kat_huge_alpha.npz'),
                        tau=np.array(tm.tau), omega=np.array(tm.omega), delta=tm.omega.delta,
                        G=np.array(tm.G), err=np.array(tm.err), D=np.array(tm.D.D),
                        alpha=np.array(res.alpha), U=tm.K.U, S=tm.K.S, V=tm.K.V,
                        entropy='normal', form='maxent', H_ref=np.array(res.H),
                        chi2_ref=np.array(res.chi2), S_ref=np.array(res.S), Q_ref=np.array(res.Q),
                        n_iter_ref=np.array(its))


def kat_srvo3():
    """reference test/python/srvo3_mesh_and_ALPS.py:37-51,100: Bryan cost
    function, n_tau = n_omega = 500, Lorentzian mesh; A(alpha_1) equals the
    ALPS 'maxspec' column to 2 decimals."""
    gt = np.loadtxt(os.path.join(TESTDATA, 'srvo3_mesh_and_ALPS_gtau.dat'))
    ms = np.loadtxt(os.path.join(TESTDATA, 'srvo3_mesh_and_ALPS_maxspec.dat'))
    tm = TauMaxEnt(cost_function='bryan')
    tm.set_verbosity(VerbosityFlags.Quiet)
    tm.set_G_tau_data(gt[:, 0], gt[:, 1])
    tm.set_error(gt[:, 2])
    tm.omega = LorentzianOmegaMesh(omega_min=-15, omega_max=15, n_points=500)
    n_tau = len(gt)
    tm.alpha_mesh = LogAlphaMesh(alpha_min=5.514845959 / n_tau, alpha_max=100, n_points=2)
    vs, its, conv = record_v(tm)
    res = tm.run()
    dmax = np.max(np.abs(res.A[1, :] - np.interp(np.array(tm.omega), ms[:, 0], ms[:, 1])))
    p = R.Problem(np.array(tm.K.K), tm.K.U, tm.K.S, tm.K.V, np.array(tm.G), np.array(tm.err),
                  np.array(tm.D.D), form='bryan')
    check_port(p, tm.omega.delta, tm.alpha_mesh, res, its)
    Htruth = truth_rows(p, np.array(res.alpha), vs, 'normal', [0, 1])
    print('%-28s n_s=%d iters=%d  max|A-ALPS|=%.2e' % ('kat_srvo3', len(tm.K.S), sum(its), dmax))
    np.savez_compressed(os.path.join(HERE, 'kat_srvo3.npz'),
                        tau=gt[:, 0], omega=np.array(tm.omega), delta=tm.omega.delta,
                        G=np.array(tm.G), err=np.array(tm.err), D=np.array(tm.D.D),
                        alpha=np.array(res.alpha), U=tm.K.U, S=tm.K.S, V=tm.K.V,
                        entropy='normal', form='bryan', rows=np.array([0, 1]),
                        H_ref=np.array(res.H), A_ref=np.array(res.A), chi2_ref=np.array(res.chi2),
                        S_ref=np.array(res.S), Q_ref=np.array(res.Q), n_iter_ref=np.array(its),
                        converged_ref=np.array(conv), v_ref=np.array(vs), H_truth=Htruth,
                        alps_maxspec=ms)



def record_solves(ew):
    """every minimisation of an element-wise run as the reference set it up -- the data its cost function held (kernel as
    rotated / blurred for that element, G, err, default model, V) and the v its minimiser ended at -- keyed by matrix element,
    complex index and alpha index: what the truth rows of the element-wise fixtures are polished from"""
    log, cur = [], {}
    for worker in (ew.maxent_diagonal, ew.maxent_offdiagonal):
        loop = worker.maxent_loop

        def run(*a, _o=loop.run, **k):
            cur['elem'], cur['ci'], cur['ia'] = k.get('matrix_element'), k.get('complex_index'), 0
            return _o(*a, **k)

        def minimize(f, v, _o=loop.minimizer.minimize):
            r = _o(f, v)
            log.append(dict(elem=tuple(cur['elem']), ci=cur['ci'], ia=cur['ia'], alpha=float(f._alpha),
                            K=np.array(f.chi2.K.K), G=np.array(f.chi2.G),
                            err=np.array(f.chi2.err, dtype=float) * np.ones(len(f.chi2.G)),
                            D=np.array(f.S.D.D), V=np.array(f.H_of_v.K.V), v=np.array(r, copy=True),
                            entropy='plusminus' if 'PlusMinus' in type(f.S).__name__ else 'normal'))
            cur['ia'] += 1
            return r
        loop.run = run
        loop.minimizer.minimize = minimize
    return log


def truth_of(log, shape, D_of=None):
    """H_truth in the layout of MaxEntResult.H (NaN where the run solved nothing): every recorded solve polished in extended
    precision from the reference's own iterate (oracle/hp_truth.py), with the reference's data"""
    Ht = np.full(shape, np.nan)
    for s in log:
        D = s['D'] if D_of is None else D_of(s)
        _, H = hp_truth.polish(s['K'], s['G'], s['err'], D, s['V'], None, s['alpha'], s['v'], s['entropy'], iters=6)
        Ht[s['elem'] + ((s['ci'] or 0,) if len(shape) == 5 else ()) + (s['ia'],)] = H
    return Ht


def elementwise_case():
    """reference test/python/elementwise_maxent.py:101-188 on its own fixture
    elementwise_g_tau.npz (2x2x201, beta=400)."""
    with np.load(os.path.join(TESTDATA, 'elementwise_g_tau.npz')) as data:
        tau = data['tau']
        G_tau_noise = data['G_tau_noise']
        G_w_rot = data['G_w_rot']
        w = data['w']
    noise = 1e-3
    out = {}
    for name, cls, herm in (('ew', ElementwiseMaxEnt, False), ('pm', PoormanMaxEnt, False)):
        ew = cls(use_hermiticity=herm)
        ew.set_verbosity(VerbosityFlags.Quiet)
        ew.set_G_tau_data(tau, G_tau_noise)
        ew.omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=80)
        ew.alpha_mesh = LogAlphaMesh(alpha_min=0.05, alpha_max=500, n_points=8)
        ew.set_error(noise)
        log = record_solves(ew)
        res = ew.run()
        lf = np.array([[res.analyzer_results[i][j]['LineFitAnalyzer']['alpha_index'] for j in range(2)] for i in range(2)])
        delta = ew.omega.delta
        if name == 'ew':
            out[name + '_H_truth'] = truth_of(log, np.array(res.H).shape)
        else:
            # Poorman (elementwise_maxent.py:637-652): the default model of an off-diagonal element is sqrt(A_ii A_jj) + 1e-6
            # of the diagonal elements' LineFit spectra.  The truth of the whole procedure takes the TRUTH of those spectra
            # (the reference's own differ from it by its stopping slack, 1e-5 relative, and so would every off-diagonal H)
            diag = truth_of([q for q in log if q['elem'][0] == q['elem'][1]], np.array(res.H).shape)

            def D_pm(q):
                i, j = q['elem']
                if i == j:
                    return q['D']
                A1, A2 = diag[i, i, lf[i, i]] / delta, diag[j, j, lf[j, j]] / delta
                D = (np.sqrt(A1 * A2) + 1e-6) * delta
                assert np.allclose(D, q['D'], rtol=1e-3, atol=1e-9), np.max(np.abs(D - q['D']) / q['D'])          # (the reference's, up to its slack)
                return D
            out[name + '_H_truth'] = truth_of(log, np.array(res.H).shape, D_pm)
        e = np.linalg.norm(np.array(res.H) - out[name + '_H_truth'], axis=-1) / np.linalg.norm(out[name + '_H_truth'], axis=-1)
        print('   %s: reference vs truth, rel L2 of H: max %.2e' % (name, np.nanmax(e)))
        out[name + '_A'] = np.array(res.A)
        out[name + '_H'] = np.array(res.H)
        out[name + '_chi2'] = np.array(res.chi2)
        out[name + '_S'] = np.array(res.S)
        out[name + '_Q'] = np.array(res.Q)
        out[name + '_A_out'] = np.array(res.A_out)
        out[name + '_alpha'] = np.array(res.alpha)
        out[name + '_linefit_idx'] = np.array(
            [[res.analyzer_results[i][j]['LineFitAnalyzer']['alpha_index'] for j in range(2)]
             for i in range(2)])
        K = ew.maxent_diagonal.K
        out['U'], out['S'], out['V'] = K.U, K.S, K.V
        out['omega'] = np.array(ew.omega)
        out['delta'] = ew.omega.delta
        out['D'] = np.array(ew.maxent_diagonal.D.D)
    print('%-28s n_s=%d' % ('elementwise', len(out['S'])))
    np.savez_compressed(os.path.join(HERE, 'elementwise.npz'), tau=tau, G_tau_noise=G_tau_noise,
                        noise=noise, w_exact=w[::25], A01_exact=(-1.0 / np.pi * np.imag(G_w_rot[:, 0, 1]))[::25],
                        **out)


def cov_case():
    """TauMaxEnt.set_cov (tau_maxent.py:253-288): full covariance -> rotated problem."""
    n_tau, n_w, n_alpha = 60, 120, 12
    beta = 40.0
    rng = np.random.RandomState(4321)
    tau = np.linspace(0, beta, n_tau)
    omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=n_w)
    K = TauKernel(tau=tau, omega=omega, beta=beta)
    A = 0.6 * np.exp(-(omega - 1.0) ** 2 / (2 * 0.5 ** 2)) + 0.4 * np.exp(-(omega + 1.5) ** 2 / (2 * 0.8 ** 2))
    A /= np.trapezoid(A, omega)
    # correlated noise: cov = L L^T
    L = 1e-4 * (np.eye(n_tau) + 0.3 * np.diag(np.ones(n_tau - 1), 1) + 0.1 * rng.randn(n_tau, n_tau) / np.sqrt(n_tau))
    cov = L @ L.T
    G = K.K_delta @ np.array(A) + L @ rng.randn(n_tau)
    tm = TauMaxEnt()
    tm.set_verbosity(VerbosityFlags.Quiet)
    tm.omega = omega
    tm.set_G_tau_data(tau, G)
    tm.set_cov(cov)
    tm.alpha_mesh = LogAlphaMesh(alpha_min=1e-2, alpha_max=1e3, n_points=n_alpha)
    vs, its, conv = record_v(tm)
    res = tm.run()
    p = R.Problem(np.array(tm.K.K), tm.K.U, tm.K.S, tm.K.V, np.array(tm.G), np.array(tm.err),
                  np.array(tm.D.D))
    check_port(p, omega.delta, tm.alpha_mesh, res, its)
    rows = list(range(n_alpha))
    Htruth = truth_rows(p, np.array(res.alpha), vs, 'normal', rows)
    e = np.linalg.norm(np.array(res.H) - Htruth, axis=1) / np.linalg.norm(Htruth, axis=1)
    print('%-28s n_s=%d iters=%d  ref-vs-truth max %.2e' % ('cov', len(tm.K.S), sum(its), e.max()))
    np.savez_compressed(os.path.join(HERE, 'cov.npz'), tau=tau, omega=np.array(omega),
                        delta=omega.delta, beta=beta, G_orig=G, cov=cov, G_rot=np.array(tm.G),
                        err_rot=np.array(tm.err), K_rot=np.array(tm.K.K), U_rot=tm.K.U, S=tm.K.S,
                        V=tm.K.V, D=np.array(tm.D.D), alpha=np.array(res.alpha),
                        H_ref=np.array(res.H), A_ref=np.array(res.A), chi2_ref=np.array(res.chi2),
                        S_ref=np.array(res.S), Q_ref=np.array(res.Q), n_iter_ref=np.array(its),
                        v_ref=np.array(vs), H_truth=Htruth, G_rec_ref=np.array(res.G_rec))


def derivs_case():
    """reference test/python/maxent_cost_function_d.py:27-58 (same inputs and seed): f, d, dd of the
    cost function in every mode of MaxEntCostFunction, of BryanCostFunction, of the plus-minus pair,
    and the component functions, at a random v."""
    np.random.seed(658436166)
    beta = 40
    tau = np.linspace(0, beta, 100)
    omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=100)
    K = TauKernel(tau=tau, omega=omega, beta=beta)
    A = np.exp(-omega ** 2)
    A /= np.trapezoid(A, omega)
    G = np.dot(K.K, A)
    G += 1.e-4 * np.random.randn(len(G))
    err = 1.e-4 * np.ones(len(G))
    D = FlatDefaultModel(omega=omega)
    out = dict(tau=tau, omega=np.array(omega), delta=omega.delta, beta=float(beta), G=G, err=err,
               D=np.array(D.D), U=K.U, S=K.S, V=K.V, alpha=0.1)
    v = np.random.rand(len(K.S))
    v = np.random.rand(len(v))          # the reference test draws twice (lines 49, 53)
    out['v'] = v

    def components(prefix, Q, v):
        b = Q(v)
        H = b.H_of_v.f()
        out[prefix + 'H'] = np.array(H)
        out[prefix + 'dH_dv'] = np.array(b.H_of_v.d())
        out[prefix + 'chi2'] = float(b.chi2.f())
        out[prefix + 'dchi2_dH'] = np.array(b.chi2.d())
        out[prefix + 'S'] = float(b.S.f())
        out[prefix + 'dS_dH'] = np.array(b.S.d())
        out[prefix + 'ddS_diag'] = np.array(np.diag(b.S.dd()))
        out[prefix + 'v_of_H'] = np.array(Q.H_of_v.inv(np.array(H)))

    chi2 = NormalChi2(K=K, G=G, err=err)
    Q = MaxEntCostFunction(chi2=chi2, S=NormalEntropy(D=D), H_of_v=NormalH_of_v(D=D, K=K))
    Q.set_alpha(0.1)
    # the reference's own finite-difference check (maxent_cost_function_d.py:55-58)
    for Q.d_dv in [True]:
        for Q.dA_projection in range(3):
            assert Q.check_derivatives(v, Q.f(v), prec=1.e-8)
    for d_dv in (False, True):
        for proj in range(3):
            Q.d_dv, Q.dA_projection = d_dv, proj
            tag = 'n_ddv%d_p%d_' % (int(d_dv), proj)
            out[tag + 'f'] = float(Q.f(v))
            out[tag + 'd'] = np.array(Q.d(v))
            out[tag + 'dd'] = np.array(Q.dd(v))
    Q.d_dv, Q.dA_projection = False, 2
    components('n_', Q, v)
    Qe = MaxEntCostFunction(chi2=NormalChi2(K=K, G=G, err=err), S=NormalEntropy(D=D),
                            H_of_v=NormalH_of_v(D=D, K=K), chi2_factor=2.5)
    Qe.set_alpha(0.1)
    out['n_eta_f'], out['n_eta_d'], out['n_eta_dd'] = float(Qe.f(v)), np.array(Qe.d(v)), np.array(Qe.dd(v))
    out['chi2_factor'] = 2.5
    Qb = BryanCostFunction()
    Qb.chi2 = NormalChi2(K=K, G=G, err=err)
    Qb.set_D(D)
    Qb.H_of_v.set_K(K)
    Qb.set_alpha(0.1)
    out['b_f'], out['b_d'], out['b_dd'] = float(Qb.f(v)), np.array(Qb.d(v)), np.array(Qb.dd(v))
    vp = 0.3 * (v - 0.5)
    out['v_pm'] = vp
    Qp = MaxEntCostFunction(chi2=NormalChi2(K=K, G=G, err=err), S=PlusMinusEntropy(D=D),
                            H_of_v=PlusMinusH_of_v(D=D, K=K))
    Qp.set_alpha(0.1)
    for d_dv in (False, True):
        Qp.d_dv = d_dv
        tag = 'pm_ddv%d_p2_' % int(d_dv)
        out[tag + 'f'], out[tag + 'd'], out[tag + 'dd'] = float(Qp.f(vp)), np.array(Qp.d(vp)), np.array(Qp.dd(vp))
    Qp.d_dv = False
    components('pm_', Qp, vp)
    # the oracle restates the default and the Bryan form: pin it here too
    p = R.Problem(np.array(K.K), K.U, K.S, K.V, G, err, np.array(D.D))
    assert np.allclose(R.Q_d(p, 0.1, v), out['n_ddv0_p2_d'], rtol=1e-12, atol=0)
    assert np.allclose(R.Q_dd(p, 0.1, v), out['n_ddv0_p2_dd'], rtol=1e-12, atol=0)
    print('%-28s n_s=%d' % ('derivs', len(K.S)))
    np.savez_compressed(os.path.join(HERE, 'derivs.npz'), **out)


def plusminus_entropy_case():
    """reference test/python/plus_minus_entropy.py:46-59: PlusMinusEntropy f, d, dd on a linear mesh
    with D = 0.9 at a random A (its closed-form twin lives in the test that reads this fixture)."""
    from triqs_maxent.functions import PlusMinusEntropy as PME, NormalEntropy as NE
    w = LinearOmegaMesh(-10, 10, 101)
    D = DataDefaultModel(0 * w + 0.9, w)
    np.random.seed(6666)
    A = np.random.rand(len(w))
    S2 = PME(D=D)(A)
    Sn = NE(D=D)(A)
    print('%-28s' % 'plusminus_entropy')
    np.savez_compressed(os.path.join(HERE, 'plusminus_entropy.npz'), omega=np.array(w), D=np.array(D.D),
                        A=A, f=float(S2.f()), d=np.array(S2.d()), dd_diag=np.array(np.diag(S2.dd())),
                        normal_f=float(Sn.f()), normal_d=np.array(Sn.d()), normal_dd_diag=np.array(np.diag(Sn.dd())))


def complex_elementwise_case():
    """ElementwiseMaxEnt(use_complex=True) (elementwise_maxent.py:203-219, 236-241, 266): a complex
    hermitian 2x2 G(tau); real and imaginary parts of the off-diagonals are separate real problems."""
    with np.load(os.path.join(TESTDATA, 'elementwise_g_tau.npz')) as data:
        tau = data['tau']
        G_re = data['G_tau_noise']
    rng = np.random.RandomState(99)
    G = np.array(G_re, dtype=complex)
    im = 0.4 * G_re[0, 1] + 2e-3 * rng.randn(len(tau))
    G[0, 1] = G_re[0, 1] + 1j * im
    G[1, 0] = G_re[0, 1] - 1j * im
    out = dict(tau=tau, G_tau=G, noise=1e-3)
    for herm in (True, False):
        ew = ElementwiseMaxEnt(use_hermiticity=herm, use_complex=True)
        ew.set_verbosity(VerbosityFlags.Quiet)
        ew.set_G_tau_data(tau, G)
        ew.omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=60)
        ew.alpha_mesh = LogAlphaMesh(alpha_min=0.05, alpha_max=500, n_points=6)
        ew.set_error(1e-3)
        log = record_solves(ew)
        res = ew.run()
        tag = 'herm%d_' % int(herm)
        out[tag + 'H_truth'] = truth_of(log, np.array(res.H).shape)
        for k in ('H', 'A', 'chi2', 'S', 'Q', 'alpha', 'A_out'):
            out[tag + k] = np.array(getattr(res, k))
        out[tag + 'zero_elements'] = np.array(res.zero_elements, dtype=int)
        out['omega'] = np.array(ew.omega)
    print('%-28s shapes H %s A_out %s' % ('complex_elementwise', out['herm1_H'].shape, out['herm1_A_out'].shape))
    np.savez_compressed(os.path.join(HERE, 'complex_elementwise.npz'), **out)


def elementwise_cov_case():
    """ElementwiseMaxEnt.set_cov with one covariance per element, (M, N, T, T)
    (elementwise_maxent.py:502-515, tau_maxent.py:253-288), with and without preblur."""
    with np.load(os.path.join(TESTDATA, 'elementwise_g_tau.npz')) as data:
        tau = data['tau'][::3]
        G = data['G_tau_noise'][:, :, ::3]
    T = len(tau)
    rng = np.random.RandomState(2718)
    cov = np.empty((2, 2, T, T))
    for i in range(2):
        for j in range(2):
            L = 1e-3 * (np.eye(T) * (1.0 + 0.2 * (i + 2 * j)) + 0.25 * np.diag(np.ones(T - 1), 1) +
                        0.1 * rng.randn(T, T) / np.sqrt(T))
            cov[i, j] = L @ L.T
    out = dict(tau=tau, G_tau=G, cov=cov)
    for blur in (False, True):
        ew = ElementwiseMaxEnt(use_hermiticity=False)
        ew.set_verbosity(VerbosityFlags.Quiet)
        ew.set_G_tau_data(tau, G)
        ew.omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=60)
        if blur:
            ew.maxent_offdiagonal.A_of_H = PreblurA_of_H(b=0.3, omega=ew.omega)
            ew.maxent_offdiagonal.K = PreblurKernel(K=ew.maxent_offdiagonal.K, b=0.3)
        ew.alpha_mesh = LogAlphaMesh(alpha_min=0.05, alpha_max=500, n_points=6)
        ew.set_cov(cov)
        log = record_solves(ew)
        res = ew.run()
        tag = 'blur%d_' % int(blur)
        out[tag + 'H_truth'] = truth_of(log, np.array(res.H).shape)
        for k in ('H', 'A', 'chi2', 'S', 'Q', 'alpha', 'A_out'):
            out[tag + k] = np.array(getattr(res, k))
        out['omega'] = np.array(ew.omega)
    print('%-28s' % 'elementwise_cov')
    np.savez_compressed(os.path.join(HERE, 'elementwise_cov.npz'), **out)


def logtaker_case():
    """expected output files of the reference's own tests (test/python/logtaker.py with logtaker.ref /
    logtaker.dat.ref, omega_meshes.py with omega_meshes.ref, alpha_meshes.py with alpha_meshes.ref): data files
    of the reference's test suite, kept byte for byte"""
    for name in ('logtaker.ref', 'logtaker.dat.ref', 'omega_meshes.ref', 'alpha_meshes.ref'):
        shutil.copyfile(os.path.join(TESTDATA, name), os.path.join(HERE, name))
    print('%-28s' % 'logtaker')


def elementwise_shared_cov_case():
    """ElementwiseMaxEnt.set_cov with ONE (T, T) covariance for all elements (elementwise_maxent.py:502-515):
    every element goes through TauMaxEnt.set_cov with the same matrix (tau_maxent.py:253-288)."""
    with np.load(os.path.join(TESTDATA, 'elementwise_g_tau.npz')) as data:
        tau = data['tau'][::3]
        G = data['G_tau_noise'][:, :, ::3]
    T = len(tau)
    rng = np.random.RandomState(1414)
    L = 1e-3 * (np.eye(T) + 0.3 * np.diag(np.ones(T - 1), 1) + 0.1 * rng.randn(T, T) / np.sqrt(T))
    cov = L @ L.T
    ew = ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(VerbosityFlags.Quiet)
    ew.set_G_tau_data(tau, G)
    ew.omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=60)
    ew.alpha_mesh = LogAlphaMesh(alpha_min=0.05, alpha_max=500, n_points=6)
    ew.set_cov(cov)
    log = record_solves(ew)
    res = ew.run()
    out = dict(tau=tau, G_tau=G, cov=cov, omega=np.array(ew.omega))
    out['H_truth'] = truth_of(log, np.array(res.H).shape)
    for k in ('H', 'A', 'chi2', 'S', 'Q', 'alpha', 'A_out'):
        out[k] = np.array(getattr(res, k))
    print('%-28s' % 'elementwise_shared_cov')
    np.savez_compressed(os.path.join(HERE, 'elementwise_shared_cov.npz'), **out)


def _ref_newton_corr(tm, alpha_scaled, v):
    """the REFERENCE's own binary64 Newton correction at v: delta = solve(Q.dd(v), Q.d(v)) with its default
    MaxEntCostFunction (maxent_cost_function.py:120-165), measured like the device audit as
    ||dH/dv delta||_2 / ||H||_2"""
    Q = tm.cost_function
    Q.set_alpha(alpha_scaled)
    v = np.array(v, copy=True)
    delta = np.linalg.solve(np.array(Q.dd(v)), np.array(Q.d(v)))
    b = Q(v)
    H = np.array(b.H_of_v.f())
    return float(np.linalg.norm(np.array(b.H_of_v.d()) @ delta) / np.linalg.norm(H))


def tight_case():
    """the reference itself under ``MaxDerivativeConvergenceMethod(1e-7)``
    (convergence_methods.py:81-89) next to its default stopping rules, for BASELINE cfg2 (normal entropy)
    and one plus-minus scan -- H of both runs, the extended-precision fixed point, and the reference's OWN
    binary64 Newton correction at the three points.  It shows what a 1e-6 parity gate can be anchored on:
    the tight run is still 1e-6 ... 1e-5 from the fixed point at the smallest alphas while its Newton
    correction there is far above that of H_truth."""
    out = {}
    for name, n_tau, n_w, n_alpha, cf, off, rows in (
            ('cfg2', 200, 500, 100, 'normal', False, list(range(0, 100, 9)) + [99]),
            ('pm', 100, 200, 20, 'plusminus', True, list(range(20)))):
        beta = 40.0
        rng = np.random.RandomState(1234)
        tau = np.linspace(0, beta, n_tau)
        omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=n_w)
        K = TauKernel(tau=tau, omega=omega, beta=beta)
        if off:
            A = 0.3 * (np.exp(-(omega - 1.0) ** 2 / (2 * 0.5 ** 2)) -
                       np.exp(-(omega + 1.5) ** 2 / (2 * 0.8 ** 2)))
        else:
            A = 0.6 * np.exp(-(omega - 1.0) ** 2 / (2 * 0.5 ** 2)) + \
                0.4 * np.exp(-(omega + 1.5) ** 2 / (2 * 0.8 ** 2))
            A /= np.trapezoid(A, omega)
        G = K.K_delta @ np.array(A) + 1e-4 * rng.randn(n_tau)
        runs = {}
        for tag, minimizer in (('ref', None),
                               ('tight', LevenbergMinimizer(convergence=MaxDerivativeConvergenceMethod(1e-7),
                                                            maxiter=20000))):
            kw = {} if minimizer is None else dict(minimizer=minimizer)
            tm = TauMaxEnt(cost_function=cf, **kw)
            tm.set_verbosity(VerbosityFlags.Quiet)
            tm.omega = omega
            tm.set_G_tau_data(tau, G)
            tm.set_error(1e-4)
            tm.alpha_mesh = LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=n_alpha)
            vs, its, conv = record_v(tm)
            res = tm.run()
            runs[tag] = (tm, res, vs, its, conv)
        tm, res, vs, its, conv = runs['ref']
        tmt, rest, vst, itst, convt = runs['tight']
        ent = 'plusminus' if cf == 'plusminus' else 'normal'
        p = R.Problem(np.array(tm.K.K), tm.K.U, tm.K.S, tm.K.V, G, 1e-4 * np.ones(n_tau), np.array(tm.D.D),
                      entropy=ent)
        alphas = np.array(res.alpha)
        Ht = np.empty((len(rows), n_w))
        corr = np.empty((3, len(rows)))
        for n, ia in enumerate(rows):
            vt, Ht[n] = hp_truth.polish(p.K, p.G, p.err, p.D, p.V, p.S, alphas[ia], vs[ia], ent, iters=5)
            corr[0, n] = _ref_newton_corr(tm, alphas[ia], vs[ia])
            corr[1, n] = _ref_newton_corr(tm, alphas[ia], vst[ia])
            corr[2, n] = _ref_newton_corr(tm, alphas[ia], vt)

        def rel(a):
            return np.linalg.norm(a - Ht, axis=1) / np.linalg.norm(Ht, axis=1)
        e_ref, e_tight = rel(np.array(res.H)[rows]), rel(np.array(rest.H)[rows])
        print('%-28s ref-vs-truth max %.2e  tight-ref-vs-truth max %.2e  iterations %d / %d' %
              ('tight_' + name, e_ref.max(), e_tight.max(), sum(its), sum(itst)))
        print('   reference Newton correction  at H_ref %.1e..%.1e  at H_tight %.1e..%.1e  at H_truth %.1e..%.1e' %
              (corr[0].min(), corr[0].max(), corr[1].min(), corr[1].max(), corr[2].min(), corr[2].max()))
        out.update({name + '_rows': np.array(rows), name + '_alpha': alphas,
                    name + '_H_ref': np.array(res.H)[rows], name + '_H_tight_ref': np.array(rest.H)[rows],
                    name + '_H_truth': Ht, name + '_n_iter_ref': np.array(its),
                    name + '_n_iter_tight_ref': np.array(itst), name + '_converged_tight_ref': np.array(convt),
                    name + '_ref_newton_corr': corr})
    np.savez_compressed(os.path.join(HERE, 'tight_ref.npz'), **out)


if __name__ == '__main__':
    only = sys.argv[1:]
    if only:
        for name in only:
            globals()[name]()
        shutil.rmtree(TMP, ignore_errors=True)
        sys.exit(0)
    single_case('cfg1_normal', 100, 200, 20, 'normal', list(range(20)))
    single_case('cfg1_bryan', 100, 200, 20, 'bryan', list(range(20)))
    single_case('cfg1_plusminus', 100, 200, 20, 'plusminus', list(range(20)), off=True)
    single_case('cfg1_tauerr', 100, 200, 20, 'normal', list(range(20)), tau_err=True)
    single_case('cfg5_preblur_pm', 100, 200, 20, 'plusminus', list(range(20)), off=True,
                preblur_b=0.1)
    single_case('cfg2_normal', 200, 500, 100, 'normal', list(range(0, 100, 9)) + [99])
    kat_tau_maxent()
    kat_huge_alpha()
    kat_srvo3()
    elementwise_case()
    cov_case()
    derivs_case()
    plusminus_entropy_case()
    complex_elementwise_case()
    elementwise_cov_case()
    elementwise_shared_cov_case()
    logtaker_case()
    tight_case()
    shutil.rmtree(TMP, ignore_errors=True)
    print('fixtures written to', HERE)
