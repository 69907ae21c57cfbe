"""CPU tests: the C-ABI library loads and exports every symbol the header
declares (no compute without a GPU), and the host-side mirror of the reference
interface behaves like the reference (meshes, kernel fill, SVD staging,
analyzers, MaxEntResult layout, attribute shadowing, error behaviour)."""
import os
import re

import numpy as np
import pytest

import maxent_amd as mx
from maxent_amd import device, synthetic
from oracle import ref_numpy as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    with np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


# ---- C-ABI ------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'maxent_hip.h')).read()
    declared = set(re.findall(r'\b(mxe_[a-z0-9_]+)\s*\(', header))
    declared -= {'mxe_ctx', 'mxe_opts'}
    lib = device.load_library()
    bound = {name for name, _, _ in device.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mxe_version().startswith(b'maxent_hip')
    assert lib.mxe_strerror(0) == b'ok'
    assert b'argument' in lib.mxe_strerror(-1)


def test_opts_defaults_and_struct_layout():
    o = device.default_opts()
    assert (o.maxiter, o.miniter) == (1000, 0)
    assert o.tol_h == 1e-9 and o.tol_d == 0.0 and o.tol_relq == 0.0
    assert o.step_max == 0.2 and o.mu_grow == 4.0 and o.decouple_tol == 1e-5
    assert (o.waves_per_chain, o.chains_per_wg, o.alpha_split) == (0, 0, 0)      # all automatic
    assert o.stop_estimate == 1
    assert o.precision == device.PRECISION_F64 and o.wg_per_cu == 0
    assert o.chi2_factor == 1.0
    # the struct of include/maxent_hip.h: 2 int32, 8 double, 6 int32, 1 double, 2 int32 (lds_basis, reserved), no padding holes
    import ctypes
    assert ctypes.sizeof(device.MxeOpts) == 2 * 4 + 8 * 8 + 6 * 4 + 8 + 2 * 4
    assert device.MxeOpts.precision.offset == 2 * 4 + 8 * 8 + 4 * 4
    assert device.MxeOpts.chi2_factor.offset == 2 * 4 + 8 * 8 + 6 * 4
    with pytest.raises(TypeError):
        device.default_opts(nonsense=1)


def test_minimizer_precision_option_reaches_the_opts():
    assert mx.LevenbergMinimizer().to_opts().precision == device.PRECISION_F64
    assert mx.LevenbergMinimizer(precision='f32').to_opts().precision == device.PRECISION_F32
    with pytest.raises(ValueError):
        mx.LevenbergMinimizer(precision='f16')


def test_no_gpu_fails_loudly_not_silently():
    if device.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(device.MaxEntDeviceError):
        device.DeviceContext(np.eye(3), np.ones(3), np.eye(3))
    tm = mx.TauMaxEnt()
    tm.set_verbosity(mx.VerbosityFlags.Quiet)
    tau = np.linspace(0, 10, 20)
    tm.set_G_tau_data(tau, -0.5 * np.ones(20))
    tm.set_error(1e-3)
    with pytest.raises(device.MaxEntDeviceError):
        tm.run()          # no CPU fallback


# ---- host mirror of the reference interface ------------------------------------
def test_meshes_default_model_kernel_match_reference_bitwise():
    g = load('cfg1_normal')
    w = mx.HyperbolicOmegaMesh(-10, 10, 200)
    np.testing.assert_array_equal(np.asarray(w), g['omega'])
    np.testing.assert_array_equal(w.delta, g['delta'])
    np.testing.assert_array_equal(mx.FlatDefaultModel(w).D, g['D'])
    K = mx.TauKernel(g['tau'], w, 40.0)
    Kref, Kdref = R.tau_kernel(g['tau'], g['omega'], 40.0)
    np.testing.assert_array_equal(K.K, Kref)
    np.testing.assert_array_equal(K.K_delta, Kdref)
    np.testing.assert_allclose(np.asarray(mx.LogAlphaMesh(1e-2, 1e4, 20)) * 100, g['alpha'], rtol=1e-15)
    s = load('kat_srvo3')
    np.testing.assert_allclose(np.asarray(mx.LorentzianOmegaMesh(-15, 15, 500)), s['omega'], atol=1e-13)
    assert list(mx.LinearAlphaMesh(1, 3, 3)) == [3.0, 2.0, 1.0]
    assert list(mx.DataAlphaMesh([1.0, 5.0, 2.0])) == [5.0, 2.0, 1.0]


def test_svd_staging_and_truncation():
    """reference test/python/tau_kernel.py:64-81."""
    g = load('cfg1_normal')
    K = mx.TauKernel(g['tau'], mx.DataOmegaMesh(g['omega']), 40.0)
    assert np.max(np.abs(np.dot(K.U * K.S, K.V.T) - K.K)) < 1e-13
    n_full = len(K.S)
    thr = np.median(K.S)
    K.reduce_singular_space(thr)
    assert len(K.S) == (n_full + 1) // 2 and K.U.shape[1] == len(K.S) == K.V.shape[1]
    K.reduce_singular_space(1e-14)           # smaller threshold: recomputes the SVD
    assert len(K.S) == len(g['S'])
    np.testing.assert_allclose(K.S, g['S'], rtol=1e-10, atol=1e-16)


def test_preblur_matrix_matches_reference():
    g = load('cfg5_preblur_pm')
    w = mx.DataOmegaMesh(g['omega'])
    np.testing.assert_allclose(mx.get_preblur(w, float(g['preblur_b'])), g['B'], rtol=1e-13)
    K = mx.PreblurKernel(mx.TauKernel(g['tau'], w, 40.0), float(g['preblur_b']))
    Kref = R.preblur_kernel(R.tau_kernel(g['tau'], g['omega'], 40.0)[0], g['omega'], float(g['preblur_b']))[0]
    np.testing.assert_allclose(K.K, Kref, rtol=1e-13)
    np.testing.assert_array_equal(K.K_delta, K.kernel.K_delta)


def test_covariance_rotation_of_the_facade():
    g = load('cov')
    tm = mx.TauMaxEnt()
    tm.omega = mx.DataOmegaMesh(g['omega'])
    tm.set_G_tau_data(g['tau'], g['G_orig'])
    tm.set_cov(g['cov'])
    np.testing.assert_allclose(np.abs(tm.G), np.abs(g['G_rot']), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(tm.err, g['err_rot'], rtol=1e-10)
    assert tm.K.K.shape == g['K_rot'].shape
    spec = tm.maxent_loop.make_spec()
    assert spec['U_rot'] is not None and spec['U_rot'].shape[0] == len(g['G_rot'])
    np.testing.assert_allclose(spec['alpha'], np.asarray(tm.alpha_mesh) * len(g['G_rot']))
    tm.set_error(1e-3)                      # undoes the rotation
    assert tm.K._T is None and len(tm.G) == len(g['G_orig'])
    np.testing.assert_allclose(tm.G, g['G_orig'], rtol=1e-12, atol=1e-15)


def test_start_vector_has_the_reference_quirk():
    """v0 = H_of_v.inv(D * delta): delta applied twice (maxent_loop.py:196-203)."""
    g = load('cfg1_plusminus')
    from maxent_amd import hostprep
    for kind, ent in ((device.ENTROPY_NORMAL, 'normal'), (device.ENTROPY_PLUSMINUS, 'plusminus')):
        p = R.Problem(np.zeros((len(g['G']), len(g['D']))), g['U'], g['S'], g['V'], g['G'], g['err'],
                      g['D'], entropy=ent)
        np.testing.assert_allclose(hostprep.initial_v(g['V'], g['D'], g['delta'], kind),
                                   R.initial_v(p, g['delta']), rtol=1e-13, atol=1e-15)


def test_attribute_shadowing_and_errors():
    tm = mx.TauMaxEnt(cost_function='bryan')
    assert isinstance(tm.cost_function, mx.BryanCostFunction)
    with pytest.raises(Exception):
        mx.TauMaxEnt(cost_function='nonsense')
    w = mx.LinearOmegaMesh(-5, 5, 30)
    tm.omega = w
    assert tm.K.K.shape[1] == 30 and len(tm.D.D) == 30 and tm.maxent_loop.omega is w
    tm.alpha_mesh = mx.LogAlphaMesh(0.1, 10, 4)
    assert len(tm.maxent_loop.alpha_mesh) == 4
    with pytest.raises(Exception):
        tm.set_error(np.ones(3))             # wrong length
    ew = mx.ElementwiseMaxEnt()
    ew.omega = w
    assert ew.maxent_diagonal.omega is w and ew.maxent_offdiagonal.omega is w
    assert ew.maxent_offdiagonal.cost_function.entropy_kind == device.ENTROPY_PLUSMINUS
    assert ew.maxent_diagonal.cost_function.entropy_kind == device.ENTROPY_NORMAL
    ew.set_G_tau_data(np.linspace(0, 1, 5), np.zeros((3, 3, 5)))
    assert ew.shape == (3, 3)
    ew.set_error(0.1)
    assert ew.get_error((0, 1)) == 0.1
    ew.set_error(np.ones((3, 3, 5)) * np.arange(3)[:, None, None])
    assert np.all(ew.get_error((2, 1)) == 2.0)
    with pytest.raises(TypeError):
        mx.DiagonalMaxEnt().run_offdiagonal()
    m = mx.LevenbergMinimizer(marquardt=True, J_squared=True)      # accepted: same minimum, other iterates
    assert m.marquardt and m.J_squared


def test_solver_details_callback_follows_the_verbosity():
    # reference maxent_loop.py:375-382
    loop = mx.MaxEntLoop()
    assert loop.minimizer.verbose_callback is None
    loop.set_verbosity(add=mx.VerbosityFlags.SolverDetails)
    assert loop.minimizer.verbose_callback == loop.logtaker.solver_verbose_callback
    loop.set_verbosity(remove=mx.VerbosityFlags.SolverDetails)
    assert loop.minimizer.verbose_callback is None


def test_minimizer_options_map_to_kernel_options():
    m = mx.LevenbergMinimizer(convergence=mx.MaxDerivativeConvergenceMethod(1e-4) |
                              mx.RelativeFunctionChangeConvergenceMethod(1e-16), maxiter=77)
    o = m.to_opts()
    assert (o.maxiter, o.tol_d, o.tol_relq, o.tol_h) == (77, 1e-4, 1e-16, 0.0)
    o = mx.LevenbergMinimizer().to_opts(waves_per_chain=2)
    assert o.tol_h == 1e-9 and o.tol_d == 0.0 and o.waves_per_chain == 2


def test_analyzers_pick_like_the_reference():
    """alpha_index / A_out of LineFit, Chi2Curvature, Entropy on the
    reference's own chi2(alpha), S(alpha), A(alpha)."""
    for name in ('cfg1_normal', 'cfg1_bryan', 'cfg1_plusminus', 'cfg1_tauerr', 'cfg2_normal'):
        g = load(name)
        if g['A_ref'].shape[0] != len(g['alpha']):
            continue
        res = mx.MaxEntResult()
        X = len(g['alpha'])
        res.add_element_results(dict(alpha=g['alpha'], v=np.zeros((X, 2)), H=g['H_ref'], A=g['A_ref'],
                                     chi2=g['chi2_ref'], S=g['S_ref'], Q=g['Q_ref'], G=g['G'],
                                     G_orig=g['G'], data_variable=g['tau'], G_rec=np.zeros((X, len(g['G']))),
                                     omega=mx.DataOmegaMesh(g['omega']), probability=np.full(X, np.nan)))
        res.analyze([mx.LineFitAnalyzer(), mx.Chi2CurvatureAnalyzer(), mx.EntropyAnalyzer(),
                     mx.BryanAnalyzer(), mx.ClassicAnalyzer()])
        ar = res.analyzer_results
        assert ar['LineFitAnalyzer']['alpha_index'] == int(g['linefit_alpha_index'])
        assert ar['Chi2CurvatureAnalyzer']['alpha_index'] == int(g['chi2curv_alpha_index'])
        np.testing.assert_array_equal(ar['LineFitAnalyzer']['A_out'], g['A_out_linefit'])
        np.testing.assert_array_equal(ar['Chi2CurvatureAnalyzer']['A_out'], g['A_out_chi2curv'])
        np.testing.assert_array_equal(ar['EntropyAnalyzer']['A_out'], g['A_out_entropy'])
        assert 'A_out' not in ar['BryanAnalyzer']       # no probability -> info only
        res._default_analyzer_name = 'LineFitAnalyzer'
        np.testing.assert_array_equal(res.A_out, g['A_out_linefit'])


def test_linefit_equals_polyfit_formulation():
    rng = np.random.RandomState(3)
    x = np.log(np.logspace(3, -1, 40))
    y = np.log(50 + np.exp(1.3 * x + 2) + 0.01 * rng.rand(40))
    idx, (p1, p2) = mx.analyzers.fit_piecewise(x, y)
    # brute force with np.polyfit like the reference (linefit_analyzer.py:28-87)
    best = None
    for i in range(2, 38):
        c1, r1 = np.polyfit(x[:i], y[:i], 1, full=True)[:2]
        c2, r2 = np.polyfit(x[i:], y[i:], 0, full=True)[:2]
        tot = (r1[0] if len(r1) else 0.0) + (r2[0] if len(r2) else 0.0)
        if best is None or tot < best[0]:
            best = (tot, c1, c2)
    np.testing.assert_allclose(p1, best[1], rtol=1e-8)
    np.testing.assert_allclose(p2, best[2], rtol=1e-8)
    xc = (best[2][0] - best[1][1]) / best[1][0]
    assert idx == int(np.argmin(np.abs(x - xc)))


def test_matrix_result_layout_and_hermiticity():
    """reference test/python/matrix_maxent_result.py:60-126."""
    X, W, T, Sn = 4, 6, 5, 3
    om = mx.LinearOmegaMesh(-1, 1, W)

    def rec(scale, X=X):
        return dict(alpha=np.logspace(1, -1, X), v=np.ones((X, Sn)) * scale, H=np.ones((X, W)) * scale,
                    A=np.ones((X, W)) * scale, chi2=np.ones(X) * scale, S=-np.ones(X), Q=np.ones(X),
                    G=np.ones(T), G_orig=np.ones(T), data_variable=np.arange(T, dtype=float),
                    G_rec=np.ones((X, T)), omega=om, probability=np.full(X, np.nan))
    res = mx.MaxEntResult(matrix_structure=(2, 2), use_hermiticity=True)
    with pytest.raises(AssertionError):
        res.add_element_results(rec(1.0))                 # matrix_element missing
    res.add_element_results(rec(1.0), (0, 0))
    res.add_element_results(rec(2.0, X=3), (0, 1))        # fewer alphas -> NaN padded
    assert res.A.shape == (2, 2, X, W) and res.v.shape == (2, 2, X, Sn)
    assert res.chi2.shape == (2, 2, X) and res.G.shape == (2, 2, T) and res.G_rec.shape == (2, 2, X, T)
    assert np.all(np.isnan(res.chi2[1, 1])) and np.isnan(res.chi2[0, 1, 3]) and res.chi2[0, 1, 2] == 2.0
    np.testing.assert_array_equal(res.A[1, 0], res.A[0, 1])       # hermitian mirror for A and H only
    assert np.all(np.isnan(res.chi2[1, 0]))
    np.testing.assert_array_equal(res._n_alphas, [[4, 3], [0, 0]])
    res.analyze([mx.EntropyAnalyzer()], (0, 0))
    res._default_analyzer_name = 'EntropyAnalyzer'
    res._zero_elements.append((1, 1))
    A_out = res.A_out
    assert A_out.shape == (2, 2, W) and np.all(A_out[1, 1] == 0.0) and np.all(np.isnan(A_out[0, 1]))
    cres = mx.MaxEntResult(matrix_structure=(2, 2), complex_elements=True)
    cres.add_element_results(rec(1.0), (0, 1), 0)
    cres.add_element_results(rec(3.0), (0, 1), 1)
    assert cres.A.shape == (2, 2, 2, X, W)
    np.testing.assert_array_equal(cres.A[1, 0, 1], -cres.A[0, 1, 1])   # conjugate


def test_probability_from_device_logdet_equals_host_evaluation():
    """NormalLogProbability.from_logdet (fed by mxe_logdet on the device) and the host
    evaluation are the same formula."""
    from maxent_amd.probabilities import NormalLogProbability
    rng = np.random.RandomState(3)
    n_tau, n_omega, n_s, X = 30, 50, 9, 4
    U = np.linalg.qr(rng.randn(n_tau, n_s))[0]
    V = np.linalg.qr(rng.randn(n_omega, n_s))[0]
    S = 10.0 ** (-np.arange(n_s))
    err = 1e-3 * (1 + rng.rand(n_tau))
    alpha = np.array([50.0, 5.0, 0.5, 0.05])
    w = rng.rand(X, n_omega) + 0.1
    Q = rng.randn(X)
    C = (U * S[None, :]) / err[:, None]
    M = C.T @ C
    logdet = np.array([np.linalg.slogdet(np.eye(n_s) + M @ ((V.T * w[i]) @ V) / alpha[i])[1] for i in range(X)])
    for kw in (dict(), dict(log_prior_alpha=lambda a: -2 * np.log(a)),
               dict(log_norm_S=lambda a, n: 0.25 * n * np.log(a))):
        p = NormalLogProbability(**kw)
        np.testing.assert_allclose(p.from_logdet(logdet, alpha, Q, n_omega),
                                   p.evaluate(U, S, V, err, alpha, w, Q), rtol=1e-12, atol=1e-12)


def test_fast_linefit_picks_the_index_of_the_plain_loop():
    """fit_piecewise selects candidates from prefix sums and evaluates them exactly; the picked
    break point and the returned polynomials are those of the O(n^2) loop."""
    from maxent_amd import analyzers as an
    rng = np.random.RandomState(11)

    def plain(logx, logy, deg):
        n = len(logx)
        ok = ~np.isnan(logy)
        best, bi, bp = np.inf, None, None
        for i in range(2, n - 2):
            x1, y1, x2, y2 = logx[:i][ok[:i]], logy[:i][ok[:i]], logx[i:][ok[i:]], logy[i:][ok[i:]]
            if len(x1) < 1 or len(x2) < 1:
                continue
            s1, c1, e1 = an._linfit_sse(x1, y1)
            if deg == 1:
                s2, c2, e2 = an._linfit_sse(x2, y2)
                q2 = np.array([s2, c2])
            else:
                c2 = float(np.mean(y2)); e2 = float(np.sum((y2 - c2) ** 2)); q2 = np.array([c2])
            if e1 + e2 < best:
                best, bi, bp = e1 + e2, i, (np.array([s1, c1]), q2)
        return bi, bp
    for trial in range(60):
        n = rng.randint(9, 120)
        logx = np.sort(rng.uniform(-5, 10, n))[::-1].copy()
        k = rng.randint(3, n - 3)
        logy = np.where(np.arange(n) < k, 2.0 + 0.8 * (logx - logx[k]), 2.0) + 0.05 * rng.randn(n)
        if trial % 3 == 0:
            logy[rng.randint(0, n, 3)] = np.nan
        for deg in (0, 1):
            bi, bp = plain(logx, logy, deg)
            idx, (q1, q2) = an.fit_piecewise(logx, logy, deg)
            np.testing.assert_array_equal(q1, bp[0])
            np.testing.assert_array_equal(q2, bp[1])


def test_one_covariance_for_many_elements_is_decomposed_once():
    """set_cov with the SAME matrix object again (ElementwiseMaxEnt with one (T, T) covariance does that per
    element, reference elementwise_maxent.py:502-515): same rotation object -- one data set on the device --
    and the data the reference's bookkeeping gives (tau_maxent.py:253-325), as without the shortcut"""
    rng = np.random.default_rng(1)
    tau = np.linspace(0, 10, 20)
    G = -np.exp(-tau) - 0.5 * np.exp(-(10 - tau))
    A = rng.standard_normal((20, 20))
    cov = np.dot(A, A.T) * 1e-6 + 1e-6 * np.eye(20)

    def states(seq):
        tm = mx.TauMaxEnt()
        tm.set_verbosity(mx.VerbosityFlags.Quiet)
        tm.omega = mx.HyperbolicOmegaMesh(-5, 5, 30)
        out = []
        for g, c in seq:
            tm.set_G_tau_data(tau, g)
            tm.set_cov(c)
            out.append((np.array(tm.G), np.array(tm.err), tm.K.rotation, np.array(tm.K.K), tm.K.U))
        return out
    same = states([(G, cov), (1.1 * G, cov)])
    copy = states([(G, cov), (1.1 * G, cov.copy())])
    assert same[1][2] is same[0][2] and same[1][4] is same[0][4]          # rotation and U untouched
    assert copy[1][2] is not copy[0][2]
    for x, y in zip(same, copy):
        np.testing.assert_array_equal(x[0], y[0])
        np.testing.assert_array_equal(x[1], y[1])
        np.testing.assert_allclose(x[3], y[3], rtol=0, atol=1e-13)


def test_logtaker_reproduces_the_reference_test_output(tmp_path, capsys):
    """reference test/python/logtaker.py:25-58 against its own logtaker.ref (terminal) and logtaker.dat.ref (log file)"""
    from maxent_amd.logtaker import Logtaker, VerbosityFlags as F
    levels = [F.Quiet, F.Header, F.ElementInfo, F.Timing, F.AlphaLoop, F.SolverDetails, F.Errors, F.Header | F.Timing, F.Default]
    log = Logtaker()
    logfile = str(tmp_path / 'logtaker.dat')
    log.open_logfile(logfile, False)
    for i, level in enumerate(levels):
        log.verbose = level
        log.message(F.Quiet, "=== Test #{} ===", i)
        log.message(F.Header, "This is a header message.")
        log.message(F.ElementInfo, "This is an element info message.")
        log.message(F.Timing, "This is a timing message.")
        log.message(F.AlphaLoop, "This is an alpha loop message.")
        log.message(F.SolverDetails, "This is a solver details message.")
        log.error_message("This is an error message")
        log.message(F.Timing | F.Header, "This is a header + timing message.")
    log.close_logfile()
    shown = capsys.readouterr().out
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    with open(os.path.join(gold, 'logtaker.ref'), newline='') as f:
        assert shown == f.read()
    with open(os.path.join(gold, 'logtaker.dat.ref'), newline='') as f, open(logfile, newline='') as mine:
        assert mine.read() == f.read()
    assert log.get_error_messages() == ['This is an error message'] * len(levels)


def test_cached_methods_of_a_user_function_follow_the_reference_protocol():
    """reference test/python/cached_view.py:27-140: evaluation counts of a user-written DoublyDerivableFunction
    with @cached methods, unpinned and pinned"""
    from maxent_amd.functions import DoublyDerivableFunction, cached
    n = dict(f=0, d=0, dd=0, sq=0)

    class Sin(DoublyDerivableFunction):
        @cached
        def f(self, v):
            n['f'] += 1
            return np.sin(v)[0]

        @cached
        def d(self, v):
            n['d'] += 1
            return np.cos(v)

        @cached
        def dd(self, v):
            n['dd'] += 1
            return -np.diag(np.sin(v))

        @cached
        def square(self, v):
            n['sq'] += 1
            return v ** 2

        @cached
        def fourth1(self, v):
            return self.square(v) * self.square(v)

        @cached
        def fourth2(self, v):
            return self.square(self.square(v))

    sin = Sin()
    v = np.random.RandomState(3).rand(1,)
    f1, d1, dd1 = sin.f(v), sin.d(v), sin.dd(v)
    assert (n['f'], n['d'], n['dd']) == (1, 1, 1)
    f1 = sin.f(v)                                 # not pinned: evaluated again
    assert (n['f'], n['d'], n['dd']) == (2, 1, 1)
    sin1 = sin(v)
    f2, d2, dd2 = sin1.f(), sin1.d(), sin1.dd()
    f2, d2 = sin1.f(), sin1.d()                   # remembered
    assert f1 == f2 and d1 == d2 and dd1 == dd2
    assert (n['f'], n['d'], n['dd']) == (3, 2, 2)
    sin1.f(2 * v)                                 # another argument: evaluated, the memo is untouched
    assert sin1.f() == f1 and n['f'] == 4
    sin1.fourth2(); sin1.fourth2(); sin1.fourth1(); sin1.fourth1()
    assert n['sq'] == 2
    v[0] += 0.01
    sin1.f(v)
    assert n['f'] == 5


def test_elementwise_inputs_from_files_and_arrays(tmp_path):
    """reference test/python/elementwise_set_G.py:62-92: file-name pattern, array of file names, one array"""
    from itertools import product
    em = mx.ElementwiseMaxEnt()
    tau = np.linspace(0, 40, 10)
    rng = np.random.RandomState(5)
    G_elems = {}
    for element in product(range(2), range(2)):
        G_elems[element] = rng.rand(len(tau))
        np.savetxt(str(tmp_path / 'g_{}_{}.dat'.format(*element)), np.column_stack((tau, G_elems[element])))

    def check():
        for element in product(range(2), range(2)):
            em.set_G_element(em.maxent_diagonal, em.G_mat, element, True)
            assert np.max(np.abs(em.maxent_diagonal.G - G_elems[element])) < 1e-13
    em.set_G_tau_filename_pattern(str(tmp_path / 'g_{i}_{j}.dat'), (2, 2))
    check()
    em.set_G_tau_filenames([[str(tmp_path / 'g_{}_{}.dat'.format(i, j)) for j in range(2)] for i in range(2)])
    check()
    em.set_G_tau_data(tau, np.array([[G_elems[i, j] for j in range(2)] for i in range(2)]))
    check()


def test_the_reference_import_paths_exist():
    """scripts written against the reference import from its sub-packages (doc/guide/preblur_example.py,
    test/python/*.py); the same paths under this package give the same objects"""
    from maxent_amd.analyzers.linefit_analyzer import fit_piecewise, LineFitAnalyzer
    from maxent_amd.analyzers.chi2_curvature_analyzer import curv, Chi2CurvatureAnalyzer
    from maxent_amd.analyzers.analyzer import Analyzer, AnalyzerResult
    from maxent_amd.cost_functions.maxent_cost_function import MaxEntCostFunction
    from maxent_amd.cost_functions.bryan_cost_function import BryanCostFunction
    from maxent_amd.minimizers.levenberg_minimizer import LevenbergMinimizer
    from maxent_amd.minimizers.convergence_methods import MaxDerivativeConvergenceMethod
    from maxent_amd.triqs_support import if_no_triqs, if_triqs_1, if_triqs_2, assert_text_files_equal
    assert LineFitAnalyzer is mx.LineFitAnalyzer and Chi2CurvatureAnalyzer is mx.Chi2CurvatureAnalyzer
    assert MaxEntCostFunction is mx.MaxEntCostFunction and BryanCostFunction is mx.BryanCostFunction
    assert LevenbergMinimizer is mx.LevenbergMinimizer and Analyzer is mx.Analyzer and AnalyzerResult is mx.AnalyzerResult
    assert MaxDerivativeConvergenceMethod is mx.MaxDerivativeConvergenceMethod
    assert if_no_triqs() and not if_triqs_1() and not if_triqs_2()
    assert callable(fit_piecewise) and callable(curv) and callable(assert_text_files_equal)


def test_bench_reads_counters_only_from_a_profile_of_the_same_build(tmp_path, monkeypatch):
    """VERDICT r02 / ADVICE: roofline.achieved must not inherit counters of another build.  bench.load_pmc takes a
    profiles/*_pmc_summary.csv only when its first line records the source hash of the library being benched."""
    import bench
    prof = tmp_path / 'profiles'
    prof.mkdir()
    body = 'counter,dispatches,mean_per_dispatch,min,max\n' + ''.join(
        '%s,10,%g,1,2\n' % (c, 100.0 + i) for i, c in enumerate(bench.PMC_NAMES.values()))
    (prof / 'r01_x_pmc_summary.csv').write_text(body)                                  # no hash line: never applied
    (prof / 'r03_a_pmc_summary.csv').write_text('#source_hash,0123456789abcdef\n' + body)
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    pmc, why = bench.load_pmc('0123456789abcdef')
    assert why is None and pmc['source'].endswith('r03_a_pmc_summary.csv')
    assert pmc['valu_active_quadcycles'] == 100.0 + list(bench.PMC_NAMES).index('valu_active_quadcycles')
    pmc, why = bench.load_pmc('ffffffffffffffff')                                      # a kernel edit without re-profiling
    assert pmc is None and 'ffffffffffffffff' in why and '0123456789abcdef' in why
    (prof / 'r03_b_pmc_summary.csv').write_text('#source_hash,ffffffffffffffff\ncounter,dispatches,mean_per_dispatch,min,max\nFETCH_SIZE,1,2,2,2\n')
    pmc, why = bench.load_pmc('ffffffffffffffff')
    assert pmc is None and 'lacks the counters' in why


def test_library_reports_the_hash_of_its_sources():
    """mxe_source_hash: what the Makefile hashed when it built the binary (sources in the order of its HDR list)"""
    import hashlib
    csrc = os.path.join(ROOT, 'maxent_amd', 'csrc')
    mk = open(os.path.join(csrc, 'Makefile')).read()
    hdr = [ln for ln in mk.splitlines() if ln.startswith('HDR')][0].split(':=')[1].split()
    data = b''.join(open(os.path.join(csrc, f), 'rb').read() for f in ['maxent_hip.hip'] + hdr)
    assert device.source_hash() == hashlib.sha256(data).hexdigest()[:16]


def test_set_cov_and_the_start_vector_see_in_place_edits():
    """caches keyed on object identity alone kept stale values after ``cov *= 4`` / ``A_init[:] = ...``"""
    tau, omega, K, G = synthetic.single_G(30, 40)
    tm = mx.TauMaxEnt()
    tm.set_verbosity(mx.VerbosityFlags.Quiet)
    tm.omega = omega
    tm.set_G_tau_data(tau, G)
    rng = np.random.RandomState(3)
    L = 1e-3 * (np.eye(30) + 0.1 * rng.randn(30, 30))
    cov = L @ L.T
    tm.set_cov(cov)
    e1 = np.array(tm.err)
    tm.set_cov(cov)
    assert np.array_equal(tm.err, e1)
    cov *= 4.0
    tm.set_cov(cov)
    assert np.allclose(tm.err, 2.0 * e1, rtol=1e-10)
    tm.set_error(1e-3)
    tm.A_init = np.array(tm.D.D) / omega.delta
    v_a = tm.maxent_loop.make_spec()['v0'] if hasattr(tm, 'maxent_loop') else tm.make_spec()['v0']
    tm.A_init[:] = tm.A_init * np.exp(-np.asarray(omega) ** 2 / 50.0)
    v_b = tm.maxent_loop.make_spec()['v0'] if hasattr(tm, 'maxent_loop') else tm.make_spec()['v0']
    assert np.max(np.abs(v_a - v_b)) > 1e-3


def test_analyzer_results_are_built_when_looked_at():
    """round 3: the analyzers' alphas come from the device and nothing is built that nobody reads -- AnalyzerResult
    computes its costly entries on first access, ElementAnalysis turns a Deferred pick into a result object when it is
    read, and both behave like the plain dicts of the reference (analyzers/analyzer.py:25-46) for everything else."""
    import pickle
    from maxent_amd.analyzers import AnalyzerResult, Picks, Deferred
    from maxent_amd.maxent_result import ElementAnalysis
    calls = []
    r = AnalyzerResult()
    r['alpha_index'] = 3
    r.later('curve', lambda: calls.append('curve') or np.arange(4.0))
    assert 'curve' in r and calls == []
    assert r['alpha_index'] == 3 and calls == []
    assert np.array_equal(r['curve'], np.arange(4.0)) and calls == ['curve']
    r['curve']
    assert calls == ['curve']                                   # once
    r2 = AnalyzerResult()
    r2.later('x', lambda: 7)
    assert sorted(r2.keys()) == ['x'] and r2.get('y', 5) == 5 and len(r2) == 1 and dict(r2) == {'x': 7}
    r3 = AnalyzerResult()
    r3['name'] = 'A'
    r3.later('info', lambda: 'text')
    back = pickle.loads(pickle.dumps(r3))
    assert dict(back) == {'name': 'A', 'info': 'text'}

    class FakeAnalyzer(object):
        name = 'LineFitAnalyzer'

    class FakeResult(object):
        alpha = np.array([8.0, 4.0, 2.0, 1.0])
    res = FakeResult()
    built = []
    picks = Picks(FakeAnalyzer(), res, [(0, 0), (0, 1)], [2, 1], [np.ones(3), 2 * np.ones(3)], dict(linefit_deg=0),
                  dict(linefit_params=lambda owner, k, alpha: built.append(k) or ('p', k)),
                  'Ideal alpha (linefit): {} (= index {} zero-based)')
    out = ElementAnalysis()
    from collections import OrderedDict
    OrderedDict.__setitem__(out, 'LineFitAnalyzer', Deferred(picks, 1))
    out['Other'] = 'chi2 is all NaN'
    assert isinstance(out.raw('LineFitAnalyzer'), Deferred) and np.array_equal(out.raw('LineFitAnalyzer').A_out(), 2 * np.ones(3))
    one = out['LineFitAnalyzer']                                # built now
    assert isinstance(one, AnalyzerResult) and one['alpha_index'] == 1 and one['name'] == 'LineFitAnalyzer'
    assert one['info'] == 'Ideal alpha (linefit): 4.0 (= index 1 zero-based)' and built == []
    assert one['linefit_params'] == ('p', (0, 1)) and built == [(0, 1)]
    assert out['LineFitAnalyzer'] is one and list(out.keys()) == ['LineFitAnalyzer', 'Other']
    assert pickle.loads(pickle.dumps(out))['Other'] == 'chi2 is all NaN'


def test_staging_is_skipped_by_contents_and_shared_arrays_are_compared_as_one_row():
    """BatchSolver._stage (no device: a stand-in context that counts the uploads): the same job again uploads nothing, an
    edit of ONE element's data or of the shared default model uploads again; arrays that every element shares (the
    element-wise drivers hand out one D, alpha, v0, error array) are held as one row"""
    from maxent_amd.batch_solver import BatchSolver
    from maxent_amd import device

    class Ctx(object):
        _n_chain = 0
        calls = 0

        def clear_datasets(self):
            pass

        def add_dataset(self, *a):
            return 0

        def set_elements(self, ds, G, D, kinds):
            self.G, self.D = np.array(G), np.array(D)

        def upload_chains(self, el, alpha, v0, opts):
            self.calls += 1
            self._n_chain = len(el)
            self.alpha, self.v0 = np.array(alpha), np.array(v0)

        data_updates = 0

        def update_data(self, G):
            # (mxe_elements_update_data: new data for the elements that are set, chains stay ready)
            self.data_updates += 1
            self.G = np.array(G)

    class K(object):
        rotation, U = None, None
    n, rng = 5, np.random.RandomState(0)
    D, al, v0, err = rng.rand(30), rng.rand(7), np.zeros(4), 1e-3 * np.ones(12)
    specs = [dict(G=rng.rand(12), err=err, D=D, alpha=al, v0=v0, kind=device.ENTROPY_NORMAL, U_rot=None) for _ in range(n)]
    rows = BatchSolver._rows_of(specs, 'D')
    assert rows.shape == (1, 30)                               # shared: one row
    assert BatchSolver._rows_of([dict(D=D), dict(D=D.copy())], 'D').shape == (1, 30)       # two workers' copies of one default model: one row
    D2 = 2.0 * D
    mixed = BatchSolver._rows_of([dict(D=D), dict(D=D2), dict(D=D), dict(D=D2.copy())], 'D')      # a few distinct objects: rows by object
    assert mixed.shape == (4, 30) and np.array_equal(mixed[0], D) and np.array_equal(mixed[1], D2) and \
        np.array_equal(mixed[2], D) and np.array_equal(mixed[3], D2)
    many = [dict(D=rng.rand(30)) for _ in range(12)]                                              # more than a handful: the general way
    assert np.array_equal(BatchSolver._rows_of(many, 'D'), np.array([m['D'] for m in many]))
    bs = BatchSolver.__new__(BatchSolver)
    ctx, opts = Ctx(), device.default_opts()
    bs._stage(ctx, K, specs, opts)
    assert ctx.calls == 1 and ctx.G.shape == (n, 12) and ctx.D.shape == (n, 30) and ctx.alpha.shape == (n, 7)
    assert np.array_equal(ctx.D[3], D) and np.array_equal(ctx.G[2], specs[2]['G'])
    bs._stage(ctx, K, [dict(s) for s in specs], opts)          # other dicts, same contents
    assert ctx.calls == 1
    again = [dict(s) for s in specs]
    again[3]['G'] = again[3]['G'] + 1e-9
    bs._stage(ctx, K, again, opts)
    # new DATA only: their projections are updated, the chains (cut, start states: they do not depend on G) stay (round 5)
    assert ctx.calls == 1 and ctx.data_updates == 1 and np.array_equal(ctx.G[3], again[3]['G'])
    bs._stage(ctx, K, [dict(s) for s in again], opts)
    assert ctx.calls == 1 and ctx.data_updates == 1
    D[5] *= 2.0                                                # the shared default model edited in place
    bs._stage(ctx, K, again, opts)
    assert ctx.calls == 2 and ctx.D[0, 5] == D[5]
    own = [dict(s, D=D.copy()) for s in again]                 # every element its own copy, equal contents: still one row, nothing staged
    bs._stage(ctx, K, own, opts)
    assert ctx.calls == 2
    own[2]['D'] = own[2]['D'] * 1.5                            # one of them differs: rows per element, staged again
    bs._stage(ctx, K, own, opts)
    assert ctx.calls == 3 and ctx.D.shape == (5, 30) and ctx.D[2, 0] == 1.5 * D[0] and ctx.D[1, 0] == D[0]
    bs._stage(ctx, K, [dict(s) for s in own], opts)
    assert ctx.calls == 3
    opts2 = device.default_opts(); opts2.maxiter = opts.maxiter + 1
    bs._stage(ctx, K, own, opts2)
    assert ctx.calls == 4
    # the same job given as ARRAYS (ElementwiseMaxEnt on array input, round 5) is recognised as what is staged, and the other way
    # round; new data rows go through update_data there as well
    Gm = np.stack([s['G'] for s in own])
    arrays = dict(n=n, G=Gm, err=err.reshape(1, -1), sel=np.array([0, 0, 1, 0, 0]),
                  D=np.stack([own[0]['D'], own[2]['D']]), alpha=np.stack([al, al]), v0=np.stack([v0, v0]),
                  kinds=np.array([device.ENTROPY_NORMAL, device.ENTROPY_NORMAL]))
    bs._stage_arrays(ctx, K, arrays, opts2)
    assert ctx.calls == 4 and ctx.data_updates == 1
    bs._stage_arrays(ctx, K, dict(arrays, G=Gm + 1e-9), opts2)
    assert ctx.calls == 4 and ctx.data_updates == 2 and np.array_equal(ctx.G, Gm + 1e-9)
    bs._stage(ctx, K, [dict(s, G=s['G'] + 1e-9) for s in own], opts2)
    assert ctx.calls == 4 and ctx.data_updates == 2
    bs._stage_arrays(ctx, K, dict(arrays, G=Gm, sel=np.array([0, 1, 1, 0, 0])), opts2)      # another default model for scan 1
    assert ctx.calls == 5 and ctx.D[1, 0] == 1.5 * D[0]


def test_records_and_picks_of_a_launch_are_made_in_one_go():
    """MaxEntLoop.make_records == make_record element by element; MaxEntResult.add_batch_results == add_element_results +
    timings; analyzers._device_picks reads the launch's arrays directly and equals the per-element path"""
    from maxent_amd import synthetic
    from maxent_amd.maxent_loop import MaxEntLoop
    from maxent_amd.maxent_result import MaxEntResult
    from maxent_amd.analyzers import _device_picks
    from maxent_amd.batch_solver import LazyA
    tau, omega, K, Gmat, _ = synthetic.matrix_G(2, 20, 30)
    tm = mx.TauMaxEnt()
    tm.set_verbosity(mx.VerbosityFlags.Quiet)
    tm.set_G_tau_data(tau, Gmat[0, 0])
    tm.omega = omega
    tm.alpha_mesh = mx.LogAlphaMesh(alpha_min=1.0, alpha_max=100.0, n_points=4)
    tm.set_error(1e-3)
    loop = tm.maxent_loop
    rng = np.random.RandomState(1)
    n, X, nw = 3, 4, len(omega)
    specs = [dict(G=rng.rand(20), G_orig=rng.rand(20), data_variable=np.asarray(tau), D=np.ones(nw), kind=0) for _ in range(n)]
    sols = [dict(alpha=np.array([8.0, 4.0, 2.0, 1.0]), H=rng.rand(X, nw), A=None, v=rng.rand(X, 5), chi2=rng.rand(X), S=-rng.rand(X),
                 Q=rng.rand(X), n_iter=np.ones(X, int), converged=np.ones(X, bool), n_evals=np.ones(X, int)) for _ in range(n)]
    many = loop.make_records(specs, sols)
    for spec, sol, rec in zip(specs, sols, many):
        one = loop.make_record(spec, sol)
        assert set(one) == set(rec)
        for k in ('A', 'G', 'G_orig', 'chi2', 'omega', 'G_rec', 'probability'):
            assert np.array_equal(np.asarray(one[k]), np.asarray(rec[k]), equal_nan=True), k
    assert many[0]['probability'] is many[1]['probability'] and not many[0]['probability'].flags.writeable
    res = MaxEntResult(matrix_structure=(2, 2), element_wise=True, complex_elements=False, use_hermiticity=False)
    import datetime
    t0, t1 = datetime.datetime(2020, 1, 1), datetime.datetime(2020, 1, 1, 0, 0, 2)
    elements = [((0, 0), 0), ((0, 1), 0), ((1, 1), 0)]
    res._start.update(dict.fromkeys([res._key(e, c) for e, c in elements], t0))
    keys = res.add_batch_results(many, elements, t_end=t1)
    assert keys == [(0, 0), (0, 1), (1, 1)] and res._records[(0, 1)] is many[1]
    assert res._end[(1, 1)] - res._start[(1, 1)] == datetime.timedelta(seconds=2)
    # device picks: from the launch's arrays (batch) and element by element
    class Map(object):
        def matrix(self):
            return None

        def f(self, H):
            return np.asarray(H) / 0.5
    m = Map()
    idx = np.array([[1, 2, 0], [3, 3, 3], [0, 1, 2]], dtype=np.int32)
    rows = rng.rand(3, n, nw)
    params = (0, 0.2)
    for c, key in enumerate(keys):
        rec = res._records[key]
        rec['A'] = LazyA(rec['H'], m)
        rec['device_select'] = dict(params=params, index=idx[:, c], H=rows[:, c], batch=(idx, rows), chain=c)
    for which in (0, 1, 2):
        got = _device_picks(res, keys, which, lambda p: p[0] == 0)
        assert got[0] == idx[which].tolist()
        assert np.array_equal(np.array(got[1]), rows[which] / 0.5)
        for rec in res._records.values():
            rec['device_select'].pop('batch')
        slow = _device_picks(res, keys, which, lambda p: p[0] == 0)
        assert slow[0] == got[0] and np.array_equal(np.array(slow[1]), np.array(got[1]))
        for c, key in enumerate(keys):
            res._records[key]['device_select']['batch'] = (idx, rows)
    assert _device_picks(res, keys, 0, lambda p: p[0] == 1) is None            # another parameter than the device used
    idx[2, 1] = -1
    assert _device_picks(res, keys, 2, lambda p: True) is None                 # an element without a choice


def test_result_arrays_are_views_of_one_fetched_array_where_they_can_be():
    """MaxEntResult._assemble_whole (no device): records that are consecutive rows of one array, in the order of the matrix,
    assemble as a VIEW of it; records elsewhere in memory as one copy; a missing or shorter element goes the general way
    (NaN fill); A = H / delta of lazy records in one division, also when the two workers bring maps of their own"""
    from maxent_amd.maxent_result import MaxEntResult
    from maxent_amd.batch_solver import LazyA
    from maxent_amd.functions import IdentityA_of_H
    omega = mx.LinearOmegaMesh(-2, 2, 9)
    rng = np.random.RandomState(3)
    big = rng.rand(4, 5, 9)                                   # [element][alpha][omega], as it comes off the device
    alpha = np.array([8.0, 4.0, 2.0, 1.0, 0.5])

    def result(order=(0, 1, 2, 3), maps=None, drop=None, short=None):
        res = MaxEntResult(matrix_structure=(2, 2), element_wise=True, complex_elements=False, use_hermiticity=False)
        maps = maps or [IdentityA_of_H(omega)] * 4
        for n, key in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
            if drop == key:
                continue
            H = big[order[n]]
            if short == key:
                H = H[:3]
            res._records[key] = dict(alpha=alpha[:len(H)], H=H, A=LazyA(H, maps[n]), chi2=rng.rand(len(H)), omega=omega)
        return res
    res = result()
    assert np.shares_memory(res.H, big) and res.H.shape == (2, 2, 5, 9) and np.array_equal(res.H[1, 0], big[2])
    assert np.array_equal(res.A, big.reshape(2, 2, 5, 9) / omega.delta)
    assert all(r['A']._val is None for r in res._records.values())           # nothing was formed element by element
    assert res.chi2.shape == (2, 2, 5)
    two = result(maps=[IdentityA_of_H(omega), IdentityA_of_H(omega), IdentityA_of_H(omega), IdentityA_of_H(omega)])
    assert np.array_equal(two.A, res.A) and all(r['A']._val is None for r in two._records.values())
    other = result(maps=[IdentityA_of_H(omega)] * 3 + [IdentityA_of_H(mx.LinearOmegaMesh(-4, 4, 9))])
    assert np.array_equal(other.A[0, 1], res.A[0, 1]) and np.array_equal(other.A[1, 1], big[3] / mx.LinearOmegaMesh(-4, 4, 9).delta)
    perm = result(order=(2, 0, 3, 1))
    assert not np.shares_memory(perm.H, big) and np.array_equal(perm.H[0, 0], big[2]) and np.array_equal(perm.H[1, 1], big[1])
    gap = result(drop=(1, 0))
    assert np.all(np.isnan(gap.H[1, 0])) and np.array_equal(gap.H[1, 1], big[3])
    sh = result(short=(0, 1))
    assert np.array_equal(sh.H[0, 1][:3], big[1][:3]) and np.all(np.isnan(sh.H[0, 1][3:]))


def test_rows_of_a_non_default_analyzer_are_formed_when_first_read():
    """analyzers._device_picks with a source that is still 'on the device' (batch_solver.LazyRows stand-in): indices at once,
    the A rows (H / delta for everybody, or map by map) when somebody reads one -- and then only once"""
    from maxent_amd.analyzers import _device_picks, DeferredRows
    from maxent_amd.maxent_result import MaxEntResult
    from maxent_amd.batch_solver import LazyA, PickedRows

    class Lazy(object):
        fetched = 0

        def __init__(self, val):
            self._val, self.on_host = val, False

        def __array__(self, dtype=None, copy=None):
            Lazy.fetched += 1
            self.on_host = True
            return self._val

        def __getitem__(self, item):
            return np.asarray(self)[item]

    class Map(object):
        def matrix(self):
            return None

        def f(self, H):
            return np.asarray(H) / 0.25
    rng = np.random.RandomState(5)
    n, nw = 4, 7
    idx = np.array([[0, 1, 2, 1], [2, 2, 0, 1], [1, 0, 0, 2]], dtype=np.int32)
    rows = rng.rand(3, n, nw)
    srcs = [rows[0], Lazy(rows[1]), Lazy(rows[2])]           # analyzer 0 came with the solve
    m = Map()
    res = MaxEntResult(matrix_structure=(2, 2), element_wise=True, complex_elements=False, use_hermiticity=False)
    keys = [(0, 0), (0, 1), (1, 0), (1, 1)]
    batch = (idx, srcs, None)
    for c, key in enumerate(keys):
        H = rng.rand(3, nw)
        res._records[key] = dict(alpha=np.array([4.0, 2.0, 1.0]), H=H, A=LazyA(H, m), chi2=rng.rand(3),
                                 device_select=dict(params=(0, 0.2), index=idx[:, c], H=PickedRows(srcs, c), batch=batch, chain=c))
    got0 = _device_picks(res, keys, 0, lambda p: True)
    assert got0[0] == idx[0].tolist() and isinstance(got0[1], list) and np.array_equal(np.array(got0[1]), rows[0] / 0.25)
    got1 = _device_picks(res, keys, 1, lambda p: True)
    assert got1[0] == idx[1].tolist() and isinstance(got1[1], DeferredRows) and len(got1[1]) == n and Lazy.fetched == 0
    assert np.array_equal(got1[1][2], rows[1][2] / 0.25) and Lazy.fetched == 1
    assert np.array_equal(np.array(list(got1[1])), rows[1] / 0.25) and Lazy.fetched == 1       # (formed once)
    # a subset of the launch's scans, in another order
    res.__dict__.pop('_picks_scan', None)
    sub = [(1, 1), (0, 1)]
    got2 = _device_picks(res, sub, 2, lambda p: True)
    assert got2[0] == [int(idx[2, 3]), int(idx[2, 1])] and np.array_equal(np.array(list(got2[1])), rows[2][[3, 1]] / 0.25)
    # the per-scan view of the same rows
    assert np.array_equal(res._records[(1, 0)]['device_select']['H'][1], rows[1][2])


def test_page_locked_blocks_fall_back_to_ordinary_memory():
    """``device.pinned_empty`` hands out ordinary arrays where the library cannot pin (no GPU here) or where the cap on small
    blocks held by results is reached; ``is_pinned`` follows views down to the block they lie in; ``run_many`` keeps the cut of
    ``run()`` unless told otherwise (the results are then those of the sequential calls bit for bit: tests/test_gpu_run_many.py)"""
    import ctypes
    import inspect
    from maxent_amd import device
    import maxent_amd as mx
    a = device.pinned_empty((3, 5), np.float64, min_bytes=0)
    assert a.shape == (3, 5) and a.dtype == np.float64
    if not device.is_pinned(a):                  # (no GPU: hipHostMalloc fails and the array is numpy's own)
        assert device._small_pinned[0] == 0 or device._small_pinned[0] >= 0
    buf = (ctypes.c_char * 480)()
    blk = np.frombuffer(buf, dtype=np.uint8)
    d = blk[:240].view(np.float64).reshape(3, 2, 5)
    assert device.is_pinned(blk) and device.is_pinned(d) and device.is_pinned(d[1, 0]) and not device.is_pinned(np.empty(4))
    assert device.pinned_empty((0,), np.float64).size == 0
    assert inspect.signature(mx.run_many).parameters['same_cut'].default is True


def test_a_multi_rank_bench_that_hangs_says_where(tmp_path):
    """bench.start_watchdog: a rank that does not finish within the limit names the phase it was in on stderr, rank 0 prints a JSON
    line with value = null and the error, exit code 7 (the first run between two GPUs has never happened: VERDICT r04 item 8)"""
    import json
    import subprocess
    import sys
    code = ("import sys, time, types; sys.path.insert(0, %r); import bench; "
            "bench.PHASE[0] = 'communicator set-up'; "
            "bench.start_watchdog(0, 2, types.SimpleNamespace(steps=5, warmup=1, scaling='strong'), seconds=0.3); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7
    assert 'rank 0 of 2' in r.stderr and 'communicator set-up' in r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['value'] is None and line['watchdog_fired'] is True and line['n_gpus'] == 2 and 'communicator set-up' in line['error']


def test_the_device_keeps_64_singular_directions_only_where_the_others_cannot_matter():
    """batch_solver.directions_to_keep: 1 000 imaginary times give 79 singular values above the reference's 1e-14; the ones beyond the
    64th sit at the rounding floor of the decomposition, and at the minimiser a direction carries |v_k| <= c_k (|ghat_k| + ...) / alpha.
    64 are kept when the sum of these bounds over the dropped directions is below 1e-7 for every element -- not with error bars of
    1e-9 (c_k a hundred thousand times larger), not with a rotated kernel, not when there are at most 64 anyway."""
    from maxent_amd import synthetic, batch_solver as bs
    tau, omega, K, Gmat, _ = synthetic.matrix_G(2, 1000, 200)
    K.reduce_singular_space(1e-14)
    assert len(K.S) > 64
    G = Gmat.reshape(-1, 1000)
    alpha = np.geomspace(1e4, 1e-2, 20)[None, :] * 1000
    D = synthetic.flat_D(omega)
    arrays = dict(G=G, err=synthetic.SIGMA * np.ones((1, 1000)), alpha=alpha, sel=np.zeros(4, dtype=int), D=D[None, :])
    assert bs.directions_to_keep(K, None, arrays) == 64
    specs = [dict(G=g, err=synthetic.SIGMA * np.ones(1000), alpha=alpha[0], U_rot=None, D=D) for g in G]
    assert bs.directions_to_keep(K, specs, None) == 64
    assert bs.directions_to_keep(K, None, dict(arrays, err=1e-9 * np.ones((1, 1000)))) is None
    assert bs.directions_to_keep(K, None, dict(arrays, alpha=alpha * 1e-9)) is None
    assert bs.directions_to_keep(K, [dict(specs[0], U_rot=np.eye(3))], None) is None
    assert bs.directions_to_keep(K, None, dict(arrays, err=synthetic.SIGMA * (1.0 + 0.1 * np.arange(1000) / 1000.0)[None, :])) is None   # (error bars that vary)
    tau2, omega2, K2, Gmat2, _ = synthetic.matrix_G(2, 100, 200)
    K2.reduce_singular_space(1e-14)
    # a mesh of 2 000 frequencies on 1 000 data points: 991 'singular values' above the absolute 1e-14 (the rounding floor of that
    # decomposition): 64 are kept at error bars of 1e-4; at 1e-9 nothing can be dropped and the job is beyond the device (n_s <= 128)
    tau3, omega3, K3, G3 = synthetic.single_G(1000, 2000)
    K3.reduce_singular_space(1e-14)
    assert len(K3.S) > 128
    sp3 = dict(G=G3, err=1e-4 * np.ones(1000), alpha=alpha[0], U_rot=None, D=synthetic.flat_D(omega3))
    assert bs.directions_to_keep(K3, [sp3], None) == 64
    assert bs.directions_to_keep(K3, [dict(sp3, err=1e-9 * np.ones(1000))], None) is None
    assert len(K2.S) <= 64 and bs.directions_to_keep(K2, None, dict(arrays, G=Gmat2.reshape(-1, 100), err=1e-4 * np.ones((1, 100)))) is None
