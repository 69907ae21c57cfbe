"""The alpha loop, batched on the device.

``MaxEntLoop.run`` keeps the reference's signature, order of operations and
return value (reference python/maxent_loop.py:144-302): G-threshold skip,
``K.reduce_singular_space``, the "Minimal chi2" header, the start vector
``v0 = H_of_v.inv(D * delta)``, the ``scale_alpha`` rule, one log line per
alpha, timing, and finally ``result.analyze(analyzers)``.  What changes is the
body of the loop ``for alpha in alpha_mesh: v = minimizer.minimize(Q, v)``
(:241-245): the whole warm-started scan is one *chain* handed to
``libmaxent_hip.so`` (``mxe_solve_chains``), and several scans -- the matrix
elements of ``ElementwiseMaxEnt`` -- go down in a single launch through
:func:`solve_elements`.
"""

from datetime import datetime, timedelta

import numpy as np

from . import device
from .alpha_meshes import LogAlphaMesh
from .analyzers import (LineFitAnalyzer, Chi2CurvatureAnalyzer,
                        EntropyAnalyzer, BryanAnalyzer, ClassicAnalyzer)
from .cost_functions import MaxEntCostFunction, BryanCostFunction
from .functions import PlusMinusEntropy, PlusMinusH_of_v
from .logtaker import Logtaker, VerbosityFlags
from .maxent_result import MaxEntResult
from .minimizers import LevenbergMinimizer
from .probabilities import NormalLogProbability


# --------------------------------------------------------------------------
#  device batch
# --------------------------------------------------------------------------

class _LazyProduct(object):
    """``A @ K_delta.T`` (MaxEntResult.G_rec, reference maxent_result.py:908) evaluated on first use:
    a 10 MFLOP product per matrix element that most jobs never look at."""

    def __init__(self, A, K_delta):
        self._A, self._K, self._val = A, K_delta, None

    def __array__(self, dtype=None, copy=None):
        if self._val is None:
            self._val = np.dot(np.asarray(self._A), self._K.T)
            self._A = self._K = None
        return self._val if dtype is None else self._val.astype(dtype, copy=False)


def solve_elements(K, specs, minimizer, device_id=0, waves_per_chain=0,
                   want_logdet=False, chi2_factor=1.0, device_ids=None, want_H='lazy', select=(0, 0.2), while_waiting=None,
                   defer=False, in_flight=0, arrays=None):
    """Solve the alpha scans of several elements in ONE kernel launch per device.

    ``K``: kernel whose singular space has been reduced (U, S, V are staged once per device and kept,
    :class:`maxent_amd.batch_solver.BatchSolver`).
    ``specs``: list of dicts with keys
        G (data vector in the element's data space), err (same length),
        U_rot (left factor if that space is rotated, else None),
        D (default model incl. delta), kind (device.ENTROPY_*),
        v0 (start vector), alpha (scaled alphas, in visiting order).
    All specs must have the same number of alphas.
    ``device_ids``: the GPUs of this process to shard the elements over (element e of the batch on
    device e mod N, one gather at the end); default: the single ``device_id``.
    ``want_logdet``: also return log det(I + M W/alpha) per alpha (device kernel; the expensive term of
    NormalLogProbability).  ``want_H``: 'lazy' (H stays on the device until it is looked at), True, False.
    Returns (list of per-spec dicts(alpha, v, H, chi2, S, Q, n_iter, converged, n_evals[, logdet]), info).
    ``defer``: stage and launch only; returns the function that waits for the device and returns the above (the caller launches
    other jobs meanwhile: ``ElementwiseMaxEnt.run_async``).  ``in_flight``: how many such jobs the caller keeps on the GPU at a time
    (``mxe_opts.in_flight``: each is cut into 1 / n as many cold-started pieces -- n jobs fill the GPU together).
    """
    if not specs and arrays is None:
        done = ([], dict(kernel_ms=0.0))
        return (lambda: done) if defer else done
    from .batch_solver import BatchSolver, directions_to_keep
    solver = BatchSolver.for_kernel(K, (device_id,) if device_ids is None else device_ids,
                                    keep=directions_to_keep(K, specs, arrays))
    extra = dict(in_flight=int(in_flight)) if in_flight and in_flight > 1 else {}
    opts = minimizer.to_opts(waves_per_chain=waves_per_chain, chi2_factor=float(chi2_factor), **extra)
    if arrays is not None:
        # (the job as arrays -- BatchSolver.solve_begin --: results as a LazySols, nothing per scan built)
        begun = solver.solve_begin(K, None, opts, select=select, early_select=True, arrays=arrays)
        return begun if defer else begun()
    if defer:
        return solver.solve_begin(K, specs, opts, want_logdet=want_logdet, want_H=want_H, select=select,
                                  while_waiting=while_waiting, early_select=True)
    return solver.solve(K, specs, opts, want_logdet=want_logdet, want_H=want_H, select=select, while_waiting=while_waiting)


def select_params(analyzers):
    """(linefit_deg, gamma, default) for ``mxe_select3_launch`` from a list of analyzers -- the device then picks the alphas
    of the LineFit / Chi2Curvature / Entropy analyzers in the list behind the solve; ``default``: which of the three (0, 1,
    2) comes first in the list: its rows come with the solve, the others' when looked at --, None when none of them is there"""
    from .analyzers import LineFitAnalyzer, Chi2CurvatureAnalyzer, EntropyAnalyzer
    deg, gamma, first = 0, 0.2, None
    for a in analyzers or ():
        if isinstance(a, LineFitAnalyzer):
            deg, which = int(a.linefit_deg), 0
        elif isinstance(a, Chi2CurvatureAnalyzer):
            gamma, which = float(a.gamma), 1
        elif isinstance(a, EntropyAnalyzer):
            which = 2
        else:
            continue
        if first is None:
            first = which
    return (deg, gamma, first) if first is not None and deg in (0, 1) and gamma > 0 else None


def solve_single(cost_function, v0, minimizer, device_id=0):
    """``Minimizer.minimize(function, v0)`` for one alpha on the device."""
    cf = cost_function
    if cf._alpha is None:
        raise Exception('call set_alpha on the cost function first')
    K = cf.K
    K.S        # trigger the SVD if needed
    spec = dict(G=cf.G, err=cf.err, U_rot=(K.U if K.rotation is not None else None),
                D=cf.D.D, kind=cf.entropy_kind, v0=np.asarray(v0, dtype=float),
                alpha=np.array([cf._alpha], dtype=float))
    res, _ = solve_elements(K, [spec], minimizer, device_id=device_id, chi2_factor=cf.chi2_factor,
                            want_H=False, select=None)
    r = res[0]
    return r['v'][0], dict(n_iter=r['n_iter'][0], converged=r['converged'][0])


# --------------------------------------------------------------------------
#  MaxEntLoop
# --------------------------------------------------------------------------

class MaxEntLoop(object):
    """alpha loop of one data vector (reference maxent_loop.py:33-140)."""

    def __init__(self, cost_function=None, minimizer=None, alpha_mesh=None,
                 probability=None, analyzers=None, logtaker=None,
                 G_threshold=1.e-10, reduce_singular_space=1.e-14,
                 A_init=None, interactive=True, scale_alpha='Ndata',
                 device_id=0, device_ids=None):
        if cost_function is None:
            cost_function = MaxEntCostFunction()
        elif isinstance(cost_function, str):
            name = cost_function.lower()
            if name == 'normal':
                cost_function = MaxEntCostFunction()
            elif name == 'plusminus':
                cost_function = MaxEntCostFunction(S=PlusMinusEntropy(),
                                                   H_of_v=PlusMinusH_of_v())
            elif name == 'bryan':
                cost_function = BryanCostFunction()
            else:
                raise Exception('Unknown cost_function str {}.'.format(
                    cost_function))
        self.cost_function = cost_function
        self.minimizer = minimizer if minimizer is not None \
            else LevenbergMinimizer()
        self.alpha_mesh = alpha_mesh if alpha_mesh is not None \
            else LogAlphaMesh()
        self.logtaker = logtaker if logtaker is not None else Logtaker()
        if isinstance(probability, str):
            if probability.lower() == 'normal':
                probability = NormalLogProbability()
            else:
                raise Exception('Unknown probability str {}.'.format(probability))
        self.probability = probability
        if analyzers is None:
            analyzers = [LineFitAnalyzer(), Chi2CurvatureAnalyzer(),
                         EntropyAnalyzer()]
            if self.probability is not None:
                analyzers += [BryanAnalyzer(), ClassicAnalyzer()]
        self.analyzers = analyzers
        self.G_threshold = G_threshold
        self.interactive = interactive
        self.A_init = A_init
        self.reduce_singular_space = reduce_singular_space
        self.scale_alpha = scale_alpha
        self.device_id = device_id
        self.device_ids = device_ids
        self.last_launch = None

    # ---- pieces of run(), also used by the element-wise driver --------
    def _alpha_scale(self):
        """reference maxent_loop.py:216-232."""
        if self.scale_alpha is None:
            return 1.0
        if isinstance(self.scale_alpha, str):
            if self.scale_alpha.lower() == 'ndata':
                return float(len(self.G))
            raise Exception('Unknown value {} for scale_alpha'.format(
                self.scale_alpha))
        return float(self.scale_alpha)

    def below_threshold(self):
        return np.max(np.abs(self.G)) < self.G_threshold

    def make_spec(self, G=None, err=None):
        """everything the device needs for this loop's current G / err / D / K.  ``G`` and ``err`` given:
        the spec of ANOTHER data vector of the same (unrotated) problem, the loop itself untouched -- the
        element-wise drivers build the specs of all matrix elements this way"""
        other = G is not None
        if other:
            assert self.K.rotation is None, 'specs of other data vectors need an unrotated kernel'
            G_use = np.array(G, dtype=float)
            err_use = np.asarray(err, dtype=float) * np.ones(len(G_use))
        else:
            assert self.err is not None, 'No error specified'
            G_use = np.array(self.G, dtype=float)
            err_use = np.array(self.err, dtype=float) * np.ones(len(G_use))
        self.K.reduce_singular_space(self.reduce_singular_space)
        start = (self.D.D if self.A_init is None else
                 np.asarray(self.A_init)) * self.omega.delta
        start = np.array(start, dtype=float)
        # the start vector depends on (V, start image, entropy kind, delta) only: the element-wise drivers ask
        # for it once per matrix element with the same inputs
        V = self.K.V
        held = self.__dict__.get('_v0_held')
        Dv = np.asarray(self.D.D, dtype=float)
        if held is None or held[0] is not V or held[1] != self.H_of_v.kind or not np.array_equal(held[2], self.omega.delta) \
                or not np.array_equal(held[3], start) or not np.array_equal(held[4], Dv):     # (contents: an in-place edit of D.D or A_init counts)
            held = self._v0_held = (V, self.H_of_v.kind, np.array(self.omega.delta), start.copy(), Dv.copy(), self.H_of_v.inv(start))
        v0 = held[5].copy()
        scale = self._alpha_scale()
        K = self.K
        return dict(G=G_use,
                    err=err_use,
                    U_rot=(K.U if K.rotation is not None else None),
                    D=np.array(self.D.D, dtype=float),
                    kind=self.cost_function.entropy_kind,
                    v0=v0,
                    alpha=np.asarray(self.alpha_mesh, dtype=float) * scale,
                    scale_alpha=scale,
                    G_orig=(G_use if other else np.array(self.cost_function.G_orig, dtype=float)),
                    data_variable=np.array(self.data_variable, dtype=float),
                    A_matrix=self.A_of_H.matrix(),
                    T=K.rotation)

    @staticmethod
    def spec_like(template, G, err):
        """the spec of another data vector of the problem ``template`` was made for (``make_spec(G=..., err=...)``):
        everything but G and err is shared with it"""
        spec = dict(template)
        G_use = np.array(G, dtype=float)
        spec['G'] = spec['G_orig'] = G_use
        # (an error array of the right length is taken as it is: the element-wise drivers hand the same one to every element)
        spec['err'] = err if (isinstance(err, np.ndarray) and err.dtype == float and err.shape == G_use.shape) \
            else np.asarray(err, dtype=float) * np.ones(len(G_use))
        return spec

    def make_record(self, spec, sol):
        """MaxEntResult arrays of one finished scan (maxent_result.py:835-967)."""
        A = sol.get('A')
        if A is None:
            from .batch_solver import LazyA, LazyH
            A = LazyA(sol['H'], self.A_of_H) if isinstance(sol['H'], LazyH) else self.A_of_H.f(sol['H'])
        rec = dict(sol)
        rec['A'] = A
        rec['G'] = spec['G']
        rec['G_orig'] = spec['G_orig']
        rec['data_variable'] = spec['data_variable']
        rec['G_rec'] = _LazyProduct(A, self.K.K_delta)      # K_delta A, formed when it is looked at
        rec['omega'] = self.omega
        X = len(sol['alpha'])
        if self.probability is not None and sol.get('logdet') is not None:
            # the determinant came from the device (mxe_logdet)
            rec['probability'] = self.probability.from_logdet(
                sol['logdet'], sol['alpha'], sol['Q'], len(self.omega))
        elif self.probability is not None:
            K = self.K
            u = np.dot(sol['v'], K.V.T)
            Dd = spec['D'][np.newaxis, :]
            w = Dd * np.exp(u) if spec['kind'] == device.ENTROPY_NORMAL \
                else Dd * (np.exp(u) + np.exp(-u))
            rec['probability'] = self.probability.evaluate(
                K.U, K.S, K.V, spec['err'], sol['alpha'], w, sol['Q'])
        else:
            rec['probability'] = np.full(X, np.nan)
        return rec

    def make_records(self, specs, sols):
        """:meth:`make_record` for the scans of one launch: what they share is set up once"""
        if self.probability is not None or any(sol.get('A') is not None for sol in sols):
            return [self.make_record(spec, sol) for spec, sol in zip(specs, sols)]
        from .batch_solver import LazyA, LazyH
        A_of_H, K_delta, omega = self.A_of_H, self.K.K_delta, self.omega
        nans = {}
        out = []
        for spec, sol in zip(specs, sols):
            H = sol['H']
            A = LazyA(H, A_of_H) if isinstance(H, LazyH) else A_of_H.f(H)
            X = len(sol['alpha'])
            nan = nans.get(X)
            if nan is None:
                nan = nans[X] = np.full(X, np.nan)
                nan.setflags(write=False)           # (one array for every scan of the launch)
            rec = dict(sol)
            rec.update(A=A, G=spec['G'], G_orig=spec['G_orig'], data_variable=spec['data_variable'],
                       G_rec=_LazyProduct(A, K_delta), omega=omega, probability=nan)
            out.append(rec)
        return out

    def log_alpha_lines(self, sol):
        """reference maxent_loop.py:248-257, 286-289."""
        n = len(sol['alpha'])
        width = int(np.ceil(np.log10(max(n, 2))))
        for i in (range(n) if self.logtaker.wants(VerbosityFlags.AlphaLoop) else ()):
            self.logtaker.message(
                VerbosityFlags.AlphaLoop,
                'alpha[{:' + str(width) + 'd}] = {:16.8e}, chi2 = {:16.8e}, n_iter={:8d}{}',
                i, sol['alpha'][i], sol['chi2'][i], int(sol['n_iter'][i]),
                ' ' if sol['converged'][i] else '!')
        callback = getattr(self.minimizer, 'verbose_callback', None)
        if callback is not None and hasattr(self.minimizer, 'to_opts'):
            # SolverDetails (levenberg_minimizer.py:165-170): the last iterate of every alpha
            for i in range(n):
                callback('{:6d} Q: {:12.6e}, evaluations: {:d}, conv: {:d}'.format(
                    int(sol['n_iter'][i]), sol['Q'][i], int(sol['n_evals'][i]), int(sol['converged'][i])))
        if not np.all(sol['converged']):
            self.logtaker.message(
                VerbosityFlags.AlphaLoop,
                '\n! ... The minimizer did not converge. Results might be wrong.\n')
        self.note_minimizer_state(sol, int(np.sum(sol['n_iter'])))

    def note_minimizer_state(self, last_sol, n_iter_total):
        """``n_iter_last`` / ``n_iter`` / ``converged`` of the minimiser as after the reference's loop
        (levenberg_minimizer.py:143,245-246): those of the last alpha of the last scan"""
        if hasattr(self.minimizer, 'to_opts'):
            self.minimizer.n_iter_last = int(last_sol['n_iter'][-1])
            self.minimizer.n_iter += int(n_iter_total)
            self.minimizer.converged = bool(last_sol['converged'][-1])

    def scan_with_user_minimizer(self, spec):
        """The reference's loop body (maxent_loop.py:241-266) for a minimiser that is not the device
        solver: any object with ``minimize(function, v0) -> v`` (minimizers/minimizer.py:23-28).  It works
        on the device-evaluated cost function (``f`` / ``d`` / ``dd`` are ``mxe_eval_batch`` calls), one
        alpha after the other, warm started; ``n_iter_last`` and ``converged`` are read off the minimiser
        where it has them."""
        cf = self.cost_function
        v = np.array(spec['v0'], dtype=float)
        keys = ('v', 'H', 'chi2', 'S', 'Q', 'n_iter', 'converged')
        rows = dict((k, []) for k in keys)
        for a in spec['alpha']:
            cf.set_alpha(float(a))
            v = np.array(self.minimizer.minimize(cf, v), dtype=float)
            at = cf(v)
            rows['v'].append(v.copy())
            rows['H'].append(np.array(at.H_of_v.f()))
            rows['chi2'].append(at.chi2.f())
            rows['S'].append(at.S.f())
            rows['Q'].append(at.f())
            rows['n_iter'].append(int(getattr(self.minimizer, 'n_iter_last', 0)))
            rows['converged'].append(bool(getattr(self.minimizer, 'converged', True)))
        sol = dict((k, np.array(rows[k])) for k in keys)
        sol['alpha'] = np.asarray(spec['alpha'], dtype=float)
        sol['n_evals'] = sol['n_iter']
        sol['A'] = None
        return sol

    # ---- main entry ------------------------------------------------------
    def run(self, result=None, matrix_element=None, complex_index=None):
        """Run the alpha scan; returns the :class:`MaxEntResult` (or None if
        max|G| < G_threshold, in which case the element is recorded in
        ``result.zero_elements``)."""
        if self.below_threshold():
            if result is not None and matrix_element is not None:
                result._zero_elements.append(matrix_element)
            self.logtaker.error_message(
                'G below threshold, not performing the calculation.')
            return None
        self.logtaker.welcome_message()
        spec = self.make_spec()
        if self.logtaker.verbose & VerbosityFlags.Header:
            A_min = np.linalg.lstsq(self.K.K, self.G, rcond=-1)[0]
            self.logtaker.message(VerbosityFlags.Header, 'Minimal chi2: {}',
                                  self.chi2.f(A_min))
            self.logtaker.message(
                VerbosityFlags.Header, 'scaling alpha by a factor {}{}',
                spec['scale_alpha'],
                ' (number of data points)' if isinstance(self.scale_alpha, str) else '')
        if result is None:
            result = MaxEntResult()
        if result._default_analyzer_name is None and self.analyzers:
            result._default_analyzer_name = self.analyzers[0].name
        self.check_consistency()
        result.start_timing(matrix_element, complex_index)
        t0 = datetime.now()
        if hasattr(self.minimizer, 'to_opts'):
            sols, info = solve_elements(self.K, [spec], self.minimizer,
                                        want_logdet=self.probability is not None,
                                        device_id=self.device_id, device_ids=self.device_ids,
                                        chi2_factor=self.cost_function.chi2_factor,
                                        select=select_params(self.analyzers))
            sol = sols[0]
        else:
            sol, info = self.scan_with_user_minimizer(spec), dict(kernel_ms=0.0)
        self.last_launch = info
        self.log_alpha_lines(sol)
        rec = self.make_record(spec, sol)
        dt = (datetime.now() - t0) / max(len(sol['alpha']), 1)
        rec['run_times'] = [dt] * len(sol['alpha'])
        result.add_element_results(rec, matrix_element, complex_index)
        run_time = result.end_timing(matrix_element, complex_index)
        self.logtaker.message(VerbosityFlags.Timing,
                              'MaxEnt loop finished in {}', run_time)
        if sol.get('device_select') is not None:
            # (the device chose the alphas of the LineFit / Chi2Curvature / Entropy analyzers behind the solve: their results are
            #  built from that choice when somebody looks, as for the scans of an element-wise job -- 0.3 ms of the 1 ms of a single scan)
            result.analyze_batch(self.analyzers, [result._key(matrix_element, complex_index)], picks_for_one=True)
        else:
            result.analyze(self.analyzers, matrix_element, complex_index)
        return result

    # ---- helpers ----------------------------------------------------------
    def check_consistency(self):
        """all children must refer to the same grids (reference
        maxent_loop.py:306-337)."""
        cf = self.cost_function
        assert cf.chi2.K is cf.H_of_v.K
        assert np.all(np.asarray(cf.K.omega) == np.asarray(self.omega))
        assert np.all(np.asarray(cf.D.omega) == np.asarray(self.omega))
        assert np.all(np.asarray(cf.S.omega) == np.asarray(self.omega))
        assert np.all(np.asarray(cf.H_of_v.omega) == np.asarray(self.omega))
        assert np.all(np.asarray(cf.A_of_H.omega) == np.asarray(self.omega))
        assert np.all(cf.H_of_v.D.D == cf.S.D.D)
        assert len(self.G) == np.asarray(cf.K.K).shape[0], \
            'G and K do not have the same number of data points'
        assert len(self.D.D) == np.asarray(cf.K.K).shape[1]

    def set_verbosity(self, verbosity=None, add=None, remove=None,
                      change_callback=True):
        if verbosity is not None:
            self.logtaker.verbose = verbosity
        if add is not None:
            self.logtaker.verbose |= add
        if remove is not None:
            self.logtaker.verbose &= ~remove
        if change_callback:          # maxent_loop.py:375-382
            wanted = self.logtaker.verbose & VerbosityFlags.SolverDetails
            self.minimizer.verbose_callback = self.logtaker.solver_verbose_callback if wanted else None


# the parameters of the problem live in the cost function; the loop offers them under the same names
# (get_X / set_X / property X; reference maxent_loop.py:384-502)
def _forwarded(name):
    def getter(self):
        return getattr(self.cost_function, 'get_' + name)()

    def setter(self, value, **update_flags):
        getattr(self.cost_function, 'set_' + name)(value, **update_flags)
    return getter, setter


for _name in ('K', 'G', 'err', 'omega', 'data_variable', 'D', 'chi2', 'S', 'H_of_v', 'A_of_H'):
    _get, _set = _forwarded(_name)
    setattr(MaxEntLoop, 'get_' + _name, _get)
    setattr(MaxEntLoop, 'set_' + _name, _set)
    setattr(MaxEntLoop, _name, property(_get, _set))
del _name, _get, _set
