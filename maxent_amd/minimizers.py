"""Minimiser front-end objects (reference python/minimizers/).

On the device the whole per-alpha minimisation runs inside the chain kernel;
these classes carry its parameters with the reference's names
(``LevenbergMinimizer``: levenberg_minimizer.py:92-121; convergence methods:
convergence_methods.py:24-122) and translate them to ``mxe_opts``.

Differences to the reference, by design (DESIGN.md "Minimiser"):
* the damped Newton step is Bryan's (damping in the entropy metric, started at
  mu = 0 and raised only until the step bound holds), not the multiplicative
  mu scan, so ``mu0``/``nu``/``max_mu`` map to ``mu_first``/``mu_grow``/
  ``mu_max`` (in units of alpha);
* the default stopping rule is :class:`NewtonStepConvergenceMethod` (1e-9):
  the reference's defaults (``max|dQ| < 1e-4`` OR relative change < 1e-16) stop
  up to ~3e-5 (relative L2 of H) short of the fixed point, which is above the
  1e-6 parity target.  They remain available and mean what they mean in the
  reference.
"""

import numpy as np

from . import device


class ConvergenceMethod(object):
    """Combinable with ``&`` and ``|`` like the reference's
    (convergence_methods.py:36-78; note that the reference's AND is an OR
    too -- ``is_conv1 or is_conv2`` -- and so is ours)."""

    def __and__(self, other):
        return AndConvergenceMethod(self, other)

    def __or__(self, other):
        return OrConvergenceMethod(self, other)

    def apply(self, opts):
        raise NotImplementedError

    def __call__(self, function, v, **kwargs):
        """(value, converged) for the host iteration of :meth:`LevenbergMinimizer.minimize` on a general function
        (reference convergence_methods.py:24-34); ``Q0``, ``Q1``: the function value before and after the last step"""
        raise NotImplementedError


class _Pair(ConvergenceMethod):
    def __init__(self, one, two):
        self.one = one
        self.two = two

    def apply(self, opts):
        self.one.apply(opts)
        self.two.apply(opts)

    def __call__(self, function, v, **kwargs):
        a, ok_a = self.one(function, v, **kwargs)
        b, ok_b = self.two(function, v, **kwargs)
        value = np.nan if (np.isnan(a) or np.isnan(b)) else min(a, b)
        return value, (ok_a or ok_b)


class AndConvergenceMethod(_Pair):
    pass


class OrConvergenceMethod(_Pair):
    pass


class MaxDerivativeConvergenceMethod(ConvergenceMethod):
    """max |dQ/dv| < criterion with dQ/dv = W g (the reference's MaxEntCostFunction.d,
    maxent_cost_function.py:85-118).  On the device the maximum runs over the coupled block of
    singular directions (``mxe_opts.tol_d``); the decoupled ones are solved by their diagonal
    Newton step and contribute at the level of ``decouple_tol`` only."""

    def __init__(self, convergence_criterion):
        self.convergence_criterion = convergence_criterion

    def apply(self, opts):
        opts.tol_d = float(self.convergence_criterion)

    def __call__(self, function, v, **kwargs):
        value = float(np.max(np.abs(function.d(v))))
        return value, value < self.convergence_criterion


class FunctionChangeConvergenceMethod(ConvergenceMethod):
    """|Q0 - Q1| < criterion (reference convergence_methods.py:99-109).  The device solver tests the RELATIVE
    change (scale free); the absolute form is accepted for the value of Q it is given with."""

    def __init__(self, convergence_criterion):
        self.convergence_criterion = convergence_criterion

    def apply(self, opts):
        raise NotImplementedError('the device solver stops on the relative change of Q: use '
                                  'RelativeFunctionChangeConvergenceMethod (or NewtonStepConvergenceMethod)')

    def __call__(self, function, v, **kwargs):
        value = abs(kwargs['Q0'] - kwargs['Q1'])
        return value, value < self.convergence_criterion


class RelativeFunctionChangeConvergenceMethod(ConvergenceMethod):
    """|Q0 - Q1| / |Q1| < criterion between two accepted iterates."""

    def __init__(self, convergence_criterion):
        self.convergence_criterion = convergence_criterion

    def apply(self, opts):
        opts.tol_relq = float(self.convergence_criterion)

    def __call__(self, function, v, **kwargs):
        with np.errstate(all='ignore'):
            value = abs(abs(kwargs['Q0'] - kwargs['Q1']) / kwargs['Q1'])
        return value, value < self.convergence_criterion


class NewtonStepConvergenceMethod(ConvergenceMethod):
    """||dH||_2 / ||H||_2 < criterion for the Newton correction dH = w o V delta
    (scale free; not in the reference).

    With ``estimate`` (default) the criterion is also applied to the estimated
    NEXT correction after a full Newton step,
    (expm1(max|du|) + decouple_tol) * ||dH||/||H||,
    which is the error left in the accepted point: the iteration that would
    only confirm convergence is not run.  ``estimate=False`` tests the
    correction actually taken."""

    def __init__(self, convergence_criterion=1.e-9, estimate=True):
        self.convergence_criterion = convergence_criterion
        self.estimate = estimate

    def apply(self, opts):
        opts.tol_h = float(self.convergence_criterion)
        opts.stop_estimate = 1 if self.estimate else 0


class NullConvergenceMethod(ConvergenceMethod):
    """everything counts as converged after ``miniter`` iterations."""

    def apply(self, opts):
        opts.tol_h = 1e300

    def __call__(self, function, v, **kwargs):
        return 0, True


class Minimizer(object):
    def minimize(self, function, v0):
        raise NotImplementedError('Use a subclass of Minimizer')


class LevenbergMinimizer(Minimizer):
    """Parameters of the per-alpha damped Newton iteration.

    ``precision='f32'`` selects the binary32 streaming variant of the chain
    kernel (``mxe_opts.precision``; BASELINE config 5's fp32-vs-fp64 sweep):
    V, u, H, exp, both mat-vecs and the Gram matrix in binary32, the Newton
    system and all scalars in binary64.

    ``J_squared`` and ``marquardt`` (levenberg_minimizer.py:177-185) choose the
    damping matrix of the reference's search (J^T J instead of J; diag J instead
    of 1).  They change the iterates, not the point where dQ/dv = 0.  For a
    general ``DoublyDerivableFunction`` -- anything but the MaxEnt cost function
    -- ``minimize`` runs that search on the host and both flags do what they do
    in the reference; for the MaxEnt cost function the device iteration damps in
    the entropy metric and raises mu only when Bryan's bound or the descent test
    asks for it: there both flags are accepted and recorded and the result is
    the same minimum.

    ``verbose_callback`` (levenberg_minimizer.py:165-170) is called once per
    alpha, after the launch, with the record of the last iterate -- the
    iterations themselves happen inside one kernel and have no host in them.

    ``n_iter_last`` / ``n_iter`` / ``converged`` are filled after a run like
    in the reference (levenberg_minimizer.py:143,245-246); for a batched run
    they refer to the last alpha of the last chain, per-problem values are in
    the result arrays.
    """

    def __init__(self, convergence=None, maxiter=1000, miniter=0,
                 J_squared=False, marquardt=False, mu0=1.e-3, nu=4.0,
                 max_mu=1.e20, step_max=0.2, verbose_callback=None,
                 precision='f64'):
        if precision not in ('f64', 'f32'):
            raise ValueError("precision must be 'f64' or 'f32'")
        self.precision = precision
        self.convergence = convergence if convergence is not None \
            else NewtonStepConvergenceMethod(1.e-9)
        self.maxiter = maxiter
        self.miniter = miniter
        self.J_squared = J_squared
        self.marquardt = marquardt
        self.mu0 = mu0
        self.nu = nu
        self.max_mu = max_mu
        self.step_max = step_max
        self.verbose_callback = verbose_callback
        self.n_iter = 0
        self.n_iter_last = 0
        self.converged = False

    def to_opts(self, **extra):
        if self.nu <= 1.0:
            raise Exception('If nu <= 1, there will be an infinite loop.')
        o = device.default_opts(maxiter=int(self.maxiter),
                                miniter=int(self.miniter),
                                tol_h=0.0, tol_d=0.0, tol_relq=0.0,
                                step_max=float(self.step_max),
                                mu_first=float(self.mu0),
                                mu_grow=float(self.nu),
                                mu_max=float(self.max_mu),
                                precision=(device.PRECISION_F32
                                           if self.precision == 'f32'
                                           else device.PRECISION_F64),
                                **extra)
        self.convergence.apply(o)
        return o

    def minimize(self, function, v0):
        """``function`` a MaxEnt cost function with ``set_alpha`` called: one alpha on the device (reference
        minimizer.py:23-28).  Any other ``DoublyDerivableFunction`` -- ``function(v)`` pins an argument, ``f / d / dd``
        (reference functions.py:96-146) --: the reference's damped Newton search on the host,
        :meth:`_minimize_general`."""
        if not hasattr(function, 'entropy_kind') or getattr(function, '_alpha', None) is None:
            return self._minimize_general(function, v0)
        from .maxent_loop import solve_single
        v, info = solve_single(function, v0, self)
        self.n_iter_last = int(info['n_iter'])
        self.n_iter += self.n_iter_last
        self.converged = bool(info['converged'])
        return v

    def _general_convergence(self):
        """the stopping rule of the host search: the caller's, or -- the device's default rule has no meaning for a
        general function -- the reference's default (levenberg_minimizer.py:103-106)"""
        if isinstance(self.convergence, NewtonStepConvergenceMethod):
            return OrConvergenceMethod(MaxDerivativeConvergenceMethod(1.e-4),
                                       RelativeFunctionChangeConvergenceMethod(1.e-16))
        return self.convergence

    def _minimize_general(self, function, v0):
        """Levenberg-Marquardt search for a root of ``function.d`` (reference levenberg_minimizer.py:123-248), for
        functions that are not the MaxEnt cost function -- the alpha scan never comes here, it runs in the chain kernel.

        Every iteration solves ``(J + mu D) dv = g`` for the step, with ``g = d(v)``, ``J = dd(v)`` (``J_squared``: the
        normal equations ``J^T J``, ``J^T g``; levenberg_minimizer.py:177-180) and ``D`` the identity or, with
        ``marquardt``, the diagonal of ``J`` (:182-185), and searches the damping ``mu``:

        1. ``mu`` grows by factors ``nu`` until the step does not increase the function (a NaN counts as an increase);
        2. the neighbour ``nu mu`` is tried: the search then walks in the direction that lowers the function --
           up while the neighbour is better, else DOWN (``mu / nu``, ``mu / nu^2`` ...: after a rejected step ``mu``
           falls back as soon as smaller values do better) --, as long as it keeps falling and ``mu`` stays inside
           ``(nu eps, max_mu)``; the last step before it rose again is taken.

        ``mu`` carries over to the next iteration."""
        if self.nu <= 1.0:
            raise Exception('If nu <= 1, there will be an infinite loop.')
        convergence = self._general_convergence()
        nu, tiny = float(self.nu), float(self.nu) * np.finfo(float).eps
        v = np.array(v0, dtype=float)
        mu = float(self.mu0)
        self.converged = False
        here = function(v)
        Q_now, Q_before = here.f(), np.nan
        done = 0
        for it in range(int(self.maxiter)):
            done = it + 1
            g, J = np.asarray(here.d(), dtype=float), np.atleast_2d(np.asarray(here.dd(), dtype=float))
            status, self.converged = convergence(here, v, Q0=Q_before, Q1=Q_now)
            if self.verbose_callback is not None:
                self.verbose_callback('{:6d} Q: {:12.6e}, max_f: {:12.6e}, conv: {:12.6e}'.format(
                    it + 1, Q_now, np.max(np.abs(g)), status))
            if self.converged and it >= self.miniter:
                break
            if self.J_squared:
                g, J = np.dot(J.T, g), np.dot(J.T, J)
            D = np.diag(np.diag(J)) if self.marquardt else np.eye(len(J))

            def trial(m):
                with np.errstate(all='ignore'):
                    try:
                        step = np.linalg.solve(J + m * D, g)
                        return step, function(v - step).f()
                    except np.linalg.LinAlgError:
                        return np.zeros_like(g), np.nan

            Q_before = Q_now
            step, Q = trial(mu)
            while (Q > Q_before or np.isnan(Q)) and mu < self.max_mu:        # 1. a step that does not go uphill
                mu *= nu
                step, Q = trial(mu)
            step_up, Q_up = trial(nu * mu)                                    # 2. which way does mu want to go?
            # The walk of levenberg_minimizer.py:209-233 with its pairings: going up, the first candidate (the step of
            # nu mu) is measured against the value at mu itself (:216-218); going down, mu is first raised by nu and the
            # step of mu is seeded with the value the neighbour nu mu gave (:222-224), so that the loop's first pass
            # re-evaluates mu and its second the real candidate mu / nu.  The step taken is the last one before the
            # function rose again; mu ends on the value that made it rise and carries over to the next iteration.
            if Q_up < Q:
                factor, next_step, next_Q = nu, step_up, Q
            else:
                factor, next_step, next_Q = 1.0 / nu, step, Q_up
            mu *= nu
            best_step, best_Q = step, np.inf
            while next_Q < best_Q and tiny < mu < self.max_mu:
                best_step, best_Q = next_step, next_Q
                mu *= factor
                next_step, next_Q = trial(mu)
            v = v - best_step
            here = function(v)
            Q_now = here.f()
        self.n_iter_last = done
        self.n_iter += done
        return v
