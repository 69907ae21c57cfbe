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


class _Pair(ConvergenceMethod):
    def __init__(self, one, two):
        self.one = one
        self.two = two

    def apply(self, opts):
        self.one.apply(opts)
        self.two.apply(opts)


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


class FunctionChangeConvergenceMethod(ConvergenceMethod):
    """|Q0 - Q1| < criterion (reference convergence_methods.py:99-109).  The device solver tests the RELATIVE
    change (scale free); the absolute form is accepted for the value of Q it is given with."""

    def __init__(self, convergence_criterion):
        self.convergence_criterion = convergence_criterion

    def apply(self, opts):
        raise NotImplementedError('the device solver stops on the relative change of Q: use '
                                  'RelativeFunctionChangeConvergenceMethod (or NewtonStepConvergenceMethod)')


class RelativeFunctionChangeConvergenceMethod(ConvergenceMethod):
    """|Q0 - Q1| / |Q1| < criterion between two accepted iterates."""

    def __init__(self, convergence_criterion):
        self.convergence_criterion = convergence_criterion

    def apply(self, opts):
        opts.tol_relq = float(self.convergence_criterion)


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
    of 1).  They change the iterates, not the point where dQ/dv = 0; the device
    iteration damps in the entropy metric and raises mu only when Bryan's bound
    or the descent test asks for it, so both flags are accepted and recorded
    and the result is the same minimum.

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
        """One alpha on the device: ``function`` is a bound cost function with
        ``set_alpha`` called (reference minimizer.py:23-28)."""
        if not hasattr(function, 'entropy_kind') or getattr(function, '_alpha', None) is None:
            raise TypeError('LevenbergMinimizer runs on the device and minimises MaxEnt cost functions '
                            '(MaxEntCostFunction / BryanCostFunction with set_alpha called); a general '
                            'DoublyDerivableFunction has no device form')
        from .maxent_loop import solve_single
        v, info = solve_single(function, v0, self)
        self.n_iter_last = int(info['n_iter'])
        self.n_iter += self.n_iter_last
        self.converged = bool(info['converged'])
        return v
