"""Building blocks of the cost function, evaluated on the GPU.

The reference composes ``Q_alpha(v) = chi2(H(v)) eta / 2 - alpha S(H(v))`` from
numpy objects with the methods ``f`` / ``d`` / ``dd`` (reference
python/functions.py).  The objects here keep those names, constructor arguments
and the calling protocol

    ``fn.f(x)``, ``fn.d(x)``, ``fn.dd(x)``      value and derivatives at ``x``
    ``b = fn(x)``; ``b.f()``, ``b.d()`` ...     ``x`` pinned, every quantity evaluated once

but hold no arithmetic of the hot path: values come from ``mxe_eval_batch`` /
``mxe_entropy`` (csrc/mxe_eval.hip.h) through :class:`maxent_amd.evaluator.Evaluator`.
What is formed on the host are products of device results with constants of
the problem -- ``diag(w) V``, ``V M V^T`` -- and the one-off maps around a solve
(start vector, A = H / delta).

The entropy kind selects the kernel variant: ``NormalEntropy`` + ``NormalH_of_v``
-> MXE_ENTROPY_NORMAL, ``PlusMinusEntropy`` + ``PlusMinusH_of_v`` ->
MXE_ENTROPY_PLUSMINUS (reference functions.py:491-564, 720-796).
"""

import copy

import numpy as np

from . import device
from .device import ENTROPY_NORMAL, ENTROPY_PLUSMINUS
from .evaluator import Evaluator
from .hostprep import safelog
from .preblur import get_preblur


def parameters(*names):
    """Class decorator: for every name a stored attribute ``_<name>``, the accessor pair
    ``get_<name>()`` / ``set_<name>(value, **update_flags)`` and the property ``<name>``.  The
    ``update_*`` flags of the reference's setters say whether dependent tables are to be
    recomputed right away; here nothing is tabulated ahead of an evaluation, so every set just
    drops what was derived from the old value (``parameter_change``)."""
    def decorate(cls):
        for name in names:
            def getter(self, _n='_' + name):
                return getattr(self, _n, None)

            def setter(self, value, _n='_' + name, **update_flags):
                setattr(self, _n, value)
                self.parameter_change()
            setattr(cls, 'get_' + name, getter)
            setattr(cls, 'set_' + name, setter)
            setattr(cls, name, property(getter, setter))
        return cls
    return decorate


def numerical_derivative(f, x, h=1e-6):
    """central differences of a scalar- or vector-valued ``f`` in every component of ``x``"""
    x = np.array(x, dtype=float)
    cols = []
    for i in range(x.size):
        step = np.zeros_like(x)
        step.flat[i] = h * max(1.0, abs(x.flat[i]))
        cols.append((np.asarray(f(x + step)) - np.asarray(f(x - step))) / (2 * step.flat[i]))
    return np.moveaxis(np.array(cols), 0, -1)


class GenericFunction(object):
    """pinning of the argument, memo of what was evaluated, derivative checks"""

    def parameter_change(self):
        self.__dict__.pop('_memo', None)
        ev = self.__dict__.pop('_evaluator', None)
        if ev is not None:
            ev.close()

    def __call__(self, x):
        bound = copy.copy(self)
        bound._x = np.array(x, dtype=float)
        bound._memo = {}
        return bound

    def _at(self, x):
        """argument of this call: the pinned one unless another is given (the pinned value handed in
        explicitly -- one memoised method calling another with its own argument -- counts as pinned)"""
        pinned = getattr(self, '_x', None)
        if x is None:
            if pinned is None:
                raise TypeError('no argument given and none pinned; use fn(x) first')
            return pinned, True
        if pinned is not None and (x is pinned or (np.shape(x) == pinned.shape and np.array_equal(x, pinned))):
            return pinned, True
        return np.asarray(x, dtype=float), False

    def _memoized(self, key, x, compute):
        x, pinned = self._at(x)
        if not pinned:
            return compute(x)
        memo = self.__dict__.setdefault('_memo', {})
        if key not in memo:
            memo[key] = compute(x)
        return memo[key]

    # the reference's self-test of analytic against numerical derivatives (functions.py:148-200)
    def _derivative_matches(self, fun, der, around, renorm, prec, what):
        err = np.abs(numerical_derivative(fun, around) - np.asarray(der(around)))
        if renorm is True:
            err = err / np.abs(fun(around))
        elif renorm is not False:
            err = err / abs(renorm)
        if np.max(err) > prec:
            print('numerical derivative does not fit analytic derivative: {} {} - difference {}'.format(
                what, type(self).__name__, np.max(err)))
            return False
        return True

    def check_d(self, around, renorm=False, prec=1.e-8):
        return self._derivative_matches(self.f, self.d, around, renorm, prec, '1st derivative')

    def check_dd(self, around, renorm=False, prec=1.e-8):
        return self._derivative_matches(self.d, self.dd, around, renorm, prec, '2nd derivative')

    def check_derivatives(self, around, renorm=False, prec=1.e-8):
        return bool(self.check_d(around, renorm, prec)) & bool(self.check_dd(around, renorm, prec))


def cached(method):
    """Decorator for the methods of a user-written function object (reference functions.py:46-93): on an object
    pinned to an argument, ``fn = F(x)``, the method is evaluated once and remembered -- ``fn.f()``, ``fn.d()``
    --, with any other argument it is evaluated afresh and nothing is remembered."""
    def method_with_memo(self, x=None):
        return self._memoized(method.__name__, x, lambda arg: method(self, arg))
    method_with_memo.__name__ = method.__name__
    method_with_memo.__doc__ = method.__doc__
    return method_with_memo


class DoublyDerivableFunction(GenericFunction):
    """base of user-written functions with ``f``, ``d``, ``dd`` (reference functions.py:96-146)"""

    def f(self, x=None):
        raise NotImplementedError

    def d(self, x=None):
        raise NotImplementedError

    def dd(self, x=None):
        raise NotImplementedError


class _OnMesh(object):
    """``omega`` of a block is the mesh of its default model / kernel; setting it re-tabulates that"""
    _mesh_owner = '_D'

    def get_omega(self):
        return getattr(self, self._mesh_owner).omega

    def set_omega(self, omega, **update_flags):
        owner = getattr(self, self._mesh_owner)
        owner.omega = omega
        if update_flags.get('update_D', update_flags.get('update_K', True)):
            owner.parameter_change()
        self.parameter_change()

    omega = property(get_omega, set_omega)


# ---------------------------------------------------------------- chi2 ----
@parameters('K', 'G', 'err')
class Chi2(_OnMesh, GenericFunction):
    """data misfit as a function of the hidden image H (reference functions.py:200-333)"""
    _mesh_owner = '_K'

    def __init__(self, K=None, G=None, err=None):
        self._K, self._G, self._err = K, G, err

    def get_data_variable(self):
        return self.K.data_variable

    def set_data_variable(self, data_variable, update_K=True, **update_flags):
        self.K.data_variable = data_variable
        if update_K:
            self.K.parameter_change()
        self.parameter_change()

    data_variable = property(get_data_variable, set_data_variable)


class NormalChi2(Chi2):
    r""":math:`\chi^2 = \sum_i (G_i - \sum_j K_{ij} H_j)^2/\sigma_i^2` (reference functions.py:336-377).
    The device evaluates it in the singular space of K: :math:`|c\,V^T H - \hat g|^2 + c_\perp`."""

    def _device(self):
        ev = self.__dict__.get('_evaluator')
        if ev is None:
            n = np.asarray(self.K.K).shape[1]
            ev = self._evaluator = Evaluator(self.K, self.G, self.err, np.ones(n), ENTROPY_NORMAL)
        return ev

    def f(self, H=None):
        return self._memoized('f', H, lambda H: float(self._device().at_H(H, want=('chi2',))['chi2']))

    def d(self, H=None):
        return self._memoized('d', H, lambda H: 2.0 * self._device().at_H(H, want=('q',))['q'])

    def dd(self, H=None):
        """2 K^T diag(1/err^2) K = 2 V M V^T, a constant of the data set"""
        ev = self._device()
        return 2.0 * np.dot(ev.V, np.dot(ev.M, ev.V.T))


# ------------------------------------------------------------- entropy ----
@parameters('D')
class Entropy(_OnMesh, GenericFunction):
    kind = None

    def __init__(self, D=None):
        self._D = D

    def _device(self, H):
        return self._memoized('SdSddS', H, lambda H: device.entropy(self.kind, H, self.D.D))

    def f(self, H=None):
        return float(self._device(H)[0])

    def d(self, H=None):
        return self._device(H)[1]

    def dd(self, H=None):
        return np.diag(self._device(H)[2])


class NormalEntropy(Entropy):
    r""":math:`S = \sum_i (H_i - D_i - H_i \log(H_i/D_i))` (reference functions.py:491-520)"""
    kind = ENTROPY_NORMAL


class PlusMinusEntropy(Entropy):
    r""":math:`S = S_n(H^+) + S_n(H^-)`, :math:`H = H^+ - H^-` (reference functions.py:523-564)"""
    kind = ENTROPY_PLUSMINUS


# ---------------------------------------------------------------- H(v) ----
@parameters('D', 'K')
class GenericH_of_v(_OnMesh, GenericFunction):
    kind = None

    def __init__(self, D=None, K=None):
        self._D, self._K = D, K

    def _device(self):
        ev = self.__dict__.get('_evaluator')
        if ev is None:
            n_tau = np.asarray(self.K.K).shape[0]
            ev = self._evaluator = Evaluator(self.K, np.zeros(n_tau), np.ones(n_tau), self.D.D, self.kind)
        return ev

    def _Hw(self, v):
        return self._memoized('Hw', v, lambda v: self._device().at_v(v, 0.0, want=('H', 'w')))

    def f(self, v=None):
        return self._Hw(v)['H']

    def d(self, v=None):
        """dH_i/dv_k = w_i V_ik"""
        return self._Hw(v)['w'][:, np.newaxis] * self.K.V

    def dd(self, v=None):
        """d2H_i/dv_k dv_l = H_i V_ik V_il (both parametrisations: d2H/du2 = H)"""
        V = self.K.V
        return self._Hw(v)['H'][:, np.newaxis, np.newaxis] * V[:, :, np.newaxis] * V[:, np.newaxis, :]


class NormalH_of_v(GenericH_of_v):
    r"""Bryan's parametrisation :math:`H = D e^{Vv}` (reference functions.py:720-755)"""
    kind = ENTROPY_NORMAL

    def inv(self, H):
        return np.dot(self.K.V.T, safelog(np.asarray(H, dtype=float) / self.D.D))


class PlusMinusH_of_v(GenericH_of_v):
    r""":math:`H = D (e^{Vv} - e^{-Vv})` (reference functions.py:758-796)"""
    kind = ENTROPY_PLUSMINUS

    def inv(self, H):
        H, D = np.asarray(H, dtype=float), self.D.D
        return np.dot(self.K.V.T, safelog((H + np.sqrt(H ** 2 + 4 * D ** 2)) / (2 * D)))


# ---------------------------------------------------------------- A(H) ----
class GenericA_of_H(GenericFunction):
    """linear map from the hidden image to the spectral function, A = B H"""

    def get_omega(self):
        return self._omega

    def set_omega(self, omega, update_A_of_H=True):
        self._omega = omega
        if update_A_of_H:
            self.parameter_change()

    omega = property(get_omega, set_omega)

    def d(self, H=None):
        B = self.matrix()
        return np.diag(1.0 / self._omega.delta) if B is None else B

    def dd(self, H=None):
        n = len(self._omega)
        return np.zeros((n, n, n))


class IdentityA_of_H(GenericA_of_H):
    """A = H / delta_omega (reference functions.py:937-964)"""

    def __init__(self, omega):
        self._omega = omega

    def f(self, H=None):
        return self._at(H)[0] / self._omega.delta

    def inv(self, A):
        return np.asarray(A) * self._omega.delta

    def matrix(self):
        return None


class PreblurA_of_H(GenericA_of_H):
    """A = B H with the blur matrix of width ``b`` (reference functions.py:967-1023); belongs with
    ``PreblurKernel``.  Whole launches are mapped on the device (``mxe_apply_output_map``)."""

    def __init__(self, b, omega):
        self._omega, self._b = omega, b
        self.parameter_change()

    def parameter_change(self):
        super(PreblurA_of_H, self).parameter_change()
        self._B = get_preblur(self._omega, self._b)

    def f(self, H=None):
        return np.dot(self._at(H)[0], self._B.T)

    def inv(self, A):
        return np.linalg.lstsq(self._B, A, rcond=None)[0]

    def matrix(self):
        return self._B

    def get_b(self):
        return self._b

    def set_b(self, b, update_A_of_H=True):
        self._b = b
        if update_A_of_H:
            self.parameter_change()

    b = property(get_b, set_b)
