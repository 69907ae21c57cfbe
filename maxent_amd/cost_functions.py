"""Cost functions ``Q_alpha(v) = eta chi2(H(v)) / 2 - alpha S(H(v))`` (reference python/cost_functions/).

A cost function object holds the four building blocks (``chi2``, ``S``, ``H_of_v``,
``A_of_H``), forwards the problem's parameters to them under the reference's names
(``K, G, err, omega, data_variable, D``; reference cost_function.py:107-245), tells the
solver which kernel variant to run (``entropy_kind``) and evaluates itself on the GPU:

    ``Q.f(v)``, ``Q.d(v)``, ``Q.dd(v)``     value, gradient, curvature at ``v``
    ``b = Q(v)``                            ``v`` pinned: ``b.f()``, ``b.d()``, ``b.dd()``, and the blocks
                                            bound to the same point, ``b.chi2.f()``, ``b.S.f()``,
                                            ``b.H_of_v.f()``, ``b.A_of_H.f()`` (cost_function.py:73-85)

One call of ``mxe_eval_batch`` per point returns Q, chi2, S, H, u = V v, w = dH/du, q = V g,
g = eta (M h - b) + alpha v and the Gram matrices W = V^T diag(w) V, W2 = V^T diag(q H) V.  With
M = S U^T diag(1/err^2) U S the derivatives of the reference's modes are (include/maxent_hip.h):

    ======================================  ===========  ==============================
    mode                                    d            dd
    ======================================  ===========  ==============================
    MaxEntCostFunction, dA_projection = 2   W g          W eta M W + alpha W     (default)
    MaxEntCostFunction, dA_projection = 1   g            eta M W + alpha 1
    MaxEntCostFunction, dA_projection = 0   V g          V eta M W + alpha V
    MaxEntCostFunction, d_dv = True         W g          W eta M W + alpha W + W2
    BryanCostFunction                       g            eta M W
    ======================================  ===========  ==============================
"""

import numpy as np

from .evaluator import Evaluator
from .functions import (GenericFunction, NormalChi2, NormalEntropy, NormalH_of_v, IdentityA_of_H)

# which block owns which parameter of the problem, and the others that must hear about a change
_OWNERS = dict(K=('chi2', ('H_of_v',)), G=('chi2', ()), err=('chi2', ()),
               D=('S', ('H_of_v',)))
_BLOCKS = ('chi2', 'S', 'H_of_v', 'A_of_H')


class _Pinned(object):
    """a block of a pinned cost function: answers ``f() / d() / dd()`` from the evaluation the cost
    function made, falls back to the block itself for everything else"""

    def __init__(self, block, x, known):
        self._block, self._x, self._known = block, x, known

    def __getattr__(self, name):
        return getattr(self._block, name)

    def _get(self, what, x):
        if x is None and what in self._known:
            return self._known[what]()
        return getattr(self._block, what)(self._x if x is None else x)

    def f(self, x=None):
        return self._get('f', x)

    def d(self, x=None):
        return self._get('d', x)

    def dd(self, x=None):
        return self._get('dd', x)


class CostFunction(GenericFunction):
    def __init__(self, chi2=None, S=None, H_of_v=None, A_of_H=None, chi2_factor=1.0):
        self._chi2 = NormalChi2() if chi2 is None else chi2
        self._S = NormalEntropy() if S is None else S
        self._H_of_v = NormalH_of_v() if H_of_v is None else H_of_v
        if A_of_H is None:
            try:
                mesh = self._chi2.omega
            except Exception:
                mesh = None
            A_of_H = IdentityA_of_H(mesh)
        self._A_of_H = A_of_H
        self.chi2_factor = chi2_factor
        self._alpha = None

    def set_alpha(self, alpha):
        self._alpha = alpha

    @property
    def entropy_kind(self):
        if self._S.kind != self._H_of_v.kind:
            raise Exception('S and H_of_v do not belong together: use NormalEntropy with NormalH_of_v '
                            'or PlusMinusEntropy with PlusMinusH_of_v')
        return self._S.kind

    @property
    def G_orig(self):
        return getattr(self, '_G_orig', self.G)

    # ---- evaluation on the device -----------------------------------------
    def _device(self):
        ev = self.__dict__.get('_evaluator')
        if ev is None:
            ev = self._evaluator = Evaluator(self.K, self.G, self.err, self.D.D, self.entropy_kind)
        return ev

    def _eval(self, v):
        if self._alpha is None:
            raise Exception('call set_alpha on the cost function first')
        want = ('Q', 'chi2', 'S', 'H', 'u', 'w', 'q', 'g', 'W') + (('W2',) if self._needs_W2() else ())
        key = ('eval', float(self._alpha), want)
        return self._memoized(key, v, lambda v: self._device().at_v(v, self._alpha, self.chi2_factor, want))

    def _needs_W2(self):
        return False

    def f(self, v=None):
        return float(self._eval(v)['Q'])

    def dH(self, v=None):
        """dQ/dH = V g"""
        return self._eval(v)['q']

    def ddH(self, v=None):
        """d2Q/dH2 = eta V M V^T + alpha diag(1/w)"""
        ev, e = self._device(), self._eval(v)
        return self.chi2_factor * np.dot(ev.V, np.dot(ev.M, ev.V.T)) + self._alpha * np.diag(1.0 / e['w'])

    def d(self, v=None):
        raise NotImplementedError('Please use a subclass of CostFunction.')

    dd = d

    def __call__(self, v):
        bound = super(CostFunction, self).__call__(v)
        e = bound._eval(None)
        H = e['H']
        bound._chi2 = _Pinned(self._chi2, H, dict(f=lambda: float(e['chi2'])))
        bound._S = _Pinned(self._S, H, dict(f=lambda: float(e['S']), d=lambda: -e['u'],
                                            dd=lambda: -np.diag(1.0 / e['w'])))
        bound._H_of_v = _Pinned(self._H_of_v, bound._x, dict(
            f=lambda: H, d=lambda: e['w'][:, np.newaxis] * self.K.V))
        bound._A_of_H = _Pinned(self._A_of_H, H, {})
        return bound

    def parameter_change(self):
        super(CostFunction, self).parameter_change()

    # ---- the four blocks ---------------------------------------------------
    def _swap_block(self, name, block):
        setattr(self, '_' + name, block)
        self.parameter_change()

    # ---- omega and the data variable touch several blocks ------------------
    def get_omega(self):
        return self.chi2.K.omega

    def set_omega(self, omega, update_K=True, update_D=True, **update_flags):
        self.chi2.set_omega(omega, update_K=update_K)
        self.H_of_v.set_K(self.K)
        self.S.set_omega(omega, update_D=update_D)
        self.H_of_v.set_omega(omega, update_D=False)
        self.A_of_H.set_omega(omega, update_A_of_H=update_flags.get('update_A_of_H', True))
        self.parameter_change()

    omega = property(get_omega, set_omega)

    def get_data_variable(self):
        return self.chi2.K.data_variable

    def set_data_variable(self, data_variable, update_K=True, **update_flags):
        self.chi2.set_data_variable(data_variable, update_K=update_K)
        self.H_of_v.set_K(self.K)
        self.parameter_change()

    data_variable = property(get_data_variable, set_data_variable)


def _forward_parameter(name, owner, listeners):
    def getter(self):
        return getattr(getattr(self, owner), name)

    def setter(self, value, **update_flags):
        getattr(getattr(self, owner), 'set_' + name)(value)
        for other in listeners:
            getattr(getattr(self, other), 'set_' + name)(value)
        if name == 'D':
            self.A_of_H.set_omega(value.omega, update_A_of_H=update_flags.get('update_A_of_H', True))
        self.parameter_change()
    return getter, setter


def _block_accessors(name):
    def getter(self):
        return getattr(self, '_' + name)

    def setter(self, block, **update_flags):
        self._swap_block(name, block)
    return getter, setter


for _name, (_owner, _listeners) in _OWNERS.items():
    _g, _s = _forward_parameter(_name, _owner, _listeners)
    setattr(CostFunction, 'get_' + _name, _g)
    setattr(CostFunction, 'set_' + _name, _s)
    setattr(CostFunction, _name, property(_g, _s))
for _name in _BLOCKS:
    _g, _s = _block_accessors(_name)
    setattr(CostFunction, 'get_' + _name, _g)
    setattr(CostFunction, 'set_' + _name, _s)
    setattr(CostFunction, _name, property(_g, _s))
del _name, _owner, _listeners, _g, _s


class MaxEntCostFunction(CostFunction):
    """the general cost function (reference maxent_cost_function.py:26-165).  ``d_dv`` and
    ``dA_projection`` select the space the derivatives are written in (table in the module
    docstring); every mode has the same minimiser."""

    def __init__(self, d_dv=False, dA_projection=2, **kwargs):
        self.d_dv = d_dv
        self.dA_projection = dA_projection
        super(MaxEntCostFunction, self).__init__(**kwargs)

    def _needs_W2(self):
        return bool(self.d_dv)

    def d(self, v=None):
        e = self._eval(v)
        if self.d_dv or self.dA_projection == 2:
            return np.dot(e['W'], e['g'])
        return e['g'] if self.dA_projection == 1 else e['q']

    def dd(self, v=None):
        e, ev = self._eval(v), self._device()
        MW = self.chi2_factor * np.dot(ev.M, e['W'])
        if self.d_dv or self.dA_projection == 2:
            out = np.dot(e['W'], MW) + self._alpha * e['W']
            return out + e['W2'] if self.d_dv else out
        if self.dA_projection == 1:
            return MW + self._alpha * np.eye(len(MW))
        return np.dot(ev.V, MW) + self._alpha * ev.V


class BryanCostFunction(CostFunction):
    """Bryan's singular-space form (reference bryan_cost_function.py:26-144): normal entropy,
    ``d = g``, ``dd = eta M W``."""

    def __init__(self, chi2_factor=1.0):
        super(BryanCostFunction, self).__init__(chi2_factor=chi2_factor)

    def d(self, v=None):
        return self._eval(v)['g']

    def dd(self, v=None):
        return self.chi2_factor * np.dot(self._device().M, self._eval(v)['W'])

    def _swap_block(self, name, block):
        if name in ('H_of_v', 'A_of_H'):
            raise NotImplementedError('Cannot change {} in BryanCostFunction.'.format(name))
        super(BryanCostFunction, self)._swap_block(name, block)
