"""Frequency meshes (host-side inputs of the alpha scan).

Same public names and semantics as the reference's ``omega_meshes`` module
(reference python/omega_meshes.py:25-222): ndarray subclasses carrying
``omega_min``, ``omega_max``, ``n_points`` and a lazily computed trapezoid
weight vector ``delta``.
"""

import numpy as np


class BaseOmegaMesh(np.ndarray):
    """ndarray of frequencies with trapezoid weights ``delta``
    (reference omega_meshes.py:25-62)."""

    def __new__(cls, omega_min=-10, omega_max=10, n_points=100, *args,
                **kwargs):
        return super(BaseOmegaMesh, cls).__new__(cls, shape=(n_points,))

    def __init__(self, omega_min=-10, omega_max=10, n_points=100, *args,
                 **kwargs):
        if omega_min > omega_max:
            raise Exception('omega_min must be smaller than omega_max')
        self.omega_min = omega_min
        self.omega_max = omega_max
        self.n_points = n_points
        self._delta = None

    def __array_finalize__(self, obj):
        for name in ('omega_min', 'omega_max', 'n_points'):
            if obj is not None and hasattr(obj, name):
                setattr(self, name, getattr(obj, name))
        self._delta = None

    @property
    def delta(self):
        if self._delta is None:
            w = np.asarray(self)
            d = np.empty(len(w))
            d[1:-1] = (w[2:] - w[:-2]) / 2.0
            d[0] = (w[1] - w[0]) / 2.0
            d[-1] = (w[-1] - w[-2]) / 2.0
            self._delta = d
        return self._delta


class LinearOmegaMesh(BaseOmegaMesh):
    """equidistant mesh (reference omega_meshes.py:65-88)."""

    def __init__(self, omega_min=-10, omega_max=10, n_points=100):
        super(LinearOmegaMesh, self).__init__(omega_min, omega_max, n_points)
        self[:] = np.linspace(omega_min, omega_max, n_points)


class DataOmegaMesh(BaseOmegaMesh):
    """mesh from a user array (reference omega_meshes.py:91-110)."""

    def __new__(cls, data):
        return super(DataOmegaMesh, cls).__new__(cls, np.min(data),
                                                 np.max(data), len(data))

    def __init__(self, data):
        super(DataOmegaMesh, self).__init__(np.min(data), np.max(data),
                                            len(data))
        self[:] = data


def _lorentzian_points(omega_min, omega_max, n_points, cut):
    u = np.linspace(0, 1, n_points + 1)
    t = np.tan(np.pi * (u * (1. - 2 * cut) + cut - 0.5))
    t = (t - t[0]) / (t[-1] - t[0])
    w = omega_min + (omega_max - omega_min) * t
    return (w[:-1] + w[1:]) / 2.0


class LorentzianOmegaMesh(BaseOmegaMesh):
    """tan-spaced mesh, end points on omega_min/max
    (reference omega_meshes.py:113-152)."""

    def __init__(self, omega_min=-10, omega_max=10, n_points=100, cut=0.01):
        super(LorentzianOmegaMesh, self).__init__(omega_min, omega_max,
                                                  n_points)
        self.cut = cut
        w = _lorentzian_points(omega_min, omega_max, n_points, cut)
        self[:] = (w - w[0]) / (w[-1] - w[0]) * (omega_max - omega_min) \
            + omega_min

    def __array_finalize__(self, obj):
        super(LorentzianOmegaMesh, self).__array_finalize__(obj)
        if obj is not None and hasattr(obj, 'cut'):
            self.cut = obj.cut


class LorentzianSmallerOmegaMesh(BaseOmegaMesh):
    """tan-spaced mesh without the end-point rescaling
    (reference omega_meshes.py:155-196)."""

    def __init__(self, omega_min=-10, omega_max=10, n_points=100, cut=0.01):
        super(LorentzianSmallerOmegaMesh, self).__init__(omega_min, omega_max,
                                                         n_points)
        self.cut = cut
        self[:] = _lorentzian_points(omega_min, omega_max, n_points, cut)

    def __array_finalize__(self, obj):
        super(LorentzianSmallerOmegaMesh, self).__array_finalize__(obj)
        if obj is not None and hasattr(obj, 'cut'):
            self.cut = obj.cut


class HyperbolicOmegaMesh(BaseOmegaMesh):
    """sign(u)(sqrt(1+u^2)-1) spacing (reference omega_meshes.py:199-222)."""

    def __init__(self, omega_min=-10, omega_max=10, n_points=100):
        super(HyperbolicOmegaMesh, self).__init__(omega_min, omega_max,
                                                  n_points)
        u = np.linspace(-1, 1, n_points)
        w = np.sign(u) * (np.sqrt(1 + u ** 2) - 1)
        self[:] = omega_min + (omega_max - omega_min) * (w - w[0]) / \
            (w[-1] - w[0])
