"""Frequency meshes.

Public names, constructor arguments and point sets of the reference's
``omega_meshes`` module (reference python/omega_meshes.py:25-222); the attribute
``delta`` holds the trapezoid weights (:54-62).  Built on :class:`maxent_amd.meshes.Mesh`:
each class is its point formula.
"""

import numpy as np

from .meshes import Mesh


class BaseOmegaMesh(Mesh):
    _defaults = dict(omega_min=-10, omega_max=10, n_points=100)

    @classmethod
    def _check(cls, omega_min=None, omega_max=None, **rest):
        if omega_min is not None and omega_max is not None and omega_min > omega_max:
            raise Exception('omega_min must be smaller than omega_max')

    @classmethod
    def _points(cls, omega_min, omega_max, n_points, **shape):
        return cls._grid(omega_min, omega_max, n_points, **shape), \
            dict(omega_min=omega_min, omega_max=omega_max, n_points=n_points, **shape)

    @staticmethod
    def _grid(lo, hi, n):
        return np.zeros(n)

    @property
    def delta(self):
        """half the distance between the two neighbours (end points: half the first / last interval)"""
        cached = self.__dict__.get('_delta')
        if cached is None or len(cached) != len(self):
            w = np.asarray(self)
            span = np.concatenate(([w[1] - w[0]], w[2:] - w[:-2], [w[-1] - w[-2]]))
            cached = self.__dict__['_delta'] = 0.5 * span
        return cached


class LinearOmegaMesh(BaseOmegaMesh):
    """equidistant (omega_meshes.py:65-88)"""
    _grid = staticmethod(lambda lo, hi, n: np.linspace(lo, hi, n))


class DataOmegaMesh(BaseOmegaMesh):
    """the user's own points (omega_meshes.py:91-110)"""
    _defaults = dict(data=None)

    @classmethod
    def _check(cls, data=None):
        if data is None:
            raise TypeError('DataOmegaMesh needs the mesh points')

    @classmethod
    def _points(cls, data):
        data = np.asarray(data, dtype=float)
        return data, dict(omega_min=np.min(data), omega_max=np.max(data), n_points=len(data))


def _tan_midpoints(lo, hi, n, cut):
    # midpoints of n intervals whose edges are tan-spaced between the cut-off angles
    edges = np.tan(np.pi * (np.linspace(0, 1, n + 1) * (1. - 2 * cut) + cut - 0.5))
    edges = lo + (hi - lo) * ((edges - edges[0]) / (edges[-1] - edges[0]))
    return 0.5 * (edges[1:] + edges[:-1])


def _onto(w, lo, hi):
    # first point on lo, last on hi (the order of operations fixes the last bit of the points, and the
    # kernel matrix is compared bit for bit with the reference's)
    return (w - w[0]) / (w[-1] - w[0]) * (hi - lo) + lo


class LorentzianOmegaMesh(BaseOmegaMesh):
    """tan spacing, first and last point on omega_min / omega_max (omega_meshes.py:113-152)"""
    _defaults = dict(omega_min=-10, omega_max=10, n_points=100, cut=0.01)
    _grid = staticmethod(lambda lo, hi, n, cut: _onto(_tan_midpoints(lo, hi, n, cut), lo, hi))


class LorentzianSmallerOmegaMesh(BaseOmegaMesh):
    """tan spacing, the interval midpoints as they are (omega_meshes.py:155-196)"""
    _defaults = dict(omega_min=-10, omega_max=10, n_points=100, cut=0.01)
    _grid = staticmethod(_tan_midpoints)


def _hyperbola(lo, hi, n):
    u = np.linspace(-1, 1, n)
    w = np.sign(u) * (np.sqrt(1 + u ** 2) - 1)
    return lo + (hi - lo) * (w - w[0]) / (w[-1] - w[0])


class HyperbolicOmegaMesh(BaseOmegaMesh):
    """sign(u) (sqrt(1 + u^2) - 1), u equidistant in [-1, 1] (omega_meshes.py:199-222)"""
    _grid = staticmethod(_hyperbola)
