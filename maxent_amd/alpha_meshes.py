"""Alpha meshes: the hyper-parameter values of a scan, largest first.

Public names and constructor arguments of the reference's ``alpha_meshes`` module
(reference python/alpha_meshes.py:26-103), built on :class:`maxent_amd.meshes.Mesh`.
"""

import numpy as np

from .meshes import Mesh


class BaseAlphaMesh(Mesh):
    _defaults = dict(alpha_min=0.0001, alpha_max=20, n_points=20)
    _spacing = None

    @classmethod
    def _check(cls, alpha_min=None, alpha_max=None, n_points=None, **rest):
        if alpha_min is None or n_points is None or n_points <= 1:
            return
        if alpha_min > alpha_max:
            raise Exception('alpha_min must be smaller than alpha_max')
        if min(alpha_min, alpha_max) <= 0:
            raise Exception('All alpha values must be positive')

    @classmethod
    def _points(cls, alpha_min, alpha_max, n_points):
        ascending = np.zeros(n_points) if cls._spacing is None else cls._spacing(alpha_min, alpha_max, n_points)
        return ascending[::-1], dict(alpha_min=alpha_min, alpha_max=alpha_max, n_points=n_points)


class LogAlphaMesh(BaseAlphaMesh):
    """equidistant in log(alpha) (alpha_meshes.py:68-85)"""
    _spacing = staticmethod(lambda lo, hi, n: np.logspace(np.log10(lo), np.log10(hi), n))


class LinearAlphaMesh(BaseAlphaMesh):
    """equidistant in alpha (alpha_meshes.py:88-103)"""
    _spacing = staticmethod(np.linspace)


class DataAlphaMesh(BaseAlphaMesh):
    """the user's own values, sorted (alpha_meshes.py:46-65)"""
    _defaults = dict(data=None)

    @classmethod
    def _check(cls, data=None):
        if data is None:
            raise TypeError('DataAlphaMesh needs the alpha values')
        if np.size(data) > 1 and np.min(data) <= 0:
            raise Exception('All alpha values must be positive')

    @classmethod
    def _points(cls, data):
        data = np.sort(np.asarray(data, dtype=float))
        return data[::-1], dict(alpha_min=data[0], alpha_max=data[-1], n_points=len(data))
