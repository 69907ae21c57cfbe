"""Alpha meshes: descending arrays of the hyper-parameter.

Public names and semantics of the reference's ``alpha_meshes`` module
(reference python/alpha_meshes.py:26-103).
"""

import numpy as np


class BaseAlphaMesh(np.ndarray):
    def __new__(cls, alpha_min=0.0001, alpha_max=20, n_points=20, *args,
                **kwargs):
        return super(BaseAlphaMesh, cls).__new__(cls, shape=(n_points,))

    def __init__(self, alpha_min=0.0001, alpha_max=20, n_points=20):
        if n_points > 1:
            if alpha_min > alpha_max:
                raise Exception('alpha_min must be smaller than alpha_max')
            if (alpha_min <= 0) or (alpha_max <= 0):
                raise Exception('All alpha values must be positive')
        self.alpha_min = alpha_min
        self.alpha_max = alpha_max
        self.n_points = n_points

    def __array_finalize__(self, obj):
        for name in ('alpha_min', 'alpha_max', 'n_points'):
            if obj is not None and hasattr(obj, name):
                setattr(self, name, getattr(obj, name))


class DataAlphaMesh(BaseAlphaMesh):
    """user-supplied values, sorted descending (alpha_meshes.py:46-65)."""

    def __new__(cls, data):
        return super(DataAlphaMesh, cls).__new__(cls, np.min(data),
                                                 np.max(data), len(data))

    def __init__(self, data):
        super(DataAlphaMesh, self).__init__(np.min(data), np.max(data),
                                            len(data))
        self[:] = sorted(data, reverse=True)


class LogAlphaMesh(BaseAlphaMesh):
    """logarithmic spacing, largest first (alpha_meshes.py:68-85)."""

    def __init__(self, alpha_min=0.0001, alpha_max=20, n_points=20):
        super(LogAlphaMesh, self).__init__(alpha_min, alpha_max, n_points)
        self[:] = np.logspace(np.log10(alpha_min), np.log10(alpha_max),
                              n_points)[::-1]


class LinearAlphaMesh(BaseAlphaMesh):
    """linear spacing, largest first (alpha_meshes.py:88-103)."""

    def __init__(self, alpha_min=0.0001, alpha_max=20, n_points=20):
        super(LinearAlphaMesh, self).__init__(alpha_min, alpha_max, n_points)
        self[:] = np.linspace(alpha_min, alpha_max, n_points)[::-1]
