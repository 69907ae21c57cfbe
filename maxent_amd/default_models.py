"""Default models D(omega); ``.D`` already contains delta-omega.

Public names and semantics of the reference's ``default_models`` module
(reference python/default_models.py:26-115).
"""

import numpy as np


class BaseDefaultModel(object):
    def __init__(self, omega):
        self.omega = omega
        self._D = None

    @property
    def D(self):
        return self._D

    def parameter_change(self):
        self._fill_values()

    def _fill_values(self):
        raise NotImplementedError('Use a subclass of BaseDefaultModel')

    def __len__(self):
        return len(self._D)


class FlatDefaultModel(BaseDefaultModel):
    """D_i = delta_i / sum(delta) (default_models.py:48-63)."""

    def __init__(self, omega):
        super(FlatDefaultModel, self).__init__(omega)
        self._fill_values()

    def _fill_values(self):
        delta = self.omega.delta
        self._D = np.ones(len(delta)) / np.sum(delta) * delta


class DataDefaultModel(BaseDefaultModel):
    """tabulated default model, interpolated onto ``omega`` if needed
    (default_models.py:66-93)."""

    def __init__(self, default, omega_in, omega=None):
        if omega is None:
            omega = omega_in
        super(DataDefaultModel, self).__init__(omega)
        self.omega_in = omega_in
        self.default = default
        self._fill_values()

    def _fill_values(self):
        same = len(self.omega_in) == len(self.omega) and \
            np.all(np.asarray(self.omega_in) == np.asarray(self.omega))
        if same:
            D = np.asarray(self.default, dtype=float)
        else:
            D = np.interp(np.asarray(self.omega), np.asarray(self.omega_in),
                          np.asarray(self.default, dtype=float))
        self._D = D * self.omega.delta


class FileDefaultModel(DataDefaultModel):
    """two-column text file: omega, D (default_models.py:96-115; the
    reference's constructor is broken, this one works)."""

    def __init__(self, filename, omega=None):
        from .omega_meshes import DataOmegaMesh
        data = np.loadtxt(filename)
        super(FileDefaultModel, self).__init__(
            default=data[:, 1], omega_in=DataOmegaMesh(data[:, 0]),
            omega=omega)
