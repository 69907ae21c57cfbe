"""Default models.  ``model.D`` is the density times delta-omega, the quantity
the entropy is written in (reference python/default_models.py:26-115; public
names and constructor arguments as there).
"""

import numpy as np


class BaseDefaultModel(object):
    """``D`` is tabulated by ``_fill_values`` -- on first use and on ``parameter_change()`` -- and kept in
    between, also across a new ``omega`` (the reference's protocol, test/python/default_models.py:56-60).  A
    model is written either as a ``density(omega)`` or, like in the reference (doc/guide/customization.rst),
    by overriding ``_fill_values`` to set ``self._D``."""

    def __init__(self, omega):
        self.omega = omega

    def density(self, omega):
        raise NotImplementedError('Use a subclass of BaseDefaultModel')

    def _fill_values(self):
        self._D = np.asarray(self.density(self.omega), dtype=float) * self.omega.delta

    def parameter_change(self):
        self._fill_values()

    @property
    def D(self):
        if '_D' not in self.__dict__:
            self._fill_values()
        return self._D

    def __len__(self):
        return len(self.omega)


class FlatDefaultModel(BaseDefaultModel):
    """constant density, normalised on the mesh (default_models.py:48-63)"""

    def density(self, omega):
        return np.full(len(omega), 1.0 / np.sum(omega.delta))


class DataDefaultModel(BaseDefaultModel):
    """tabulated density ``default`` on ``omega_in``; linear interpolation onto another mesh
    (default_models.py:66-93)"""

    def __init__(self, default, omega_in, omega=None):
        self.default, self.omega_in = default, omega_in
        super(DataDefaultModel, self).__init__(omega_in if omega is None else omega)

    def density(self, omega):
        src, dst = np.asarray(self.omega_in, dtype=float), np.asarray(omega, dtype=float)
        table = np.asarray(self.default, dtype=float)
        if src.shape == dst.shape and np.array_equal(src, dst):
            return table
        return np.interp(dst, src, table)


class FileDefaultModel(DataDefaultModel):
    """text file with the columns omega, density (default_models.py:96-115)"""

    def __init__(self, filename, omega=None):
        from .omega_meshes import DataOmegaMesh
        table = np.loadtxt(filename)
        super(FileDefaultModel, self).__init__(table[:, 1], DataOmegaMesh(table[:, 0]), omega)
