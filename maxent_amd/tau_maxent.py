"""``TauMaxEnt``: MaxEnt with the imaginary-time kernel (user facade).

Keeps the reference's surface (reference python/tau_maxent.py:37-356): owns a
:class:`MaxEntLoop` and shadows its attributes (``tm.omega = ...``,
``tm.alpha_mesh = ...``), default 100-point hyperbolic omega mesh on
[-10, 10] with a flat default model, setters for G(tau) from arrays or text
files, scalar / per-tau errors and full covariance matrices (the problem is
rotated into the covariance eigenbasis).  ``set_G_tau`` / ``set_G_iw`` need
TRIQS Green-function objects and are not provided.
"""

import copy

import numpy as np

from .default_models import FlatDefaultModel
from .kernels import TauKernel
from .maxent_loop import MaxEntLoop
from .omega_meshes import HyperbolicOmegaMesh


class TauMaxEnt(object):
    maxent_loop = None      # needed by the attribute shadowing below

    def __init__(self, cov_threshold=1.e-14, svd_backend='host', **kwargs):
        self.maxent_loop = MaxEntLoop(**kwargs)
        omega = HyperbolicOmegaMesh()
        self.D = FlatDefaultModel(omega)
        # svd_backend='device': kernel fill + SVD on the GPU (mxe_kernel_svd)
        self.K = TauKernel([0, 1], omega, svd_backend=svd_backend)      # placeholder tau grid
        self.omega = omega
        self.cov_threshold = cov_threshold

    # attributes of the loop can be used as if they were ours
    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, 'maxent_loop'), name)

    def __setattr__(self, name, value):
        if hasattr(self.maxent_loop, name):
            setattr(self.maxent_loop, name, value)
        else:
            object.__setattr__(self, name, value)

    def set_G_tau(self, *args, **kwargs):
        raise NotImplementedError('set_G_tau needs TRIQS Green functions; '
                                  'use set_G_tau_data or set_G_tau_file')

    set_G_iw = set_G_tau

    # ---- data ------------------------------------------------------------
    def set_G_tau_data(self, tau, G_tau):
        """G(tau) from arrays (reference tau_maxent.py:181-196)."""
        assert len(tau) == len(G_tau), \
            "tau and G_tau don't have the same dimension"
        self.tau = tau
        self.G = G_tau
        self._transform(self._T, G_original_basis=True)

    def set_G_tau_file(self, filename, tau_col=0, G_col=1, err_col=None):
        """G(tau) (and optionally its error) from a text file
        (reference tau_maxent.py:198-225)."""
        dat = np.loadtxt(filename)
        self.tau = dat[:, tau_col]
        self.G = dat[:, G_col]
        if err_col is not None:
            self.err = dat[:, err_col]
            self._transform(None, G_original_basis=True)
        else:
            self._transform(self._T, G_original_basis=True)

    def set_error(self, error):
        """scalar or per-tau standard deviation; undoes a covariance rotation
        (reference tau_maxent.py:227-251)."""
        if not np.all(np.isreal(error)):
            raise Exception('complex error supplied, only real accepted')
        error = np.real(error)
        if np.ndim(error) == 0:
            self.err = float(error) * np.ones(np.shape(self.G))
        elif len(error) == len(self.G):
            self.err = np.asarray(error, dtype=float)
        else:
            raise Exception('Supply scalar error or with length of G_tau.')
        self._transform(None)

    def set_cov(self, cov):
        """full covariance matrix: diagonalise, drop eigenvalues below
        ``cov_threshold``, rotate G and K into the eigenbasis, use
        sqrt(eigenvalues) as errors (reference tau_maxent.py:253-288)."""
        self.cov = cov
        assert np.max(np.abs(cov - cov.transpose())) < 1.e-10, \
            'Supplied covariance matrix is not symmetric.'
        e, vec = np.linalg.eigh(cov)
        if np.any(e < 0):
            self.logtaker.error_message(
                'Eigenvalues of the covariance matrix are not all positive; '
                'they will be ignored. Smallest negative value: {}', np.min(e))
        keep = e >= self.cov_threshold
        e, vec = e[keep], vec[:, keep]
        self.err = None
        if hasattr(self.cost_function, '_G_orig'):
            self.G = self.cost_function._G_orig
        self._transform(vec.conjugate().transpose())
        self.err = np.sqrt(e)

    def set_cov_file(self, filename):
        self.set_cov(np.loadtxt(filename))

    # ---- rotation bookkeeping -------------------------------------------
    def _transform_G(self, T_to, T_from=None):
        if T_to is None:
            T = 1 if T_from is None else T_from.conjugate().transpose()
        elif T_from is None:
            T = T_to
        else:
            T = np.dot(T_to, T_from.conjugate().transpose())
        self.G = np.dot(T, self.G)

    def _transform(self, T_, G_original_basis=False):
        """rotate G and K from the left by the absolute rotation ``T_``
        (reference tau_maxent.py:303-325)."""
        if G_original_basis:
            self.cost_function._G_orig = copy.deepcopy(self.G)
        self._transform_G(T_, None if G_original_basis else self._T)
        self.K.transform(T_)
        self.K = self.K        # re-announce K to chi2 / H_of_v

    # ---- tau ----------------------------------------------------------------
    def get_tau(self):
        return self.maxent_loop.get_data_variable()

    def set_tau(self, tau, update_K=True, update_chi2=True, update_Q=True,
                update_H_of_v=True):
        """a new tau grid refills the kernel (and drops its SVD); setting the
        grid it already has is free -- this is what lets the element-wise
        driver reuse one SVD for all matrix elements, where the reference
        recomputes it per element."""
        old = self.maxent_loop.get_data_variable()
        same = old is not None and np.shape(old) == np.shape(tau) and \
            np.array_equal(np.asarray(old), np.asarray(tau))
        if same:
            return
        self.maxent_loop.set_data_variable(tau, update_K=update_K,
                                           update_chi2=update_chi2,
                                           update_Q=update_Q,
                                           update_H_of_v=update_H_of_v)

    tau = property(get_tau, set_tau)

    @property
    def _T(self):
        return self.K._T
