"""``TauMaxEnt``: MaxEnt with the imaginary-time kernel (user facade).

Keeps the reference's surface (reference python/tau_maxent.py:37-356): owns a
:class:`MaxEntLoop` and shadows its attributes (``tm.omega = ...``,
``tm.alpha_mesh = ...``), default 100-point hyperbolic omega mesh on
[-10, 10] with a flat default model, setters for G(tau) from arrays or text
files, scalar / per-tau errors and full covariance matrices (the problem is
rotated into the covariance eigenbasis).  ``set_G_tau`` / ``set_G_iw`` need
TRIQS Green-function objects and are not provided.
"""

from copy import deepcopy

import numpy as np

from . import default_models, kernels, maxent_loop as loop_module, omega_meshes


class TauMaxEnt(object):
    maxent_loop = None      # needed by the attribute shadowing below

    def __init__(self, cov_threshold=1.e-14, svd_backend='host', **kwargs):
        self.maxent_loop = loop_module.MaxEntLoop(**kwargs)
        omega = omega_meshes.HyperbolicOmegaMesh()
        self.D = default_models.FlatDefaultModel(omega)
        # svd_backend='device': kernel fill + SVD on the GPU (mxe_kernel_svd)
        self.K = kernels.TauKernel([0, 1], omega, svd_backend=svd_backend)      # placeholder tau grid
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

    # ---- data, errors and the rotation of the data space -------------------
    # State: ``G`` (the data as the solver sees them, possibly rotated), ``cost_function._G_orig``
    # (as supplied) and the absolute rotation ``K._T`` of the kernel (None: unrotated).  Two moves:
    # ``_adopt_data`` -- new data arrive in the original basis and are brought into the current
    # rotation; ``_rotate_to`` -- data and kernel go from the current rotation to another one.
    # (reference tau_maxent.py:181-325)

    def _move_data(self, T_to, T_from):
        """data from rotation ``T_from`` to rotation ``T_to`` (None = unrotated): back, then forth"""
        G = self.G
        if T_from is not None:
            G = np.dot(T_from.conjugate().transpose(), G)
        if T_to is not None:
            G = np.dot(T_to, G)
        if G is not self.G:
            self.G = G

    def _announce_kernel(self, T):
        self.K.transform(T)          # sets K._T
        self.K = self.K              # chi2 and H_of_v hear about the changed kernel

    def _adopt_data(self, keep_rotation=True):
        T = self._T if keep_rotation else None
        self.cost_function._G_orig = deepcopy(self.G)
        self._move_data(T, None)
        self._announce_kernel(T)

    def _rotate_to(self, T):
        self._move_data(T, self._T)
        self._announce_kernel(T)

    def _transform(self, T_, G_original_basis=False):
        """the reference's name for the two moves (tau_maxent.py:303-325)"""
        if G_original_basis:
            self.cost_function._G_orig = deepcopy(self.G)
            self._move_data(T_, None)
            self._announce_kernel(T_)
        else:
            self._rotate_to(T_)

    def set_G_tau_data(self, tau, G_tau):
        """G(tau) from arrays (reference tau_maxent.py:181-196)"""
        if len(tau) != len(G_tau):
            raise AssertionError("tau and G_tau don't have the same dimension")
        self.tau, self.G = tau, G_tau
        self._adopt_data()

    def set_G_tau_file(self, filename, tau_col=0, G_col=1, err_col=None):
        """G(tau), optionally with its error bar, from the columns of a text file
        (reference tau_maxent.py:198-225); a file that brings errors ends any rotation"""
        table = np.loadtxt(filename)
        self.tau, self.G = table[:, tau_col], table[:, G_col]
        if err_col is not None:
            self.err = table[:, err_col]
        self._adopt_data(keep_rotation=err_col is None)

    def set_error(self, error):
        """one standard deviation for all tau or one per tau; ends a covariance rotation
        (reference tau_maxent.py:227-251)"""
        if not np.all(np.isreal(error)):
            raise Exception('complex error supplied, only real accepted')
        sigma = np.real(error) * np.ones(np.shape(self.G)) if np.ndim(error) == 0 \
            else np.asarray(np.real(error), dtype=float)
        if sigma.shape != np.shape(self.G):
            raise Exception('Supply scalar error or with length of G_tau.')
        self.err = sigma
        self._rotate_to(None)

    def set_cov(self, cov):
        """Full covariance matrix of the data (reference tau_maxent.py:253-288): the problem is rotated
        into the eigenbasis of ``cov`` (eigenvalues below ``cov_threshold`` dropped), where the errors are
        the square roots of the eigenvalues.  As in the reference, the data are first reset to the supplied
        ones and then moved by the hop from the PREVIOUS rotation to the new one -- after an earlier
        ``set_cov`` that is not the new rotation alone; the element-wise drivers rely on reproducing it."""
        given = cov
        known = self.__dict__.get('_cov_eig')
        if known is not None and known[0] is given and known[1] == self.cov_threshold and \
                np.array_equal(known[2], np.asarray(given)):        # (the same object with the same CONTENT: an in-place edit recomputes)
            # the same matrix again (one covariance for all matrix elements of an element-wise job): its
            # eigenbasis, and the SAME rotation object -- the kernel then has nothing to do, and the batch
            # solver sees one data set instead of one per element
            cov, sigma, T = known[2:]
        else:
            cov = np.array(cov)                   # (a private copy: the cache compares against it)
            if np.max(np.abs(cov - cov.transpose())) >= 1.e-10:
                raise AssertionError('Supplied covariance matrix is not symmetric.')
            var, vec = np.linalg.eigh(cov)
            if var.min() < 0:
                self.logtaker.error_message(
                    'Eigenvalues of the covariance matrix are not all positive; they will be ignored. '
                    'Smallest negative value: {}', var.min())
            keep = var >= self.cov_threshold
            sigma, T = np.sqrt(var[keep]), vec[:, keep].conjugate().transpose()
            object.__setattr__(self, '_cov_eig', (given, self.cov_threshold, cov, sigma, T))
        self.cov = cov
        self.err = None              # no chi2 with stale errors while the kernel changes
        if hasattr(self.cost_function, '_G_orig'):
            self.G = self.cost_function._G_orig
        self._rotate_to(T)
        self.err = sigma

    def set_cov_file(self, filename):
        self.set_cov(np.loadtxt(filename))

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
