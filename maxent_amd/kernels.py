"""Kernels of the analytic continuation and their SVD staging (host side).

``G_i = sum_j K_ij H_j`` with ``H = A * delta_omega``.  The kernel matrix is
filled and decomposed once on the host with numpy/LAPACK; the truncated
factors ``U, S, V`` are what :class:`maxent_amd.device.DeviceContext` stages
in HBM.  Public names and semantics follow the reference's ``kernels`` module
(reference python/kernels.py:37-413): ``KernelSVD``, ``Kernel``,
``DataKernel``, ``TauKernel``, ``PreblurKernel``.
"""

import numpy as np

from .preblur import get_preblur


class _one_blas_thread(object):
    """``threadpoolctl.threadpool_limits(1)`` where the package is there, nothing otherwise"""

    def __enter__(self):
        self._ctx = None
        try:
            from threadpoolctl import threadpool_limits
            self._ctx = threadpool_limits(limits=1)
            self._ctx.__enter__()
        except Exception:
            self._ctx = None
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def _fingerprint(a):
    """64-bit content hash of a contiguous float array (xxhash where the package is there, zlib.crc32 + adler32 otherwise)"""
    a = np.ascontiguousarray(a)
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(memoryview(a).cast('B'))
    except Exception:
        import zlib
        m = memoryview(a).cast('B')
        return (zlib.crc32(m) << 32) | zlib.adler32(m)


class _Recent(object):
    """the few most recently used results of an expensive function of array CONTENTS (kernel fill, SVD): a new TauMaxEnt /
    ElementwiseMaxEnt on the grids of an earlier one -- every iteration of a self-consistency loop, both workers of an
    element-wise run -- fills and decomposes the same 200 x 500 matrix again (reference elementwise_maxent.py:170-221: one fresh
    SVD per ELEMENT; here it was one per object, 6.5-7.7 ms of the 13-15 ms a fresh object took).  Entries are found by a content
    hash and confirmed by comparing the arrays; what they hand out is shared and READ-ONLY.

    Contract (differs from the reference, where ``K.K``, ``K.K_delta``, ``K.U``, ``K.S``, ``K.V`` are private writable arrays):
    in-place edits such as ``K.K[...] *= x`` raise numpy's "assignment destination is read-only".  Code that wants to change
    a kernel assigns a NEW array (``kernel._K = kernel.K * x``) or works on ``np.array(kernel.K)``; README.md, "Differences a
    user of the reference will notice"."""

    def __init__(self, size=4):
        import collections
        import threading
        self._d, self._size, self._lock = collections.OrderedDict(), size, threading.Lock()

    def get(self, key, confirm):
        with self._lock:
            hit = self._d.get(key)
            if hit is not None and confirm(hit[0]):
                self._d.move_to_end(key)
                return hit[1]
        return None

    def put(self, key, witness, value):
        with self._lock:
            self._d[key] = (witness, value)
            while len(self._d) > self._size:
                self._d.popitem(last=False)


def _frozen(a):
    a = np.asarray(a)
    a.setflags(write=False)
    return a


_recent_svd = _Recent()
_recent_fill = _Recent()


class KernelSVD(object):
    """Matrix with a lazily computed thin SVD ``K = U diag(S) V^T``.

    ``V`` is stored as ``n_omega x n_s`` (reference kernels.py:53-64).
    """

    #: 'host' (numpy / LAPACK, the reference's path) or 'device'
    #: (``mxe_kernel_svd``: fill, preblur product and a preconditioned one-sided
    #: Jacobi SVD on the GPU; kernels that know how they are filled -- TauKernel,
    #: PreblurKernel of a TauKernel -- implement ``_device_svd``)
    svd_backend = 'host'

    def __init__(self, K=None):
        self._U = self._S = self._V = None
        self._K = K
        self._last_threshold = None

    def _invalidate_svd(self):
        self._U = self._S = self._V = None

    def _device_svd(self):
        raise NotImplementedError('svd_backend="device" needs a kernel that can be '
                                  'filled on the device (TauKernel, PreblurKernel)')

    def svd(self):
        if self._U is None:
            if self.svd_backend == 'device':
                self._U, self._S, self._V = self._device_svd()
            elif self.svd_backend == 'host':
                # (one BLAS thread: the decomposition of a few hundred rows is no faster on many -- 8 threads were
                #  slower than 1 in BASELINE.md -- and a BLAS pool that spins up on every core of the host burns the
                #  CPU quota of a container: the process then stalls for most of a scheduler period, 70 ms, at some
                #  later point of the run)
                K = np.ascontiguousarray(self.K, dtype=float)
                key = (K.shape, _fingerprint(K))
                hit = _recent_svd.get(key, lambda K0: K0.shape == K.shape and np.array_equal(K0, K))
                if hit is None:
                    with _one_blas_thread():
                        U, S, Vh = np.linalg.svd(K, full_matrices=False)
                    hit = (_frozen(U), _frozen(S), _frozen(Vh.transpose()))
                    _recent_svd.put(key, K.copy(), hit)
                self._U, self._S, self._V = hit
            else:
                raise ValueError("svd_backend must be 'host' or 'device'")
        return (self._U, self._S, self._V)

    @property
    def U(self):
        return self.svd()[0]

    @property
    def S(self):
        return self.svd()[1]

    @property
    def V(self):
        return self.svd()[2]

    @property
    def K(self):
        return self._K

    def reduce_singular_space(self, threshold=1.e-14):
        """Drop singular values below the ABSOLUTE ``threshold``
        (reference kernels.py:101-122); a later call with a smaller
        threshold recomputes the SVD."""
        if self._last_threshold is not None:
            if threshold is None or threshold < self._last_threshold:
                self._invalidate_svd()
        self._last_threshold = threshold
        keep = np.where(self.S >= threshold)[0]
        if len(keep) < len(self._S):             # nothing to drop: U, S, V stay the objects they are
            self._U = self._U[:, keep]
            self._S = self._S[keep]
            self._V = self._V[:, keep]
        return self


class Kernel(KernelSVD):
    """Kernel on an omega mesh with an optional left rotation ``T``
    (covariance eigenbasis; reference kernels.py:125-180)."""

    def __init__(self):
        super(Kernel, self).__init__()
        self.omega = None
        self._T = None

    @property
    def rotation(self):
        """the absolute left rotation the matrix and U currently carry (None: unrotated)"""
        return self._T

    @property
    def K_delta(self):
        """K * delta_omega, never rotated: ``G_rec = K_delta A``."""
        return self._K_delta

    @property
    def data_variable(self):
        raise NotImplementedError('Use a subclass of Kernel')

    def parameter_change(self):
        self._fill_values()

    def _fill_values(self):
        raise NotImplementedError('Use a subclass of Kernel')

    def transform(self, T_):
        """Left-multiply K (and U) by ``T_``, given as the absolute rotation
        with respect to the unrotated kernel; ``None`` undoes it."""
        T = self._relative_rotation(T_, self._T)
        if T is None:
            return
        self._T = T_
        self._U = np.dot(T, self.U)
        self._K = np.dot(T, self._K)

    @staticmethod
    def _relative_rotation(T_to, T_from):
        """the matrix that takes the kernel from rotation ``T_from`` to ``T_to``; None if there is
        nothing to do (the same rotation object, or unrotated to unrotated)"""
        if T_to is T_from:
            return None
        if T_to is None:
            return T_from.conjugate().transpose()
        if T_from is None:
            return T_to
        return np.dot(T_to, T_from.conjugate().transpose())


class DataKernel(Kernel):
    """Kernel given as a matrix (reference kernels.py:183-207)."""

    def __init__(self, data_variable, omega, K):
        super(DataKernel, self).__init__()
        self._data_variable = data_variable
        self.omega = omega
        self._K = K
        self._K_delta = K * omega.delta[np.newaxis, :]

    @property
    def data_variable(self):
        return self._data_variable


class TauKernel(Kernel):
    r"""Fermionic imaginary-time kernel
    :math:`K(\tau,\omega) = -e^{-\tau\omega}/(1+e^{-\beta\omega})`
    (reference kernels.py:210-280).  ``beta`` defaults to ``tau[-1]``."""

    def __init__(self, tau, omega, beta=None, svd_backend='host'):
        super(TauKernel, self).__init__()
        self.tau = tau
        self.omega = omega
        self.beta = beta
        self.svd_backend = svd_backend
        self._fill_values()

    def _device_args(self):
        tau = np.asarray(self.tau, dtype=float)
        beta = tau[-1] if self.beta is None else self.beta
        return tau, np.asarray(self.omega, dtype=float), self.omega.delta, beta

    def _device_svd(self, preblur_b=0.0):
        """U, S, V of the UNROTATED kernel from the device: everything the QR stage kept
        (singular values down to eps * sigma_max; the reference's LAPACK values below
        that are rounding noise), ``reduce_singular_space`` cuts as usual."""
        from . import device
        tau, w, delta, beta = self._device_args()
        r = device.kernel_svd(tau, w, delta, beta, [preblur_b], threshold=0.0)[0]
        return r['U'], r['S'], r['V']

    def _fill_values(self):
        self._invalidate_svd()
        tau = np.asarray(self.tau, dtype=float)
        w = np.asarray(self.omega, dtype=float)
        beta = tau[-1] if self.beta is None else self.beta
        delta = np.asarray(self.omega.delta, dtype=float)
        key = (tau.tobytes(), w.tobytes(), delta.tobytes(), float(beta))
        hit = _recent_fill.get(key, lambda _: True)               # (the key IS the contents)
        if hit is not None:
            self._K, self._K_delta = hit
            T = self._T
            self._T = None
            self.transform(T)
            return
        ww = w[np.newaxis, :] * np.ones((len(tau), 1))
        tt = tau[:, np.newaxis] * np.ones((1, len(w)))
        pos = ww >= 0.0
        K = np.empty(ww.shape)
        # two algebraically equal forms, each overflow-free on its half-axis
        K[pos] = -np.exp(-ww[pos] * tt[pos]) / (np.exp(-beta * ww[pos]) + 1.0)
        neg = np.logical_not(pos)
        K[neg] = -np.exp(ww[neg] * (beta - tt[neg])) / \
            (1.0 + np.exp(beta * ww[neg]))
        self._K = _frozen(K)
        self._K_delta = _frozen(K * self.omega.delta[np.newaxis, :])
        _recent_fill.put(key, None, (self._K, self._K_delta))
        T = self._T
        self._T = None
        self.transform(T)

    @property
    def data_variable(self):
        return self.tau

    @data_variable.setter
    def data_variable(self, value):
        self.tau = value


class PreblurKernel(Kernel):
    """``K' = K diag(delta) B`` for the preblur formalism; ``K_delta`` stays
    un-blurred (reference kernels.py:349-413)."""

    def __init__(self, K, b, svd_backend=None):
        KernelSVD.__init__(self)
        self._T = None
        self.kernel = K
        self._b = b
        self.svd_backend = K.svd_backend if svd_backend is None else svd_backend
        self._fill_values()

    def _device_svd(self):
        if not isinstance(self.kernel, TauKernel):
            raise NotImplementedError('device SVD of a PreblurKernel needs a TauKernel inside')
        U, S, V = self.kernel._device_svd(preblur_b=self._b)
        T = self.kernel._T
        return (U if T is None else np.dot(T, U)), S, V

    @classmethod
    def scan(cls, K, b_values, threshold=1.e-14):
        """The kernels of a b-scan (reference doc/guide/preblur_example.py:49-56) with
        their truncated SVDs from ONE batched device launch (``mxe_kernel_svd``)."""
        from . import device
        if not isinstance(K, TauKernel) or K._T is not None:
            raise NotImplementedError('PreblurKernel.scan needs an unrotated TauKernel')
        tau, w, delta, beta = K._device_args()
        res = device.kernel_svd(tau, w, delta, beta, list(b_values), threshold=threshold)
        out = []
        for b, r in zip(b_values, res):
            Kb = cls(K, b, svd_backend='device')
            Kb._U, Kb._S, Kb._V = r['U'], r['S'], r['V']
            Kb._last_threshold = threshold
            out.append(Kb)
        return out

    def parameter_change(self):
        self.kernel.parameter_change()
        self._fill_values()

    def _fill_values(self):
        self._invalidate_svd()
        self._B = get_preblur(self.omega, self._b)
        self._K = np.dot(self.kernel.K,
                         self._B * self.omega.delta[:, np.newaxis])
        self._K_delta = self.kernel.K_delta

    def transform(self, T):
        """rotate the blurred kernel like the plain one: ``U <- T U``, ``K <- T K``, S and V stay.  (The
        reference refills and decomposes T K' again for every rotation, kernels.py:395-397; the
        decomposition of the unrotated K' spans every rotated kernel's row space, so one V serves all
        matrix elements of a job and the SVD is done once.)"""
        rel = self._relative_rotation(T, self.kernel._T)
        if rel is None:
            return
        U = self.U
        self.kernel.transform(T)
        self._U = np.dot(rel, U)
        self._K = np.dot(rel, self._K)

    # ``_T`` of a PreblurKernel stays None in the reference (kernels.py:374: only the wrapped kernel's is
    # updated), and TauMaxEnt's bookkeeping of the DATA rotation reads it: with a preblur the data are
    # never rotated back before a new rotation.  The results of the reference depend on it
    # (tests/golden/elementwise_cov.npz), so it is kept; the rotation the matrix really carries is
    # ``rotation``.
    @property
    def rotation(self):
        return self.kernel.rotation

    @property
    def b(self):
        return self._b

    @b.setter
    def b(self, value):
        # (reference test/python/cov.py:158-159 assigns ``K.b`` and calls ``parameter_change()``; in the reference the
        #  assignment creates an attribute nobody reads and the refill blurs with the old width -- here it is the width)
        self._b = value

    @property
    def B(self):
        return self._B

    def get_omega(self):
        return self.kernel.omega

    def set_omega(self, omega):
        self.kernel.omega = omega

    omega = property(get_omega, set_omega)

    @property
    def data_variable(self):
        return self.kernel.data_variable

    @data_variable.setter
    def data_variable(self, value):
        self.kernel.data_variable = value
