"""Building blocks of the cost function (host-side descriptors).

The reference composes Q_alpha(v) from doubly-derivable numpy objects
(reference python/functions.py).  On the MI355X the arithmetic of
chi2 / S / H(v) and all their derivatives is inside the chain kernel
(maxent_amd/csrc/mxe_kernel.hip.h); what remains on the host are the objects
that *hold the inputs* under the reference's names so that user scripts keep
working (``tm.chi2.K``, ``tm.S.D``, ``tm.A_of_H = PreblurA_of_H(...)``), plus
the cheap one-off maps the host needs around the solve:

* ``NormalChi2.f(H)``            -- header line "Minimal chi2" only
* ``H_of_v.f(v)`` / ``.inv(A)``  -- start vector v0 (maxent_loop.py:196-203)
* ``A_of_H.f(H)``                -- output map A = H/delta or A = B H

The entropy kind selects the kernel variant:
``NormalEntropy`` + ``NormalH_of_v`` -> MXE_ENTROPY_NORMAL,
``PlusMinusEntropy`` + ``PlusMinusH_of_v`` -> MXE_ENTROPY_PLUSMINUS
(reference functions.py:491-564, 720-796).
"""

import numpy as np

from .device import ENTROPY_NORMAL, ENTROPY_PLUSMINUS
from .hostprep import safelog
from .preblur import get_preblur


class GenericFunction(object):
    def parameter_change(self):
        pass


# ---------------------------------------------------------------- chi2 ----
class Chi2(GenericFunction):
    """holds K, G, err (reference functions.py:200-333)."""

    def __init__(self, K=None, G=None, err=None):
        self._K, self._G, self._err = K, G, err

    def get_K(self):
        return self._K

    def set_K(self, K, update_chi2=True):
        self._K = K

    K = property(get_K, set_K)

    def get_G(self):
        return self._G

    def set_G(self, G, update_chi2=True):
        self._G = G

    G = property(get_G, set_G)

    def get_err(self):
        return self._err

    def set_err(self, err, update_chi2=True):
        self._err = err

    err = property(get_err, set_err)

    def get_omega(self):
        return self.K.omega

    def set_omega(self, omega, update_K=True, update_chi2=True):
        self.K.omega = omega
        if update_K:
            self.K.parameter_change()

    omega = property(get_omega, set_omega)

    def get_data_variable(self):
        return self.K.data_variable

    def set_data_variable(self, data_variable, update_K=True,
                          update_chi2=True):
        self.K.data_variable = data_variable
        if update_K:
            self.K.parameter_change()

    data_variable = property(get_data_variable, set_data_variable)


class NormalChi2(Chi2):
    r""":math:`\chi^2 = \sum_i (G_i - \sum_j K_{ij} H_j)^2/\sigma_i^2`
    (reference functions.py:336-377)."""

    def f(self, H):
        r = np.dot(self.K.K, H) - self.G
        return float(np.sum(np.abs(r) ** 2 / self.err ** 2))


# ------------------------------------------------------------- entropy ----
class Entropy(GenericFunction):
    kind = None

    def __init__(self, D=None):
        self._D = D

    def get_D(self):
        return self._D

    def set_D(self, D, update_S=True):
        self._D = D

    D = property(get_D, set_D)

    def get_omega(self):
        return self.D.omega

    def set_omega(self, omega, update_D=True, update_S=True):
        self.D.omega = omega
        if update_D:
            self.D.parameter_change()

    omega = property(get_omega, set_omega)


class NormalEntropy(Entropy):
    r""":math:`S = \sum_i (H_i - D_i - H_i \log(H_i/D_i))`
    (reference functions.py:491-520)."""
    kind = ENTROPY_NORMAL


class PlusMinusEntropy(Entropy):
    r""":math:`S = S_n(H^+) + S_n(H^-)`, :math:`H = H^+ - H^-`
    (reference functions.py:523-564)."""
    kind = ENTROPY_PLUSMINUS


# ---------------------------------------------------------------- H(v) ----
class GenericH_of_v(GenericFunction):
    kind = None

    def __init__(self, D=None, K=None):
        self._D, self._K = D, K

    def get_D(self):
        return self._D

    def set_D(self, D, update_H_of_v=True):
        self._D = D

    D = property(get_D, set_D)

    def get_K(self):
        return self._K

    def set_K(self, K, update_H_of_v=True):
        self._K = K

    K = property(get_K, set_K)

    def get_omega(self):
        return self.D.omega

    def set_omega(self, omega, update_D=True, update_H_of_v=True):
        self.D.omega = omega
        if update_D:
            self.D.parameter_change()

    omega = property(get_omega, set_omega)


class NormalH_of_v(GenericH_of_v):
    r"""Bryan's parametrisation :math:`H = D e^{Vv}`
    (reference functions.py:720-755)."""
    kind = ENTROPY_NORMAL

    def f(self, v):
        return self.D.D * np.exp(np.dot(self.K.V, v))

    def inv(self, A):
        return np.dot(self.K.V.transpose(), safelog(A / self.D.D))


class PlusMinusH_of_v(GenericH_of_v):
    r""":math:`H = D (e^{Vv} - e^{-Vv})` (reference functions.py:758-796)."""
    kind = ENTROPY_PLUSMINUS

    def f(self, v):
        u = np.dot(self.K.V, v)
        return self.D.D * (np.exp(u) - np.exp(-u))

    def inv(self, A):
        D = self.D.D
        return np.dot(self.K.V.transpose(),
                      safelog((A + np.sqrt(A ** 2 + 4 * D ** 2)) / (2 * D)))


# ---------------------------------------------------------------- A(H) ----
class GenericA_of_H(GenericFunction):
    def get_omega(self):
        return self._omega

    def set_omega(self, omega, update_A_of_H=True):
        self._omega = omega
        if update_A_of_H:
            self.parameter_change()

    omega = property(get_omega, set_omega)


class IdentityA_of_H(GenericA_of_H):
    """A = H / delta_omega (reference functions.py:937-964)."""

    def __init__(self, omega):
        self._omega = omega

    def f(self, H):
        return np.asarray(H) / self._omega.delta

    def inv(self, A):
        return np.asarray(A) * self._omega.delta

    def matrix(self):
        return None


class PreblurA_of_H(GenericA_of_H):
    """A = B H with the blur matrix of width ``b``
    (reference functions.py:967-1023); pair with ``PreblurKernel``."""

    def __init__(self, b, omega):
        self._omega = omega
        self._b = b
        self.parameter_change()

    def parameter_change(self):
        self._B = get_preblur(self._omega, self._b)

    def f(self, H):
        return np.dot(np.asarray(H), self._B.T)

    def inv(self, A):
        return np.linalg.lstsq(self._B, A, rcond=None)[0]

    def matrix(self):
        return self._B

    def get_b(self):
        return self._b

    def set_b(self, b, update_A_of_H=True):
        self._b = b
        if update_A_of_H:
            self.parameter_change()

    b = property(get_b, set_b)
