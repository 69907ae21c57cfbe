"""Cost function front-end objects (reference python/cost_functions/).

``Q_alpha(v) = 1/2 chi2(H(v)) eta - alpha S(H(v))``.  The objects keep the
reference's attribute plumbing (``K, G, err, omega, D, data_variable, chi2, S,
H_of_v, A_of_H`` with the same setter side effects: changing omega or tau
refills the kernel, reference cost_function.py:107-245) and tell the device
layer which kernel variant to run.  ``MaxEntCostFunction`` and
``BryanCostFunction`` (reference maxent_cost_function.py:26-165,
bryan_cost_function.py:26-144) differ in the reference only in how the Newton
system is written (W M W + alpha W vs. M W); both have the same minimiser and
the device solves Bryan's form of it, so they select the same kernel.
"""

from .functions import (NormalChi2, NormalEntropy, NormalH_of_v,
                        IdentityA_of_H)


class CostFunction(object):
    def __init__(self, chi2=None, S=None, H_of_v=None, A_of_H=None,
                 chi2_factor=1.0):
        self._chi2 = chi2 if chi2 is not None else NormalChi2()
        self._S = S if S is not None else NormalEntropy()
        self._H_of_v = H_of_v if H_of_v is not None else NormalH_of_v()
        if A_of_H is None:
            omega = None
            try:
                omega = self._chi2.omega
            except Exception:
                pass
            A_of_H = IdentityA_of_H(omega)
        self._A_of_H = A_of_H
        if chi2_factor != 1.0:
            raise NotImplementedError('chi2_factor != 1 is not supported by '
                                      'the device solver')
        self.chi2_factor = chi2_factor
        self._alpha = None

    def set_alpha(self, alpha):
        self._alpha = alpha

    @property
    def entropy_kind(self):
        if self._S.kind != self._H_of_v.kind:
            raise Exception('S and H_of_v do not belong together: use '
                            'NormalEntropy with NormalH_of_v or '
                            'PlusMinusEntropy with PlusMinusH_of_v')
        return self._S.kind

    def parameter_change(self):
        pass

    # ---- K ----
    def get_K(self):
        return self.chi2.K

    def set_K(self, K, update_chi2=True, update_H_of_v=True, update_Q=True):
        self.chi2.set_K(K, update_chi2=update_chi2)
        self.H_of_v.set_K(K, update_H_of_v=update_H_of_v)

    K = property(get_K, set_K)

    # ---- G ----
    def get_G(self):
        return self.chi2.G

    def set_G(self, G, update_chi2=True, update_Q=True):
        self.chi2.set_G(G, update_chi2=update_chi2)

    G = property(get_G, set_G)

    # ---- err ----
    def get_err(self):
        return self.chi2.err

    def set_err(self, err, update_chi2=True, update_Q=True):
        self.chi2.set_err(err, update_chi2=update_chi2)

    err = property(get_err, set_err)

    # ---- omega ----
    def get_omega(self):
        return self.chi2.K.omega

    def set_omega(self, omega, update_K=True, update_chi2=True, update_D=True,
                  update_S=True, update_H_of_v=True, update_A_of_H=True,
                  update_Q=True):
        self.chi2.set_omega(omega, update_K=update_K, update_chi2=update_chi2)
        if update_K:
            self.H_of_v.set_K(self.K, update_H_of_v=False)
        self.S.set_omega(omega, update_D=update_D, update_S=update_S)
        self.H_of_v.set_omega(omega, update_D=update_D,
                              update_H_of_v=update_H_of_v)
        self.A_of_H.set_omega(omega, update_A_of_H=update_A_of_H)

    omega = property(get_omega, set_omega)

    # ---- data variable (tau) ----
    def get_data_variable(self):
        return self.chi2.K.data_variable

    def set_data_variable(self, data_variable, update_K=True,
                          update_chi2=True, update_Q=True, update_H_of_v=True):
        self.chi2.set_data_variable(data_variable, update_K=update_K,
                                    update_chi2=update_chi2)
        if update_K:
            self.H_of_v.set_K(self.K, update_H_of_v=update_H_of_v)

    data_variable = property(get_data_variable, set_data_variable)

    # ---- D ----
    def get_D(self):
        return self.S.D

    def set_D(self, D, update_S=True, update_H_of_v=True, update_Q=True,
              update_A_of_H=True):
        self.S.set_D(D, update_S=update_S)
        self.H_of_v.set_D(D, update_H_of_v=update_H_of_v)
        self.A_of_H.set_omega(D.omega, update_A_of_H=update_A_of_H)

    D = property(get_D, set_D)

    # ---- components ----
    def get_chi2(self):
        return self._chi2

    def set_chi2(self, chi2, update_Q=True):
        self._chi2 = chi2

    chi2 = property(get_chi2, set_chi2)

    def get_S(self):
        return self._S

    def set_S(self, S, update_Q=True):
        self._S = S

    S = property(get_S, set_S)

    def get_H_of_v(self):
        return self._H_of_v

    def set_H_of_v(self, H_of_v, update_Q=True):
        self._H_of_v = H_of_v

    H_of_v = property(get_H_of_v, set_H_of_v)

    def get_A_of_H(self):
        return self._A_of_H

    def set_A_of_H(self, A_of_H, update_Q=True):
        self._A_of_H = A_of_H

    A_of_H = property(get_A_of_H, set_A_of_H)

    @property
    def G_orig(self):
        return getattr(self, '_G_orig', self.G)


class MaxEntCostFunction(CostFunction):
    """the general cost function (reference default; ``d_dv`` and
    ``dA_projection`` only change how the reference writes its Newton system
    and are accepted for compatibility)."""

    def __init__(self, d_dv=False, dA_projection=2, **kwargs):
        self.d_dv = d_dv
        self.dA_projection = dA_projection
        super(MaxEntCostFunction, self).__init__(**kwargs)


class BryanCostFunction(CostFunction):
    """Bryan's singular-space form; normal entropy only."""

    def __init__(self, chi2_factor=1.0):
        super(BryanCostFunction, self).__init__(chi2_factor=chi2_factor)

    def set_H_of_v(self, H_of_v, update_Q=True):
        raise NotImplementedError('Cannot change H_of_v in BryanCostFunction.')

    H_of_v = property(CostFunction.get_H_of_v, set_H_of_v)

    def set_A_of_H(self, A_of_H, update_Q=True):
        raise NotImplementedError('Cannot change A_of_H in BryanCostFunction.')

    A_of_H = property(CostFunction.get_A_of_H, set_A_of_H)
