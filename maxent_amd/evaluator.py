"""Device evaluation of the cost function and its ingredients at a given point.

``Evaluator`` owns one :class:`maxent_amd.device.DeviceContext` for a fixed
(kernel, data, error, default model, entropy) and turns ``mxe_eval_batch`` into
the quantities the reference's function objects return (reference
python/functions.py, python/cost_functions/):

======================  ====================================================
device output           reference quantity
======================  ====================================================
``Q, chi2, S``          ``Q.f``, ``chi2.f``, ``S.f``
``H``, ``u``, ``w``     ``H_of_v.f``; ``-S.d``; ``H_of_v.d = diag(w) V``, ``S.dd = -diag(1/w)``
``q = V g``             ``Q.dH`` (with alpha = 0, eta = 1: ``chi2.d / 2``)
``g``                   ``Q.d`` for ``BryanCostFunction`` and ``dA_projection = 1``
``W``, ``W2``           Gram matrices of dH/dv and of the d2H/dv2 term
======================  ====================================================

There is no CPU path here: without the library and a GPU the constructor raises.
"""

import numpy as np

from . import device


class Evaluator(object):
    def __init__(self, K, G, err, D, kind, device_id=0):
        K.S                                     # decompose if needed
        U, S, V = np.array(K.U), np.array(K.S), np.array(K.V)
        self.v_length = len(S)
        if len(S) > 128:
            # A kernel that has not been through reduce_singular_space: functions of H (chi2(H), S(H), their
            # derivatives) do not see the singular directions below 1e-14 of the largest -- they carry that
            # fraction of K H --, so the device gets the rest; functions of v need every component of v
            keep = S >= 1e-14 * S[0]
            if int(keep.sum()) > 128:
                raise device.MaxEntDeviceError(
                    'the device evaluates cost functions with at most 128 singular values; call '
                    'K.reduce_singular_space() first ({} kept now)'.format(len(S)))
            U, S, V = U[:, keep], S[keep], V[:, keep]
        G = np.asarray(G, dtype=float)
        err = np.asarray(err, dtype=float) * np.ones(len(G))
        self.ctx = device.DeviceContext(None if K.rotation is not None else U, S, V, device=device_id)
        ds = self.ctx.add_dataset(err, U if K.rotation is not None else None)
        self.ctx.set_elements([ds], [G], np.asarray(D, dtype=float)[np.newaxis, :], [kind])
        self.U, self.S, self.V, self.err = U, S, V, err
        self._M = None

    @property
    def M(self):
        """S U^T diag(1/err^2) U S: the chi2 curvature in the singular space (n_s x n_s, a constant of
        the data set)"""
        if self._M is None:
            C = self.U * self.S[np.newaxis, :] / self.err[:, np.newaxis]
            self._M = np.dot(C.T, C)
        return self._M

    def at_v(self, v, alpha, eta=1.0, want=('Q', 'chi2', 'S', 'H', 'u', 'w', 'q', 'g', 'W')):
        if len(self.S) != self.v_length:
            raise device.MaxEntDeviceError(
                'functions of v are evaluated with at most 128 singular values; call K.reduce_singular_space() '
                'first ({} kept now)'.format(self.v_length))
        out = self.ctx.eval_batch([0], [alpha], np.asarray(v, dtype=float)[np.newaxis, :],
                                  chi2_factor=eta, want=want)
        return {k: a[0] for k, a in out.items()}

    def at_H(self, H, want=('chi2', 'S', 'u', 'w', 'q', 'h')):
        out = self.ctx.eval_batch([0], [0.0], np.asarray(H, dtype=float)[np.newaxis, :],
                                  input_is_H=True, want=want)
        return {k: a[0] for k, a in out.items()}

    def close(self):
        self.ctx.close()
