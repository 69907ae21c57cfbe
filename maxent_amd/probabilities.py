"""log p(alpha | G) in singular space (reference python/probabilities.py:26-85).

The reference evaluates, per alpha, two n_omega x n_omega ``slogdet``:

    log p = -1/2 logdet(d2Q/dH2) + 1/2 logdet(-d2S/dH2)
            + (N_omega/2) log(alpha) - Q - log(alpha)

with d2Q/dH2 = K^T diag(1/err^2) K + alpha diag(1/w) and -d2S/dH2 = diag(1/w)
(w = H for the normal entropy, H+ + H- for plusminus).  With
K = U S V^T and Sylvester's identity the two determinants and the
normalisation collapse to an n_s x n_s one:

    log p = -1/2 logdet(I + M W / alpha) - Q - log(alpha),
    M = S U^T diag(1/err^2) U S,   W = V^T diag(w) V

(custom ``log_norm_S`` / ``log_prior_alpha`` are honoured; a custom
``log_measure`` is not supported).  Known answer: reference
test/python/tau_maxent.py:134-135.
"""

import numpy as np


class NormalLogProbability(object):
    def __init__(self, log_measure=None, log_norm_S=None,
                 log_prior_alpha=None):
        if log_measure is not None:
            raise NotImplementedError('custom log_measure is not supported')
        self.log_norm_S = log_norm_S
        self.log_prior_alpha = log_prior_alpha

    def from_logdet(self, logdet, alpha, Q, n_omega):
        """log p from ``logdet`` = log det(I + M W/alpha) (``mxe_logdet``)."""
        alpha = np.asarray(alpha, dtype=float)
        lp = -0.5 * np.asarray(logdet, dtype=float) - np.asarray(Q, dtype=float)
        if self.log_norm_S is not None:
            lp = lp + np.array([self.log_norm_S(a, n_omega) - (n_omega / 2.0) * np.log(a)
                                for a in alpha])
        if self.log_prior_alpha is None:
            lp = lp - np.log(alpha)
        else:
            lp = lp + np.array([self.log_prior_alpha(a) for a in alpha])
        return lp

    def evaluate(self, U, S, V, err, alpha, w, Q):
        """vectorised over alpha: ``alpha`` (X,), ``w`` (X, n_omega), ``Q`` (X,)."""
        C = (U * S[np.newaxis, :]) / np.asarray(err)[:, np.newaxis]
        M = np.dot(C.T, C)
        n_omega = V.shape[0]
        out = np.empty(len(alpha))
        for i, a in enumerate(alpha):
            W = np.dot(V.T * w[i][np.newaxis, :], V)
            _, ld = np.linalg.slogdet(np.eye(len(S)) + np.dot(M, W) / a)
            lp = -0.5 * ld - Q[i]
            if self.log_norm_S is not None:
                # default norm (N/2) log(alpha) is already folded in
                lp += self.log_norm_S(a, n_omega) - (n_omega / 2.0) * np.log(a)
            lp += (-np.log(a) if self.log_prior_alpha is None
                   else self.log_prior_alpha(a))
            out[i] = lp
        return out
