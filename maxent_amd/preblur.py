"""Blur matrix of the preblur formalism (reference python/preblur.py:31-58)."""

import numpy as np


def get_preblur(omega, b):
    """Gaussian of width ``b`` on the omega mesh, normalised first along the
    rows, then along the columns, with the trapezoid weights."""
    w = np.asarray(omega, dtype=float)
    delta = omega.delta
    diff = w[np.newaxis, :] - w[:, np.newaxis]      # [i, j] = w_j - w_i
    B = np.exp(-diff ** 2 / 2.0 / b ** 2) / np.sqrt(2.0 * np.pi * b ** 2)
    B = B / np.dot(delta, B)[:, np.newaxis]
    B = B / np.dot(B, delta)[np.newaxis, :]
    return B
