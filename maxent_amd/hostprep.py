"""Host-side preparation shared by MaxEntLoop, ElementwiseMaxEnt and bench.py."""

import numpy as np

from .device import ENTROPY_NORMAL, ENTROPY_PLUSMINUS


def safelog(A):
    """log with |x| <= 1e-100 clamped (reference functions.py:53-56)."""
    A = np.array(A, dtype=float)
    A[np.abs(A) <= 1.e-100] = 1.e-100
    return np.log(A)


def initial_v(V, D, delta, kind, A_init=None):
    """Start vector of the first alpha (reference maxent_loop.py:196-203):
    ``v0 = H_of_v.inv((D or A_init) * delta)``.  Note that ``D`` already
    contains delta; the reference multiplies by it once more and so do we.
    ``H_of_v.inv``: functions.py:753-755 (normal), 793-796 (plusminus)."""
    D = np.asarray(D, dtype=float)
    A = (D if A_init is None else np.asarray(A_init, dtype=float)) * delta
    if kind == ENTROPY_NORMAL:
        return np.dot(V.T, safelog(A / D))
    if kind == ENTROPY_PLUSMINUS:
        return np.dot(V.T, safelog((A + np.sqrt(A ** 2 + 4 * D ** 2)) /
                                   (2 * D)))
    raise ValueError('unknown entropy kind')
