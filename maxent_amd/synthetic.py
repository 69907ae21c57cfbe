"""Seeded synthetic inputs of the BASELINE.json configurations.

SURVEY.md section 8(d): beta = 40, tau = linspace(0, beta, n_tau), hyperbolic
omega mesh on [-10, 10], flat default model, alpha = LogAlphaMesh(1e-2, 1e4,
n_alpha) (scaled by n_tau like the reference does, maxent_loop.py:216-232),
noise 1e-4 * randn, error bar 1e-4.
"""

import numpy as np

from .omega_meshes import HyperbolicOmegaMesh
from .alpha_meshes import LogAlphaMesh
from .default_models import FlatDefaultModel
from .kernels import TauKernel

BETA = 40.0
SIGMA = 1.e-4


def grids(n_tau, n_omega):
    tau = np.linspace(0, BETA, n_tau)
    omega = HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=n_omega)
    return tau, omega


def two_gaussian_spectrum(omega):
    """cfg1/cfg2: 0.6 N(1, 0.5^2) + 0.4 N(-1.5, 0.8^2), trapezoid-normalised."""
    w = np.asarray(omega)
    A = 0.6 * np.exp(-(w - 1.0) ** 2 / (2 * 0.5 ** 2)) + \
        0.4 * np.exp(-(w + 1.5) ** 2 / (2 * 0.8 ** 2))
    return A / np.trapezoid(A, w)


def single_G(n_tau, n_omega, seed=1234):
    """cfg1 (100, 200) / cfg2 (200, 500): one scalar G(tau)."""
    tau, omega = grids(n_tau, n_omega)
    K = TauKernel(tau=tau, omega=omega, beta=BETA)
    rng = np.random.RandomState(seed)
    G = np.dot(K.K_delta, two_gaussian_spectrum(omega)) + \
        SIGMA * rng.randn(n_tau)
    return tau, omega, K, G


def matrix_G(n_orb, n_tau, n_omega, seed=2024, noise_seed=None):
    """cfg3/4/5: n_orb x n_orb G(tau) from Gaussian spectra rotated by a fixed
    random orthogonal matrix (all elements non-zero), symmetrised noise."""
    tau, omega = grids(n_tau, n_omega)
    K = TauKernel(tau=tau, omega=omega, beta=BETA)
    w = np.asarray(omega)
    mu = np.linspace(-1.5, 1.5, n_orb)
    s = np.linspace(0.4, 0.7, n_orb)
    A_diag = np.exp(-(w[np.newaxis, :] - mu[:, np.newaxis]) ** 2 /
                    (2 * s[:, np.newaxis] ** 2))
    A_diag /= np.trapezoid(A_diag, w, axis=1)[:, np.newaxis]
    rng = np.random.RandomState(seed)
    R, _ = np.linalg.qr(rng.randn(n_orb, n_orb))
    A_mat = np.einsum('ik,kw,jk->ijw', R, A_diag, R)
    G = np.einsum('tw,ijw->ijt', K.K_delta, A_mat)
    rng2 = np.random.RandomState(seed + 1 if noise_seed is None else noise_seed)
    noise = SIGMA * rng2.randn(n_orb, n_orb, n_tau)
    noise = 0.5 * (noise + noise.transpose(1, 0, 2))
    return tau, omega, K, G + noise, A_mat


def alpha_mesh(n_alpha):
    return LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=n_alpha)


def flat_D(omega):
    return FlatDefaultModel(omega).D
