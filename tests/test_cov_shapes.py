"""reference test/python/cov.py:96-180 restated: under a covariance rotation that keeps n_cov of the 201 data points,
the shapes of K (rotated), U, V and K_delta (never rotated) survive every parameter change -- omega mesh, tau grid,
truncation of the singular space, a preblur kernel and its width, new data -- whether the rotation is applied before
the change (first round) or again after it on a fresh kernel (second round).  Host logic only."""
import copy
import os

import numpy as np

import maxent_amd as mx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_rotated_kernel_shapes_survive_parameter_changes():
    z = np.load(os.path.join(GOLD, 'kat_tau_maxent.npz'))
    table = z['G_clean_file']                      # the reference's g_tau_semicircular.dat (201 x 2)
    tau, G = table[:, 0].copy(), table[:, 1].copy()
    rng = np.random.RandomState(298347923 % (2 ** 31))
    err = -1.e-3 * G
    G = G + err * rng.randn(len(err))
    tm = mx.TauMaxEnt(cost_function='bryan')
    tm.set_verbosity(mx.VerbosityFlags.Quiet)
    tm.set_G_tau_data(tau, G)
    tm.omega = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=100)
    tm.set_error(err)
    K_orig = copy.deepcopy(tm.K)
    n_cov = 9

    def check(N_om):
        assert tm.K.K.shape == (n_cov, N_om)
        assert tm.K.U.shape[0] == n_cov
        assert tm.K.V.shape[0] == N_om
        assert tm.K.K_delta.shape == (201, N_om)

    for i in range(2):
        N_om = 80
        err = np.array(err)
        err[n_cov:] = 0.0
        C = np.diag(err ** 2)

        def fresh():
            if i == 1:
                tm.err = None
                tm.K = copy.deepcopy(K_orig)

        def again():
            if i == 1:
                tm.set_cov(C)
        tm.err = None
        tm.K = copy.deepcopy(K_orig)
        tm.set_cov(C)
        check(100)
        fresh()
        tm.omega = mx.HyperbolicOmegaMesh(n_points=N_om)
        again()
        check(N_om)
        if i == 1:
            fresh()
            N_om = 100
        tm.tau = np.linspace(0, 50, 201)
        again()
        check(N_om)
        fresh()
        tm.K.reduce_singular_space(1.e-2)
        again()
        check(N_om)
        fresh()
        tm.K.reduce_singular_space(1.e-14)
        again()
        check(N_om)
        fresh()
        tm.K = mx.PreblurKernel(tm.K, 0.1)
        again()
        check(N_om)
        fresh()
        if hasattr(tm.K, 'b'):
            tm.K.b = 0.2
            tm.K.parameter_change()
        again()
        check(N_om)
        fresh()
        tm.G = G.copy()
        again()
        check(N_om)
