"""GPU parity of the chain kernel (through the C-ABI) against the oracle."""
import functools
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import anchor                                                # noqa: E402
from maxent_amd import device, synthetic                     # noqa: E402
from oracle import ref_numpy as R, sform as SF, hp_truth     # noqa: E402,F401

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


def _setup(n_tau, n_omega, entropy='normal', err=None):
    tau, omega, K, G = synthetic.single_G(n_tau, n_omega)
    K.reduce_singular_space(1e-14)
    U, S, V = K.U, K.S, K.V
    if err is None:
        err = synthetic.SIGMA * np.ones(n_tau)
    D = synthetic.flat_D(omega)
    p = R.Problem(np.array(K.K), U, S, V, G, err, D, entropy=entropy)
    return tau, omega, K, G, err, D, p


@functools.lru_cache(maxsize=None)
def _truth(n_tau, n_omega, n_alpha, entropy):
    tau, omega, K, G, err, D, p = _setup(n_tau, n_omega, entropy)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    return anchor.truth_rows(p, omega.delta, alphas, n_tau, (0, n_alpha // 2, n_alpha - 1), entropy)


@pytest.mark.parametrize('n_tau,n_omega,n_alpha,entropy,nw,layout,split', [
    (100, 200, 20, 'normal', 0, 0, 0),
    (200, 500, 100, 'normal', 4, 1, 1),
    (200, 500, 100, 'normal', 1, 1, 1),
    (200, 500, 100, 'plusminus', 2, 1, 1),
    (200, 500, 100, 'normal', 0, 4, 1),       # four-chains-per-workgroup kernel, 1 chain (3 empty slots)
    (200, 500, 100, 'plusminus', 0, 4, 5),    # ... with the alpha scan cut into 5 cold-started pieces
    (200, 500, 100, 'normal', 8, 1, 8),
    (200, 500, 100, 'plusminus', 8, 4, 3),    # lock-step kernel (waves_per_chain does not apply to it)
    (100, 200, 20, 'normal', 0, 4, 2),        # lock-step kernel, n_omega_pad = 256
])
def test_chain_matches_kernel_model_and_truth(n_tau, n_omega, n_alpha, entropy, nw, layout, split):
    tau, omega, K, G, err, D, p = _setup(n_tau, n_omega, entropy)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    v0 = R.initial_v(p, omega.delta)
    # numpy model of the kernel (same control flow)
    basis = SF.Basis(p.U, p.S, p.V, p.err)
    el = SF.Element(basis, G, D, entropy)
    ctx = device.DeviceContext(p.U, p.S, p.V)
    ds = ctx.add_dataset(err)
    ctx.set_elements([ds], [G], D[np.newaxis, :],
                     [device.ENTROPY_PLUSMINUS if entropy == 'plusminus' else device.ENTROPY_NORMAL])
    out = ctx.solve_chains([0], alphas, v0[np.newaxis, :],
                           device.default_opts(waves_per_chain=nw, chains_per_wg=layout,
                                               alpha_split=split))
    assert out['converged'].all()
    H = out['H'][0]
    assert np.all(np.isfinite(H))
    # extended precision truth at a few alphas, reached from the iterates of the reference's algorithm
    truth, _ = _truth(n_tau, n_omega, n_alpha, entropy)
    for ia in (0, n_alpha // 2, n_alpha - 1):
        e = np.linalg.norm(H[ia] - truth[ia]) / np.linalg.norm(truth[ia])
        assert e < 1e-6, (ia, e)
    # chi2/S/Q consistent with the reference's H-form evaluation at the returned v
    for ia in (0, n_alpha - 1):
        v = out['v'][0, ia]
        Hr = R.H_of_v(p, v)
        assert np.linalg.norm(Hr - H[ia]) / np.linalg.norm(Hr) < 1e-10
        assert abs(R.chi2_f(p, Hr) - out['chi2'][0, ia]) / out['chi2'][0, ia] < 1e-8
        assert abs(R.S_f(p, Hr) - out['S'][0, ia]) < 1e-8 * max(1.0, abs(out['S'][0, ia]))
    print('n_iter total', out['n_iter'].sum(), 'evals', out['n_evals'].sum(),
          'kernel ms', ctx.last_kernel_ms(), ctx.last_launch_info())
    ctx.close()


def test_four_chain_kernel_equals_single_chain_kernel():
    """a 3x3 matrix problem (mixed normal / plusminus chains in one workgroup)
    through both layouts; per-alpha results agree to the convergence level."""
    tau, omega, K, Gmat, _ = synthetic.matrix_G(3, 120, 300)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(120)
    alphas = np.array(synthetic.alpha_mesh(40)) * 120
    elems = [(i, j) for i in range(3) for j in range(3)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    from maxent_amd import hostprep
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    ctx.set_elements([ds] * 9, [Gmat[i, j] for i, j in elems], np.tile(D, (9, 1)), kinds)
    a = ctx.solve_chains(np.arange(9), alphas, v0, device.default_opts(chains_per_wg=1, alpha_split=1))
    b = ctx.solve_chains(np.arange(9), alphas, v0, device.default_opts(chains_per_wg=4, alpha_split=1))
    c = ctx.solve_chains(np.arange(9), alphas, v0, device.default_opts(chains_per_wg=4, alpha_split=4))
    for o in (a, b, c):
        assert o['converged'].all()
    assert ctx.last_launch_info()['n_workgroups'] == 9      # 36 pieces, four per workgroup
    for o in (b, c):
        assert rel_l2(o['H'], a['H']).max() < 1e-8
        np.testing.assert_allclose(o['chi2'], a['chi2'], rtol=1e-7)
        np.testing.assert_allclose(o['Q'], a['Q'], rtol=1e-10)
    # same Newton iteration in both kernels up to the precision of the Gram matrix (binary32 tiles
    # in the lock-step kernel) and the order of the sums: iteration counts differ by one at most
    d = b['n_iter'].astype(int) - a['n_iter'].astype(int)
    assert d.min() >= -1 and d.max() <= 1 and abs(d.mean()) < 0.1
    ctx.close()


def test_logdet_kernel_matches_numpy_slogdet():
    """mxe_logdet = log det(I + M W/alpha) over all kept singular directions, for both
    entropies and a tau-dependent error (rotated whitened basis on the device)."""
    tau, omega, K, Gmat, _ = synthetic.matrix_G(2, 120, 300)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    rng = np.random.RandomState(5)
    err = synthetic.SIGMA * (1.0 + rng.rand(120))
    alphas = np.array(synthetic.alpha_mesh(12)) * 120
    elems = [(0, 0), (0, 1)]
    kinds = [device.ENTROPY_NORMAL, device.ENTROPY_PLUSMINUS]
    from maxent_amd import hostprep
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    ctx.set_elements([ds] * 2, [Gmat[i, j] for i, j in elems], np.tile(D, (2, 1)), kinds)
    out = ctx.solve_chains(np.arange(2), alphas, v0)
    ld = ctx.logdet()
    ctx.close()
    assert out['converged'].all()
    C = (K.U * K.S[None, :]) / err[:, None]
    M = C.T @ C
    for c in range(2):
        for ia, a in enumerate(alphas):
            H = out['H'][c, ia]
            w = H if kinds[c] == device.ENTROPY_NORMAL else np.sqrt(H * H + 4.0 * D * D)
            W = (K.V.T * w[None, :]) @ K.V
            sign, ref = np.linalg.slogdet(np.eye(len(K.S)) + M @ W / a)
            assert sign > 0
            assert abs(ld[c, ia] - ref) < 1e-8 * max(1.0, abs(ref)), (c, ia, ld[c, ia], ref)
