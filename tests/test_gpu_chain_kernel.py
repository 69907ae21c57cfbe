"""GPU parity of the chain kernel (through the C-ABI) against the oracle."""
import numpy as np
import pytest

from maxent_amd import device, synthetic
from oracle import ref_numpy as R, sform as SF, hp_truth

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


@pytest.mark.parametrize('n_tau,n_omega,n_alpha,entropy,nw', [
    (100, 200, 20, 'normal', 0),
    (200, 500, 100, 'normal', 4),
    (200, 500, 100, 'normal', 1),
    (200, 500, 100, 'plusminus', 2),
])
def test_chain_matches_kernel_model_and_truth(n_tau, n_omega, n_alpha, entropy, nw):
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
                           device.default_opts(waves_per_chain=nw))
    assert out['converged'].all()
    H = out['H'][0]
    assert np.all(np.isfinite(H))
    # extended precision truth at a few alphas
    for ia in (0, n_alpha // 2, n_alpha - 1):
        vt, Ht = hp_truth.polish(p.K, G, err, D, p.V, p.S, alphas[ia], out['v'][0, ia], entropy, iters=4)
        e = np.linalg.norm(H[ia] - Ht) / np.linalg.norm(Ht)
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
