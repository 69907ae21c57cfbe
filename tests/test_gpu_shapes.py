"""Odd problem shapes through both kernel layouts: n_omega not a multiple of 16 / 64 / 128, few and
many singular values, short alpha scans, mixed entropies, pieces of one or two alphas.  The two
layouts must agree with each other and with the extended-precision fixed point."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import anchor                                                # noqa: E402
import maxent_amd as mx                                     # noqa: E402
from maxent_amd import device, synthetic, hostprep           # noqa: E402
from oracle import ref_numpy as R                            # noqa: E402

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


@pytest.mark.parametrize('n_orb,n_tau,n_omega,n_alpha,split', [
    (2, 50, 64, 7, 0),
    (3, 60, 100, 12, 3),
    (2, 120, 130, 30, 0),
    (3, 80, 257, 9, 9),          # one alpha per piece
    (2, 200, 640, 16, 4),
    (4, 40, 33, 5, 0),
])
def test_layouts_agree_on_odd_shapes(n_orb, n_tau, n_omega, n_alpha, split):
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_omega)
    K.reduce_singular_space(1e-14)
    assert len(K.S) <= 64
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    elems = [(i, j) for i in range(n_orb) for j in range(n_orb)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    n = len(elems)
    ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
    a = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(chains_per_wg=1, alpha_split=max(split, 1)))
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel<')
    b = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(chains_per_wg=4, alpha_split=split))
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel_mc<')
    for o in (a, b):
        assert o['converged'].all()
        assert np.all(np.isfinite(o['H']))
    assert rel_l2(b['H'], a['H']).max() < 1e-7
    np.testing.assert_allclose(b['chi2'], a['chi2'], rtol=1e-6)
    np.testing.assert_allclose(b['S'], a['S'], rtol=1e-6, atol=1e-10)
    # the fixed point itself, at the two ends of one diagonal and one off-diagonal scan
    for c in (0, 1):
        i, j = elems[c]
        ent = 'normal' if kinds[c] == device.ENTROPY_NORMAL else 'plusminus'
        p = R.Problem(np.array(K.K), K.U, K.S, K.V, Gmat[i, j], err, D, entropy=ent)
        truth, _ = anchor.truth_rows(p, omega.delta, alphas, n_tau, (0, n_alpha - 1), ent)     # from the reference's iterates
        for ia in (0, n_alpha - 1):
            assert np.linalg.norm(b['H'][c, ia] - truth[ia]) / np.linalg.norm(truth[ia]) < 1e-6
    ctx.close()


@pytest.mark.parametrize('n_omega,precision', [(3100, device.PRECISION_F64), (7000, device.PRECISION_F32)])
def test_frequency_mesh_beyond_the_lds(n_omega, precision):
    """A frequency mesh whose omega-space state does not fit the 160 KB of LDS: the one-chain kernel keeps
    the state in device memory.  binary64: against the extended-precision fixed point reached from the
    reference's iterates (the oracle runs the three largest alphas only: its dense n_omega x n_omega algebra
    takes a minute there) and the all-problem audit; binary32: against the binary64 run."""
    n_orb, n_tau, n_alpha = 2, 40, 6
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_omega)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    elems = [(i, j) for i in range(n_orb) for j in range(n_orb)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    n = len(elems)
    ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
    out = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(precision=device.PRECISION_F64))
    # (round 3: the lock-step layout itself keeps u, H, sw in device memory; until then the one-chain kernel took these)
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel_mc<32, 1, lead') and \
        'device-memory state' in ctx.last_launch_info()['kernel']
    one = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(chains_per_wg=1))
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel<') and rel_l2(out['H'], one['H']).max() < 1e-7
    out = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(precision=device.PRECISION_F64))
    assert out['converged'].all() and np.all(np.isfinite(out['H']))
    assert ctx.audit()['corr'].max() < 1e-6
    if precision == device.PRECISION_F64:
        for c in (0, 1):
            i, j = elems[c]
            ent = 'normal' if kinds[c] == device.ENTROPY_NORMAL else 'plusminus'
            p = R.Problem(np.array(K.K), K.U, K.S, K.V, Gmat[i, j], err, D, entropy=ent)
            truth, _ = anchor.truth_rows(p, omega.delta, alphas[:3], n_tau, (0, 2), ent)
            for ia in (0, 2):
                assert np.linalg.norm(out['H'][c, ia] - truth[ia]) / np.linalg.norm(truth[ia]) < 1e-6
    else:
        # (a binary32 request on a mesh whose basis does not fit the LDS is promoted to the binary64 lock-step kernel -- round 5 --;
        #  lds_basis = 2 keeps the one-chain binary32 kernel, here with its state in device memory)
        ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(precision=precision))
        assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel_mc<32, 1')
        o32 = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(precision=precision, lds_basis=2))
        assert 'float, device-memory state' in ctx.last_launch_info()['kernel']
        assert o32['converged'].all()
        assert rel_l2(o32['H'], out['H']).max() < 1e-3
    ctx.close()


def test_more_than_64_singular_values_on_a_long_mesh():
    """n_s > 64 (the 128 x 128 Newton matrix takes 132 KB of LDS) with a frequency mesh that no longer fits
    beside it: one wave per chain, state in device memory; checked by the all-problem audit and against the
    mesh cut to the size that does fit (same grid spacing is not needed: the fixed point of each run is
    audited on its own)."""
    n_tau, n_omega, n_alpha = 110, 700, 5
    tau, omega, K, Gmat, _ = synthetic.matrix_G(2, n_tau, n_omega)
    K.reduce_singular_space(1e-18)
    assert len(K.S) > 64
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    kinds = [device.ENTROPY_NORMAL, device.ENTROPY_PLUSMINUS]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    ctx.set_elements([ds] * 2, [Gmat[0, 0], Gmat[0, 1]], np.tile(D, (2, 1)), kinds)
    out = ctx.solve_chains(np.arange(2), alphas, v0)
    assert ctx.last_launch_info()['kernel'] == 'mxe::chain_kernel<1, 4, double, device-memory state>'
    assert out['converged'].all() and np.all(np.isfinite(out['H']))
    assert ctx.audit()['corr'].max() < 1e-6
    ctx.close()


def test_alphas_the_lock_step_layout_gives_up_on_are_finished_in_the_one_chain_layout():
    """Few data points, small alpha: the Newton matrix is so ill conditioned that the binary16 Gram products of
    the lock-step kernel stall the iteration (hundreds of iterations per alpha where the binary64 Gram matrix
    takes five).  The kernel gives such an alpha up after 32 iterations and mxe_chains_finish solves it again
    in the one-chain layout: every alpha converges, to the result of the one-chain layout."""
    n_tau, n_omega, n_alpha = 40, 257, 20
    tau, omega, K, Gmat, _ = synthetic.matrix_G(1, n_tau, n_omega, seed=275795323)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(n_tau)
    alphas = np.array(mx.LogAlphaMesh(alpha_min=0.011225438059035777, alpha_max=160.6748373405088, n_points=n_alpha)) * n_tau
    kinds = [device.ENTROPY_NORMAL]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    ctx.set_elements([ds], [Gmat[0, 0]], np.tile(D, (1, 1)), kinds)
    ref = ctx.solve_chains(np.arange(1), alphas, v0, device.default_opts(chains_per_wg=1, alpha_split=1))
    assert ref['converged'].all()
    ctx.upload_chains(np.arange(1), alphas, v0, device.default_opts(alpha_split=1))
    ctx.launch()
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel_mc<')
    ctx.sync()
    first = ctx.fetch(want_v=False, want_H=False)
    n_fin = ctx.finish()
    out = ctx.fetch()
    assert n_fin == int((first["converged"] == 0).sum()) and n_fin > 0
    assert out['converged'].all() and np.all(np.isfinite(out['H']))
    assert rel_l2(out['H'], ref['H']).max() < 1e-7
    assert np.all(out['n_iter'] >= first['n_iter'])
    assert ctx.audit()['corr'].max() < 1e-6
    assert ctx.finish() == 0                      # nothing left to do
    ctx.close()


def test_more_than_32_coupled_directions_stay_in_the_lock_step_layout():
    """Error bars of 1e-6: at the smallest alphas of the scan more than 32 singular directions couple.  Plus-minus scans
    run in the build with a 64-row active block; the pieces of the normal-entropy scans are cut there (their systems are
    too ill conditioned for the binary16 Gram products): the alphas behind the cut read NaN / not converged after the
    launch alone and are solved by mxe_chains_finish (one warm chain per scan).  Against the one-chain layout on the
    whole batch, with the audit."""
    n_orb, n_tau, n_omega, n_alpha, sigma = 3, 200, 500, 40, 1e-6
    tau, omega, K, _, A_mat = synthetic.matrix_G(n_orb, n_tau, n_omega)
    rng = np.random.RandomState(7)
    noise = sigma * rng.randn(n_orb, n_orb, n_tau)
    Gmat = np.einsum('tw,ijw->ijt', K.K_delta, A_mat) + 0.5 * (noise + noise.transpose(1, 0, 2))
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = sigma * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    elems = [(i, j) for i in range(n_orb) for j in range(n_orb)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    n = len(elems)
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
    a = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(chains_per_wg=1))
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel<')
    # the launch alone: lock-step layout, the tail of every scan left out
    ctx.upload_chains(np.arange(n, dtype=np.int32), alphas, v0, device.default_opts())
    ctx.launch()
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel_mc<64, 1')
    raw = ctx.fetch(want_v=False, want_H=True)
    left = ~raw['converged'].astype(bool)
    # (in ``left`` also: alphas the lock-step kernel tried and gave up on, and the rest of their pieces)
    out = left & (raw['n_iter'] == 0) & np.isnan(raw['chi2']) & np.all(np.isnan(raw['H']), axis=-1)
    assert 0 < out.sum() <= out.size // 3
    normal = np.array(kinds) == device.ENTROPY_NORMAL
    assert all(out[c, -1] for c in range(n) if normal[c])             # the tail of every normal-entropy scan
    assert not out[~normal].any() and not left[~normal].any()         # the plus-minus scans are done
    assert ctx.finish() == int(left.sum())
    b = ctx.fetch(want_v=True, want_H=True)
    # (the three smallest alphas of the last diagonal element take more than 1000 iterations in either layout)
    ok = b['converged'].astype(bool)
    assert np.array_equal(ok, a['converged'].astype(bool)) and (~ok).sum() <= 4
    assert np.all(b['n_iter'][left] > 0)
    assert rel_l2(b['H'], a['H'])[ok].max() < 1e-6
    np.testing.assert_allclose(b['chi2'][ok], a['chi2'][ok], rtol=1e-5)
    audit = ctx.audit()['corr']
    assert np.nanmax(audit[ok]) < 1e-6
    # the whole thing in one call
    c_ = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts())
    assert np.array_equal(c_['converged'].astype(bool), ok) and rel_l2(c_['H'], b['H'])[ok].max() < 1e-9
    ctx.close()


def test_off_diagonal_batch_with_64_coupled_directions_needs_no_finishing_pass():
    """plus-minus scans only, error bars of 1e-6: the 64-row build solves everything; against the one-chain layout"""
    n_orb, n_tau, n_omega, n_alpha, sigma = 3, 200, 500, 30, 1e-6
    tau, omega, K, _, A_mat = synthetic.matrix_G(n_orb, n_tau, n_omega)
    rng = np.random.RandomState(11)
    noise = sigma * rng.randn(n_orb, n_orb, n_tau)
    Gmat = np.einsum('tw,ijw->ijt', K.K_delta, A_mat) + 0.5 * (noise + noise.transpose(1, 0, 2))
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = sigma * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(n_alpha)) * n_tau
    elems = [(i, j) for i in range(n_orb) for j in range(n_orb) if i != j]
    kinds = [device.ENTROPY_PLUSMINUS] * len(elems)
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    n = len(elems)
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
    a = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(chains_per_wg=1))
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel<')
    ctx.upload_chains(np.arange(n, dtype=np.int32), alphas, v0, device.default_opts())
    ctx.launch()
    assert ctx.last_launch_info()['kernel'].startswith('mxe::chain_kernel_mc<64, 1')
    b = ctx.fetch(want_v=True, want_H=True)
    assert b['converged'].all() and a['converged'].all() and ctx.finish() == 0
    assert b['n_act'].max() > 32 if 'n_act' in b else True
    assert rel_l2(b['H'], a['H']).max() < 1e-7
    np.testing.assert_allclose(b['chi2'], a['chi2'], rtol=1e-6)
    assert np.nanmax(ctx.audit()['corr']) < 1e-8
    ctx.close()


def test_a_thousand_data_points_run_in_the_lock_step_kernel(monkeypatch):
    """1 000 imaginary times: 70-79 singular values above the reference's threshold, more than the 64 the lock-step kernels hold --
    until round 5 such a job ran in the one-chain kernel with its 128 x 128 Newton matrix (16 x 16 x 100 alphas: 18.8 ms against
    0.92).  The host layer now stages 64 directions where the others cannot be told from zero in the job
    (batch_solver.directions_to_keep): the lock-step kernel runs, the answers are those of all directions to the stopping
    tolerance and pass the gate against the extended-precision fixed point reached from the ORACLE's iterates (which keep every
    direction: reference kernels.py:53-122, maxent_loop.py:184); v comes back with zeros in the directions that were not staged."""
    n_tau, n_omega, n_alpha = 1000, 300, 12
    tau, omega, K, Gmat, _ = synthetic.matrix_G(2, n_tau, n_omega)

    def run():
        ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
        ew.set_verbosity(mx.VerbosityFlags.Quiet)
        ew.set_G_tau_data(tau, Gmat)
        ew.omega = omega
        ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e3, n_points=n_alpha)
        ew.set_error(synthetic.SIGMA)
        r = ew.run()
        return ew, np.array(r.A), np.array(r.H), np.array(r.v), np.array(r.chi2), ew.last_launches[-1]['kernel']
    ew, A, H, v, chi2, kernel = run()
    n_s = len(ew.maxent_diagonal.K.S)
    assert n_s > 64 and v.shape[-1] == n_s
    assert kernel.startswith('mxe::chain_kernel_mc<'), kernel
    assert np.all(v[..., 64:] == 0.0) and np.any(v[..., :64] != 0.0)
    monkeypatch.setenv('MAXENT_AMD_ALL_DIRECTIONS', '1')
    ew2, A2, H2, v2, chi22, kernel2 = run()
    assert kernel2.startswith('mxe::chain_kernel<'), kernel2
    assert np.abs(v2[..., 64:]).max() < 1e-8                    # (what the dropped directions would have carried)
    assert (np.linalg.norm(A - A2, axis=-1) / np.linalg.norm(A2, axis=-1)).max() < 1e-7
    np.testing.assert_allclose(chi2, chi22, rtol=1e-6)
    # against the truth reached from the reference's algorithm on ALL directions
    Kk = ew2.maxent_diagonal.K
    alphas = np.array(ew.alpha_mesh) * n_tau
    D = np.asarray(ew.maxent_diagonal.D.D)
    for (i, j), ent in (((0, 0), 'normal'), ((0, 1), 'plusminus')):
        p = R.Problem(np.array(Kk.K), Kk.U, Kk.S, Kk.V, Gmat[i, j], synthetic.SIGMA * np.ones(n_tau), D, entropy=ent)
        truth, _ = anchor.truth_rows(p, omega.delta, alphas, n_tau, (0, n_alpha - 1), ent)
        for ia in (0, n_alpha - 1):
            assert anchor.rel_l2_checked(H[i, j, ia], truth[ia]) < 1e-6, (i, j, ia)
