"""Several ranks of one process through the C entry points of the multi-GPU path.  The test
box has one GPU: two contexts on device 0 stand for two devices -- same sharding, staging, launches and
gather call (``mxe_gather_local``), with device copies where distinct devices would use RCCL send / recv.
N > 1 distinct devices are not available to the tests and remain unmeasured."""
import numpy as np
import pytest

import maxent_amd as mx
from maxent_amd import device, synthetic
from maxent_amd.batch_solver import BatchSolver, LazyH

pytestmark = pytest.mark.gpu


def job(n_orb=3, n_tau=60, n_omega=120, n_alpha=12, **kw):
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_omega)
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False, **kw)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(tau, Gmat)
    ew.omega = omega
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-1, alpha_max=1e3, n_points=n_alpha)
    ew.set_error(synthetic.SIGMA)
    return ew


def test_two_ranks_on_one_device_equal_one_rank():
    one = job().run()
    two_job = job(device_ids=(0, 0))
    two = two_job.run()
    assert len(two_job.last_launches[-1]['devices']) == 2
    for name in ('chi2', 'S', 'Q', 'H', 'A', 'A_out', 'v'):
        a, b = np.asarray(getattr(one, name)), np.asarray(getattr(two, name))
        assert a.shape == b.shape
        # the pieces of the alpha scans are cut per launch: answers agree to the solver's tolerance
        assert np.max(np.abs(a - b)) <= 1e-7 * np.max(np.abs(a)), name
    for i in range(3):
        for j in range(3):
            for an in ('LineFitAnalyzer', 'Chi2CurvatureAnalyzer'):
                assert one.analyzer_results[i][j][an]['alpha_index'] == two.analyzer_results[i][j][an]['alpha_index']


def test_gather_brings_every_ranks_pack_and_the_device_linefit_agrees_with_the_host():
    tau, omega, K, Gmat, _ = synthetic.matrix_G(3, 60, 120)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    from maxent_amd import hostprep
    alphas = np.array(mx.LogAlphaMesh(alpha_min=1e-1, alpha_max=1e3, n_points=12)) * 60
    specs = []
    for i in range(3):
        for j in range(3):
            kind = device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS
            specs.append(dict(G=Gmat[i, j], err=synthetic.SIGMA * np.ones(60), U_rot=None, D=D, kind=kind,
                              v0=hostprep.initial_v(K.V, D, omega.delta, kind), alpha=alphas))
    opts = mx.LevenbergMinimizer().to_opts()
    solo = BatchSolver(K, (0,))
    ref, _ = solo.solve(K, specs, opts, want_H=True)
    trio = BatchSolver(K, (0, 0, 0))
    got, info = trio.solve(K, specs, opts)
    assert info['devices'] == [0, 0, 0]
    from maxent_amd.analyzers import fit_piecewise
    assert all(isinstance(b['H'], LazyH) and not b['H'].on_host for b in got)      # nothing of H has moved yet
    for e, (a, b) in enumerate(zip(ref, got)):
        np.testing.assert_allclose(b['chi2'], a['chi2'], rtol=1e-7)
        idx, _ = fit_piecewise(np.log(alphas), np.log(b['chi2']))
        assert b['device_linefit_index'] == idx, e
        # the row the gather brought is the H of that alpha; a single-row fetch and the full fetch agree
        row = b['H'][idx]
        assert np.array_equal(row, b['device_linefit_H'])
        if e >= 6:           # the last element of every rank: now the full fetch (of the rank's whole shard)
            assert np.array_equal(np.asarray(b['H'])[idx], row) and b['H'].on_host
        assert np.max(np.abs(np.asarray(b['H']) - np.asarray(a['H']))) <= 1e-7 * np.max(np.abs(np.asarray(a['H'])))
    solo.close()
    trio.close()


def test_contexts_are_kept_between_runs_and_dropped_when_the_kernel_changes():
    ew = job()
    ew.run()
    K = ew.maxent_diagonal.K
    first = K._batch_solvers[(0,)]
    ew.maxent_result = None
    ew.run()
    assert K._batch_solvers[(0,)] is first
    ew.omega = mx.HyperbolicOmegaMesh(-8, 8, 100)      # refills the kernel: new decomposition, new context
    ew.maxent_result = None
    res = ew.run()
    K = ew.maxent_diagonal.K
    assert K._batch_solvers[(0,)] is not first and res.A.shape[-1] == 100


def test_process_ranks_comm_with_one_rank():
    """the entry points a one-process-per-GPU launcher uses (bench.py --gpus N), with the single rank this box has"""
    tau, omega, K, G = synthetic.single_G(40, 80)
    K.reduce_singular_space(1e-14)
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(synthetic.SIGMA * np.ones(40))
    D = synthetic.flat_D(omega)
    ctx.set_elements([ds], [G], D[np.newaxis, :], [device.ENTROPY_NORMAL])
    from maxent_amd import hostprep
    alphas = np.array([100.0, 10.0, 1.0, 0.5, 0.2, 0.1]) * 40
    ctx.upload_chains([0], alphas, hostprep.initial_v(K.V, D, omega.delta, device.ENTROPY_NORMAL)[np.newaxis, :])
    ctx.comm_init(1, 0, device.comm_unique_id())
    ctx.launch()
    ctx.select_launch(0)
    recv = np.empty(ctx.compact_count())
    ctx.gather(0, [ctx.compact_count()], recv=recv)
    out = ctx.fetch()
    assert np.array_equal(recv[:6], out['chi2'][0]) and np.array_equal(recv[12:18], out['Q'][0])
    idx, Hs = ctx.select_fetch()
    assert recv[-1] == idx[0] and np.array_equal(recv[18:18 + 80], out['H'][0, idx[0]])
    assert ctx.allreduce([3.0], 'max')[0] == 3.0
    full = np.empty(ctx.full_count())
    ctx.gather(0, [ctx.full_count()], full=True, recv=full)
    assert np.array_equal(full[:6 * 80].reshape(6, 80), out['H'][0])
    ctx.comm_destroy()
    ctx.close()
