"""Coarse alpha meshes: 3-8 alphas over 4-6 decades -- the reference's own tests and defaults (test/python/tau_maxent.py:44
``LogAlphaMesh(alpha_min=0.08, n_points=5)``, alpha_meshes.py:81).  A warm step over a factor 7 ... 600 in alpha into the region
where the entropy term no longer holds the solution took 250-2 300 evaluations (profiles/r05_b_coarse_mesh.txt); such an alpha is
now a piece of its own that starts cold higher up and walks down a ladder of alphas the library lays for it (mxe_chains_upload).
The minimiser does not depend on the path: answers as before, in a fraction of the rounds."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import anchor                                                # noqa: E402
import maxent_amd as mx                                      # noqa: E402
from maxent_amd import device, synthetic, hostprep           # noqa: E402
from oracle import ref_numpy as R                            # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def solve(K, omega, Gs, kinds, err, alphas):
    D = synthetic.flat_D(omega)
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    n = len(Gs)
    ctx.set_elements([ds] * n, list(Gs), np.tile(D, (n, 1)), kinds)
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    out = ctx.solve_chains(np.arange(n), alphas, v0)
    info, depth = ctx.last_launch_info(), ctx.launch_depth()
    audit = ctx.audit()['corr']          # (of the finished results: the launch below is for the clock and runs no finishing pass)
    ctx.launch()
    ctx.sync()
    ms = ctx.last_kernel_ms()
    ctx.close()
    return out, info, depth, ms, audit


def test_eight_alphas_over_six_decades_like_smoke(monkeypatch):
    """the launch of __graft_entry__.smoke(): 2 scans x 8 alphas, ratio 7.2 between neighbours.  Round 4: 308 rounds deep, 2.6 ms
    (913 evaluations for the last alpha of the normal-entropy scan); now one cold start + one walk deep."""
    n_tau, n_omega = 60, 120
    tau, omega, K, G = synthetic.single_G(n_tau, n_omega)
    K.reduce_singular_space(1e-14)
    err = synthetic.SIGMA * np.ones(n_tau)
    alphas = np.array(synthetic.alpha_mesh(8)) * n_tau
    kinds = [device.ENTROPY_NORMAL, device.ENTROPY_PLUSMINUS]
    out, info, depth, ms, audit = solve(K, omega, [G, G], kinds, err, alphas)
    assert out['converged'].all() and np.nanmax(audit) < 1e-8
    assert 'lead' in info['kernel'] and depth['max_rounds'][0] <= 45, (info, depth)
    assert ms < 1.0, ms                                   # (measured 0.28 ms; round 4: 2.6)
    assert out['n_evals'].max() <= 32
    # the same fixed points as the scan without ladders (every alpha warm from its neighbour), and as the oracle's truth
    monkeypatch.setenv('MXE_NO_LADDER', '1')
    old, info_old, depth_old, _, _ = solve(K, omega, [G, G], kinds, err, alphas)
    monkeypatch.delenv('MXE_NO_LADDER')
    assert depth_old['max_rounds'][0] > 100                # (what this replaces)
    assert (np.linalg.norm(out['H'] - old['H'], axis=-1) / np.linalg.norm(old['H'], axis=-1)).max() < 1e-7
    D = synthetic.flat_D(omega)
    for c, ent in enumerate(('normal', 'plusminus')):
        p = R.Problem(np.array(K.K), K.U, K.S, K.V, G, err, D, entropy=ent)
        truth, _ = anchor.truth_rows(p, omega.delta, alphas, n_tau, (0, 6, 7), ent)
        for ia, Ht in truth.items():
            assert anchor.rel_l2_checked(out['H'][c, ia], Ht) < 1e-6, (c, ia)


def test_three_alphas_over_five_decades_on_a_matrix():
    """tools/stress.py case 1 in small: a 4 x 4 matrix, three alphas a factor ~600 apart (round 4: 253 evaluations for the last)"""
    tau, omega, K, Gmat, _ = synthetic.matrix_G(4, 200, 257, seed=3)
    K.reduce_singular_space(1e-14)
    err = 9e-5 * np.ones(200)
    alphas = np.array(mx.LogAlphaMesh(alpha_min=3.2e-2, alpha_max=1.1e4, n_points=3)) * 200
    elems = [(i, j) for i in range(4) for j in range(4)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    out, info, depth, ms, audit = solve(K, omega, [Gmat[i, j] for i, j in elems], kinds, err, alphas)
    assert out['converged'].all() and np.nanmax(audit) < 1e-7
    assert depth['max_rounds'][0] <= 45 and out['n_evals'].max() <= 32, (depth, out['n_evals'].max())
    assert ms < 1.0, ms


def test_the_reference_test_mesh_has_a_time_bound():
    """reference test/python/tau_maxent.py:44: five alphas from 20 down to 0.08 on 201 data points -- the known-answer test of
    tests/test_gpu_api.py::test_known_answer_log_probability checks its numbers; here: how long the launch takes"""
    with np.load(os.path.join(GOLD, 'kat_tau_maxent.npz'), allow_pickle=False) as d:
        g = {k: d[k] for k in d.files}
    tm = mx.TauMaxEnt()
    tm.set_verbosity(mx.VerbosityFlags.Quiet)
    tm.set_G_tau_data(g['tau'], g['G'])
    tm.alpha_mesh = mx.LogAlphaMesh(alpha_min=0.08, n_points=5)
    tm.omega = mx.HyperbolicOmegaMesh(omega_min=-10, omega_max=10, n_points=200)
    tm.set_error(1.e-3)
    res = tm.run()
    assert np.all(res.converged)
    assert (np.linalg.norm(res.H - g['H_truth'], axis=-1) / np.linalg.norm(g['H_truth'], axis=-1)).max() < 1e-6
    res = None
    tm.run()                                               # (warm: the code object is loaded)
    assert tm.maxent_loop.last_launch['kernel_ms'] < 1.0, tm.maxent_loop.last_launch


def test_the_reference_default_alpha_mesh_reaches_far_below_the_noise():
    """``LogAlphaMesh()`` -- the reference's default, alpha_meshes.py:81: 20 alphas from 20 down to 1e-4, a factor 1.9 apart -- on 200 data
    points ends at alpha~ = 0.02, where the last alphas couple more directions than the lock-step layout takes and are left to the
    finishing pass (``mxe_chains_finish``).  A warm step over the factor 1.9 took 500-2 300 evaluations there (round 4: 921 / 486 /
    2 269 for the last three alphas); the pass now lays rungs of <= 1.3 between the alphas of such a mesh
    (``KParams::out_index``): tens of evaluations, the same minimisers (device audit)."""
    tau, omega, K, G = synthetic.single_G(200, 500)
    K.reduce_singular_space(1e-14)
    alphas = np.array(mx.LogAlphaMesh()) * 200
    out, info, depth, ms, audit = solve(K, omega, [G], [device.ENTROPY_NORMAL], synthetic.SIGMA * np.ones(200), alphas)
    assert out['converged'].all()
    assert np.nanmax(audit) < 1e-7, np.nanmax(audit)
    assert out['n_evals'].max() <= 150, out['n_evals']            # (round 4: 2 269)
    assert np.all(np.diff(out['chi2'][0]) < 0)                     # chi2 falls with alpha down to the last
