"""chain_kernel_lv (mxe_kernel_lv.hip.h): the lock-step layout with V^T resident in the LDS as binary32.  It IS the launch
for mxe_opts.precision = F32 where the basis fits the LDS (reference path: the alpha scan of maxent_loop.py:241-245 with the
minimiser of levenberg_minimizer.py:123-248; BASELINE config 5 asks for the binary32 leg), and on request
(mxe_opts.lds_basis = 1) the first pass of a binary64 launch."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                 # noqa: E402
from maxent_amd import device                                # noqa: E402

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


def solve(batch, **opts):
    ctx = bench.stage(batch, 0)
    n = len(batch['elems'])
    ctx.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(**opts))
    ctx.launch()
    info = ctx.last_launch_info()
    left = ctx.finish()
    out = ctx.fetch(want_v=False, want_H=True)
    out['H'] = np.array(out['H'])
    out['audit'] = ctx.audit()['corr']
    out['depth'] = ctx.launch_depth()
    ctx.close()
    return out, info, left


@pytest.mark.parametrize('n_omega,n_alpha', [(200, 20), (500, 40)])
def test_binary32_launch_runs_in_the_lds_resident_kernel(n_omega, n_alpha):
    batch = bench.build_batch(3, 100 if n_omega == 200 else 200, n_omega, n_alpha, 0)      # 3 normal + 6 plus-minus scans
    ref, info64, _ = solve(batch)
    out, info, left = solve(batch, precision=device.PRECISION_F32)
    assert info['kernel'] == 'mxe::chain_kernel_lv' and info['waves_per_chain'] == 8, info
    assert info['lds_bytes'] <= 160 * 1024 - 2304
    assert out['converged'].all() and left == 0
    e = rel_l2(out['H'], ref['H'])
    # binary32 classes: everything within 1e-4, nearly everything within 1e-5 -- and the device's own audit
    # (exact binary64 Newton correction at the returned v) says the same
    assert e.max() < 1e-4 and np.mean(e < 1e-5) > 0.97, (e.max(), np.mean(e < 1e-5))
    assert np.nanmax(out['audit']) < 1e-4
    np.testing.assert_allclose(out['chi2'], ref['chi2'], rtol=2e-3)
    # the one-chain binary32 kernel is still there (lds_basis = 2), and agrees
    old, info_old, _ = solve(batch, precision=device.PRECISION_F32, lds_basis=2)
    assert info_old['kernel'].startswith('mxe::chain_kernel<') and 'float' in info_old['kernel']
    assert rel_l2(old['H'], out['H']).max() < 2e-4


def test_chain_kernel_lv_against_the_reference_anchored_truth_cfg2():
    """chain_kernel_lv against the ORACLE, not against the repo's own binary64 kernel: BASELINE config 2 (the
    reference-generated fixture cfg2_normal.npz: n_tau = 200, n_omega = 500, 100 alpha) in precision = F32 -- the kernel that ran
    is asserted by name --, class 1e-4 against ``H_truth`` (the reference's own iterate polished in extended precision,
    tests/golden/make_golden.py), 5e-5 against the reference's raw H (its own stopping slack is 9e-6), device audit < 1e-4.
    Reference path: levenberg_minimizer.py:123-248 around maxent_loop.py:241-245; BASELINE.json config 5 (fp32 leg)."""
    from maxent_amd import hostprep
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cfg2_normal.npz')
    with np.load(gold, allow_pickle=False) as d:
        g = {k: d[k] for k in d.files}
    ctx = device.DeviceContext(g['U'], g['S'], g['V'], device=0)          # the reference's own decomposition
    ds = ctx.add_dataset(np.asarray(g['err']) * np.ones(len(g['tau'])))
    D = np.asarray(g['D'])
    ctx.set_elements([ds], [g['G']], D[None, :], [device.ENTROPY_NORMAL])
    v0 = hostprep.initial_v(g['V'], D, np.asarray(g['delta']), device.ENTROPY_NORMAL)[None, :]
    out = ctx.solve_chains([0], g['alpha'], v0, device.default_opts(precision=device.PRECISION_F32))      # g['alpha']: alpha * n_tau
    info = ctx.last_launch_info()
    audit = ctx.audit()['corr']
    ctx.close()
    assert info['kernel'] == 'mxe::chain_kernel_lv', info
    assert out['converged'].all()
    rows = g['rows']
    e_truth = rel_l2(out['H'][0][rows], g['H_truth'])
    assert np.all(np.isfinite(e_truth)) and 1e-9 < e_truth.max() < 1e-4, e_truth.max()
    e_ref = rel_l2(out['H'][0][rows], g['H_ref'])
    assert e_ref.max() < 5e-5, e_ref.max()
    assert np.all(np.isfinite(audit)) and np.nanmax(audit) < 1e-4
    np.testing.assert_allclose(out['chi2'][0], g['chi2_ref'], rtol=2e-3)


def test_chain_kernel_lv_against_the_reference_anchored_truth_cfg3():
    """BASELINE config 3 (4 x 4 matrix, 16 scans x 100 alpha) in precision = F32: chain_kernel_lv by name, one diagonal and one
    off-diagonal element against the extended-precision truth reached from the ORACLE PORT's iterates (tests/anchor.py), class
    1e-4; every problem's device audit < 1e-4."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import anchor
    from oracle import ref_numpy as R
    batch = bench.build_batch(4, 200, 500, 100, 0)
    out, info, left = solve(batch, precision=device.PRECISION_F32)
    assert info['kernel'] == 'mxe::chain_kernel_lv', info
    assert out['converged'].all() and left == 0
    assert np.all(np.isfinite(out['audit'])) and np.nanmax(out['audit']) < 1e-4
    K = batch['K']
    rows = (0, 40, 80, 99)
    for c in (0, 1):                       # element (0, 0): normal entropy; (0, 1): plus-minus
        i, j = batch['elems'][c]
        ent = 'normal' if i == j else 'plusminus'
        p = R.Problem(np.array(K.K), K.U, K.S, K.V, batch['Gmat'][i, j], batch['err'], batch['D'], entropy=ent)
        truth, ref = anchor.truth_rows(p, batch['omega'].delta, batch['alphas'], len(batch['tau']), rows, ent)
        for ia in rows:
            e = anchor.rel_l2_checked(out['H'][c, ia], truth[ia])
            assert e < 1e-4, (c, ia, e)
        assert rel_l2(out['H'][c], ref['H']).max() < 1e-4


def test_a_binary32_request_on_a_basis_that_does_not_fit_the_lds_is_promoted():
    """n_omega_pad = 1024: 56 x 1028 floats are 230 KB, chain_kernel_lv cannot hold them.  The one-chain binary32 kernel -- V
    streamed from the L2 by every chain -- takes 3.6-7.8 ms where the binary64 lock-step kernel takes 0.5-1.7 (8 x 8 / 16 x 16
    elements x 100 alpha, n_omega 640 ... 1500) and stops at a rounding floor of ~1e-3: binary32 is asked for as the cheaper
    arithmetic, so the launch is promoted (missing 5).  lds_basis = 2 keeps the one-chain binary32 kernel."""
    batch = bench.build_batch(2, 100, 1000, 10, 0)
    out, info, _ = solve(batch, precision=device.PRECISION_F32)
    assert info['kernel'].startswith('mxe::chain_kernel_mc<'), info
    assert out['converged'].all() and np.nanmax(out['audit']) < 1e-6
    old, info_old, _ = solve(batch, precision=device.PRECISION_F32, lds_basis=2)
    assert info_old['kernel'].startswith('mxe::chain_kernel<') and 'float' in info_old['kernel'], info_old
    assert old['converged'].all()
    assert rel_l2(old['H'], out['H']).max() < 2e-3


def test_binary32_first_pass_then_one_binary64_step_per_alpha():
    """mxe_opts.lds_basis = 1: every alpha to 1e-5 in chain_kernel_lv, then every alpha as a piece of its own in the binary64
    lock-step kernel from that v.  Same fixed points as the single binary64 pass, iteration counts of both passes added."""
    batch = bench.build_batch(3, 200, 500, 40, 0)
    ref, info64, _ = solve(batch, lds_basis=2)
    out, info, left = solve(batch, lds_basis=1)
    assert info['kernel'].startswith('mxe::chain_kernel_lv + mxe::chain_kernel_mc<32'), info
    assert 'chain_kernel_lv' not in info64['kernel']
    assert out['converged'].all() and left == 0
    assert rel_l2(out['H'], ref['H']).max() < 1e-7
    assert np.nanmax(out['audit']) < 1e-6 and np.nanpercentile(out['audit'], 99) < 1e-8
    np.testing.assert_allclose(out['chi2'], ref['chi2'], rtol=1e-7)
    assert np.all(out['n_iter'] >= 2) and np.all(out['n_evals'] >= out['n_iter'])      # (first pass + at least the step and its check)
    # the depth of both passes as the kernels counted it (mxe_launch_depth): the second pass is a few rounds per alpha
    d, d64 = out['depth'], ref['depth']
    assert d['max_rounds'][0] >= 10 and 2 <= d['max_rounds'][1] <= 12 and d64['max_rounds'][0] >= 10 and d64['max_rounds'][1] == 0, (d, d64)
    assert all(0 < m <= x for m, x in zip(d['mean_rounds'], d['max_rounds']))


def test_a_binary32_batch_that_fills_the_gpu():
    """A binary32 request on a batch that fills the GPU at two workgroups per CU is promoted to the binary64 kernel that runs that
    way (0.81 ms against 1.24 ms in chain_kernel_lv: binary32 is asked for as the cheaper arithmetic).  Held in chain_kernel_lv
    (wg_per_cu = 1) it keeps the uniform pieces: one workgroup per CU is not always 'a launch that does not fill the GPU' -- the cut of
    the pieces by cost (one piece per slot) is for small launches; applied to the 25 600-problem batch it left ONE piece of 100
    alphas per plus-minus scan (227 rounds deep, 3.2 ms instead of 1.2)."""
    batch = bench.build_batch(16, 200, 500, 100, 0)
    ref, info64, _ = solve(batch)
    out, info, left = solve(batch, precision=device.PRECISION_F32)
    assert info['kernel'] == info64['kernel'] == 'mxe::chain_kernel_mc<32, 2>' and out['converged'].all() and left == 0
    assert rel_l2(out['H'], ref['H']).max() < 1e-7 and np.nanmax(out['audit']) < 1e-6      # (the binary64 kernel: binary64 answers)
    out, info, left = solve(batch, precision=device.PRECISION_F32, wg_per_cu=1)
    assert info['kernel'] == 'mxe::chain_kernel_lv' and out['converged'].all() and left == 0
    assert 40 <= out['depth']['max_rounds'][0] <= 130, out['depth']
    assert np.nanmax(out['audit']) < 1e-4


def test_binary32_with_more_than_32_coupled_directions_is_promoted_to_the_binary64_lock_step_build():
    """chain_kernel_lv has the plain 32-row build only.  A job whose smallest alphas couple more directions (error bars far below
    the noise) is promoted to binary64 -- the lock-step build with the 64-row block is the cheaper arithmetic there (the one-chain
    binary32 kernel took 300 times as long) -- and that is decided BEFORE the scans are cut: the first form of the fallback came
    after the lock-step schedule had dropped the alphas behind its cuts, whose records then were garbage (STRESS_F32=1
    tools/stress.py)."""
    from maxent_amd import synthetic, hostprep
    import maxent_amd as mx
    tau, omega, K, Gmat, _ = synthetic.matrix_G(2, 200, 500, seed=11)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    alphas = np.array(mx.LogAlphaMesh(alpha_min=1.2e-2, alpha_max=1.6e4, n_points=20)) * 200
    elems = [(0, 0), (0, 1), (1, 0), (1, 1)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    outs = {}
    for prec in (device.PRECISION_F64, device.PRECISION_F32):
        ctx = device.DeviceContext(K.U, K.S, K.V)
        ds = ctx.add_dataset(2e-5 * np.ones(200))
        ctx.set_elements([ds] * 4, [Gmat[i, j] for i, j in elems], np.tile(D, (4, 1)), kinds)
        ctx.upload_chains(np.arange(4), alphas, v0, device.default_opts(precision=prec, maxiter=300))
        ctx.launch()
        kernel = ctx.last_launch_info()['kernel']
        ctx.finish()
        outs[prec] = (ctx.fetch(want_v=False, want_H=True), kernel)
        ctx.close()
    o32, k32 = outs[device.PRECISION_F32]
    o64, _ = outs[device.PRECISION_F64]
    assert k32 == outs[device.PRECISION_F64][1] and k32.startswith('mxe::chain_kernel_mc<64'), k32
    assert np.isin(o32['converged'], (0, 1)).all() and o32['n_evals'].min() >= 1 and o32['n_evals'].max() < 100000
    assert np.array_equal(o32['converged'], o64['converged']) and o32['converged'].sum() >= 40
    assert np.array_equal(o32['H'], o64['H'])                 # (the same launch)
