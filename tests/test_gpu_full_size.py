"""GPU tests at BASELINE.json's full sizes through size-independent properties
(the oracle cannot run 25 600 alpha-solves in seconds): convergence of every
problem, stationarity of the returned points, symmetry, invariance under the
way the library cuts and schedules the alpha scans, and agreement with the
extended-precision truth on a sample."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import anchor                                                # noqa: E402
import bench                                                 # noqa: E402
from maxent_amd import device                                # noqa: E402
from oracle import ref_numpy as R, sform as SF               # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cfg4():
    batch = bench.build_batch(16, 200, 500, 100, 0)
    ctx = bench.stage(batch, 0)
    out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
    info = ctx.last_launch_info()
    yield batch, ctx, out, info
    ctx.close()


def test_cfg4_all_converged_and_finite(cfg4):
    batch, ctx, out, info = cfg4
    assert out['converged'].all() and out['converged'].shape == (256, 100)
    for k in ('H', 'v', 'chi2', 'S', 'Q'):
        assert np.all(np.isfinite(out[k])), k
    # lock-step layout, persistent grid: one or two workgroups of four waves per CU
    assert info['waves_per_chain'] == 4 and info['n_workgroups'] in (256, 512) and 'chain_kernel_mc' in info['kernel']
    assert out['n_iter'].max() < 60 and 1.5 < out['n_iter'].mean() < 5.0
    # chi2 decreases and the entropy becomes more negative as alpha decreases
    assert np.all(np.diff(out['chi2'], axis=1) < 1e-9 * out['chi2'][:, 1:])
    assert np.all(np.diff(out['S'], axis=1) < 1e-12)
    np.testing.assert_allclose(out['Q'], 0.5 * out['chi2'] - batch['alphas'][None, :] * out['S'], rtol=1e-12)


def test_cfg4_every_problem_passes_the_device_audit(cfg4):
    """mxe_audit: the exact Newton correction (binary64, all n_s directions) at the returned v of ALL 25 600
    problems -- to first order the distance of the returned H from the minimiser.  The sampled truth
    comparison below calibrates it: where both exist they agree in magnitude."""
    batch, ctx, out, info = cfg4
    a = ctx.audit()
    c = a['corr']
    assert c.shape == (256, 100) and np.all(np.isfinite(c))
    assert c.max() < 1e-6, c.max()               # the parity gate, every problem
    assert np.percentile(c, 99) < 1e-8 and np.median(c) < 1e-10


def test_cfg4_symmetric_input_gives_symmetric_output(cfg4):
    """G_ij = G_ji (symmetrised noise): chains (i,j) and (j,i) are solved by
    different slots / pieces and must agree to the convergence level."""
    batch, ctx, out, info = cfg4
    H = out['H'].reshape(16, 16, 100, 500)
    chi2 = out['chi2'].reshape(16, 16, 100)
    num = np.linalg.norm(H - H.transpose(1, 0, 2, 3), axis=-1)
    den = np.linalg.norm(H, axis=-1)
    assert (num / den).max() < 1e-8
    np.testing.assert_allclose(chi2, chi2.transpose(1, 0, 2), rtol=1e-7)


def test_cfg4_stationarity_and_truth_on_a_sample(cfg4):
    """at the returned v the singular-space gradient W g vanishes (oracle
    S-form evaluation) and H equals the extended-precision fixed point."""
    batch, ctx, out, info = cfg4
    K = batch['K']
    basis = SF.Basis(K.U, K.S, K.V, batch['err'])
    rng = np.random.RandomState(7)
    worst = 0.0
    rows = (0, 37, 87, 99)
    for c in [0, 1, 17, 100, 255] + list(rng.randint(0, 256, 2)):
        i, j = batch['elems'][c]
        ent = 'normal' if batch['kinds'][c] == device.ENTROPY_NORMAL else 'plusminus'
        el = SF.Element(basis, batch['Gmat'][i, j], batch['D'], ent)
        # the truth is reached from the iterates of the reference's algorithm (oracle port), not from the GPU's
        p = R.Problem(np.array(K.K), K.U, K.S, K.V, batch['Gmat'][i, j], batch['err'], batch['D'], entropy=ent)
        truth, _ = anchor.truth_rows(p, batch['omega'].delta, batch['alphas'], len(batch['tau']), rows, ent)
        for ia in rows:
            a, v = batch['alphas'][ia], out['v'][c, ia]
            ev = SF.evaluate(basis, el, a, basis.from_v(v))
            assert np.linalg.norm(ev['H'] - out['H'][c, ia]) / np.linalg.norm(ev['H']) < 1e-11
            assert abs(ev['chi2'] - out['chi2'][c, ia]) / ev['chi2'] < 1e-10
            g = basis.c * ev['rho'] + a * basis.from_v(v)
            d = SF.gram(basis, ev['w']) @ g
            scale = SF.gram(basis, ev['w']) @ (np.abs(basis.c * ev['rho']) + a * np.abs(basis.from_v(v)))
            # gradient small against the size of its (cancelling) terms; the accuracy of H
            # itself is checked against the extended-precision truth below
            assert np.max(np.abs(d)) < 5e-3 * np.max(scale)
            e = anchor.rel_l2_checked(out['H'][c, ia], truth[ia])
            assert e < 1e-6, (c, ia, e)
            worst = max(worst, e)
    assert 0 < worst < 1e-6, worst


@pytest.mark.parametrize('opts', [dict(alpha_split=1, chains_per_wg=1), dict(alpha_split=4, chains_per_wg=1),
                                  dict(alpha_split=3, chains_per_wg=4), dict(alpha_split=16, chains_per_wg=4)])
def test_cfg4_invariant_under_scheduling(cfg4, opts):
    """cutting the alpha scans into cold-started pieces and the workgroup
    layout must not change the per-alpha answers."""
    batch, ctx, out, info = cfg4
    other = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'],
                             device.default_opts(**opts), want_v=False)
    assert other['converged'].all()
    e = np.linalg.norm(other['H'] - out['H'], axis=-1) / np.linalg.norm(out['H'], axis=-1)
    # both runs stop on the same estimate (tol_h = 1e-9) at different iterates: they differ by at most
    # the sum of what each has left (measured 1.1e-8 in the worst of the 25 600; gate vs the truth: 1e-6)
    assert e.max() < 3e-8, e.max()
    np.testing.assert_allclose(other['chi2'], out['chi2'], rtol=1e-7)
    np.testing.assert_allclose(other['S'], out['S'], rtol=1e-7, atol=1e-12)


def test_pieces_in_the_tail_of_a_normal_entropy_scan_are_led_or_joined(cfg4):
    """50 pieces of two alphas per scan: pieces of the normal-entropy scans that start in the last 6 % of the
    logarithmic alpha range are led by the last alpha above it (the `lead` build of the kernel) or joined to the piece
    before them; the answers are those of the default schedule, and no cold start runs away (the rank of a sharded job
    that holds element 221 took 2.7 ms instead of 0.5 before, profiles/r02_f_cold_start_profile.txt)"""
    batch, ctx, out, info = cfg4
    diag = np.array([e for e in range(256) if batch['kinds'][e] == 0], dtype=np.int32)
    assert len(diag) == 16
    other = ctx.solve_chains(diag, batch['alphas'], batch['v0'][diag], device.default_opts(alpha_split=50, chains_per_wg=4), want_v=False)
    assert ctx.last_launch_info()['kernel'].endswith('lead>')
    assert other['converged'].all()
    e = np.linalg.norm(other['H'] - out['H'][diag], axis=-1) / np.linalg.norm(out['H'][diag], axis=-1)
    assert e.max() < 3e-8, e.max()
    assert other['n_evals'].max() <= 30, other['n_evals'].max()          # was 157 (element 221, alpha index 98)


def test_cfg3_elementwise_api_matches_direct_batch():
    """cfg3 (4x4 elements x 100 alpha) through ElementwiseMaxEnt equals the same
    problems handed to the C-ABI directly."""
    import maxent_amd as mx
    batch = bench.build_batch(4, 200, 500, 100, 0)
    ctx = bench.stage(batch, 0)
    direct = ctx.solve_chains(np.arange(16, dtype=np.int32), batch['alphas'], batch['v0'], want_v=False)
    ctx.close()
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(batch['tau'], batch['Gmat'])
    ew.omega = batch['omega']
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=100)
    ew.set_error(1e-4)
    res = ew.run()
    assert res.A.shape == (4, 4, 100, 500) and not np.any(np.isnan(res.A))
    H = direct['H'].reshape(4, 4, 100, 500)
    e = np.linalg.norm(res.H - H, axis=-1) / np.linalg.norm(H, axis=-1)
    assert e.max() < 1e-8
    np.testing.assert_allclose(res.A, res.H / batch['omega'].delta, rtol=1e-14)
    assert res.A_out.shape == (4, 4, 500)
    # diagonal spectra are normalised, off-diagonal ones integrate to ~0
    w = np.asarray(batch['omega'])
    for i in range(4):
        assert abs(np.trapezoid(res.A_out[i, i], w) - 1) < 2e-2


def test_the_solo_schedule_is_dropped_where_the_placement_rule_does_not_hold(cfg4, monkeypatch):
    """the workgroups that get a CU to themselves rest on an observed placement (workgroups b and
    b + gridDim / 2 share a CU).  The library probes it per device; forced to 'does not hold' the schedule has no solo
    workgroups and the batch still passes the audit."""
    batch, ctx0, out0, info0 = cfg4
    ctx = bench.stage(batch, 0)
    ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
    normal = ctx.schedule_info()
    assert normal['placement_rule'] in (1, 2) and (normal['n_solo'] > 0) == (normal['placement_rule'] == 1), normal
    monkeypatch.setenv('MXE_FORCE_NO_SOLO_RULE', '1')
    ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
    forced = ctx.schedule_info()
    assert forced == dict(n_solo=0, placement_rule=2), forced
    ctx.launch()
    assert ctx.finish() == 0
    out = ctx.fetch(want_v=False, want_H=False)
    depth = ctx.launch_depth()
    c = ctx.audit()['corr']
    ctx.close()
    assert out['converged'].all() and c.max() < 1e-6 and np.percentile(c, 99) < 1e-8
    np.testing.assert_allclose(out['chi2'], out0['chi2'], rtol=1e-7)
    assert depth['max_rounds'] == [0, 0]       # (the build of the batch that fills the GPU does not count its rounds: see mxe_launch_depth)


def test_the_depth_of_the_full_batch_launch_behind_its_switch(cfg4, monkeypatch):
    """MXE_COUNT_ROUNDS: the launch of the batch that fills the GPU counts the rounds of its workgroups too (a diagnostic launch:
    bench.py reports it as launch_depth) -- the slowest workgroups run 38-46 rounds where the mean runs 33-34: the tail that
    profiles/r05_experiments.txt is about, as a number"""
    batch, ctx, out, info = cfg4
    ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])      # (the default schedule: other tests re-cut the fixture's chains)
    monkeypatch.setenv('MXE_COUNT_ROUNDS', '1')
    ctx.launch()
    ctx.sync()
    d = ctx.launch_depth()
    monkeypatch.delenv('MXE_COUNT_ROUNDS')
    assert 30 <= d['mean_rounds'][0] <= 37 and 38 <= d['max_rounds'][0] <= 50, d
    ctx.launch()
    ctx.sync()
    assert ctx.launch_depth()['max_rounds'] == [0, 0]


def test_the_full_batch_repeats_bit_for_bit(cfg4):
    """chain_kernel_mc<32, 2>, 512 workgroups that take their pieces from a queue in whatever order they get to it: three
    launches of the same chains return the same bits -- v, H, chi2, S, Q, the iteration and evaluation counts.  A piece does not
    depend on its company in the workgroup, and the partial h of the four waves are summed in pairs (two additions onto zero
    commute), the pair sums in a fixed order; rounds 2-4 added the four with LDS atomics as they arrived and half the chi2 of
    this batch moved by ~1e-10 from launch to launch."""
    batch, ctx, out, info = cfg4
    ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
    runs = []
    for _ in range(3):
        ctx.launch()
        ctx.finish()
        runs.append(ctx.fetch())
    assert 'chain_kernel_mc<32, 2>' in ctx.last_launch_info()['kernel'] and ctx.last_launch_info()['n_workgroups'] == 512
    for r in runs[1:]:
        for k in ('v', 'H', 'chi2', 'S', 'Q', 'n_iter', 'n_evals', 'converged'):
            assert np.array_equal(r[k], runs[0][k], equal_nan=True), k


def test_batches_in_flight_are_cut_into_fewer_pieces_and_give_the_same_answers(cfg4):
    """mxe_opts.in_flight = n: the caller keeps n batches of this size in flight (bench.py --in-flight n: n contexts take the
    steps in turn), so each is cut into about 1 / n as many cold-started pieces -- fewer evaluations per alpha, the same fixed
    points.  Two contexts launched without waiting in between both deliver."""
    batch, ctx0, out0, info0 = cfg4
    lanes = []
    for _ in range(2):
        c = bench.stage(batch, 0)
        c.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(in_flight=4))
        lanes.append(c)
    for _ in range(3):
        for c in lanes:
            c.launch()                         # (no wait in between: the launches of the two contexts overlap)
    outs = []
    for c in lanes:
        assert c.finish() == 0
        info = c.last_launch_info()
        assert info['kernel'] == info0['kernel'] and info['n_workgroups'] < info0['n_workgroups'], (info, info0)
        o = c.fetch(want_v=False, want_H=True)
        a = c.audit()['corr']
        assert o['converged'].all() and a.max() < 1e-6 and np.percentile(a, 99) < 1e-8
        outs.append(o)
    for c in lanes:
        c.close()
    for o in outs:
        assert o['n_evals'].mean() < out0['n_evals'].mean() - 0.15          # (4 cold starts per scan instead of 15)
        np.testing.assert_allclose(o['chi2'], out0['chi2'], rtol=1e-7)
        e = np.linalg.norm(o['H'] - out0['H'], axis=-1) / np.linalg.norm(out0['H'], axis=-1)
        assert e.max() < 1e-7
    with pytest.raises(device.MaxEntDeviceError):
        ctx0.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(in_flight=-1))
    ctx0.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])       # (the fixture's state back)
    ctx0.launch(); ctx0.finish()


def test_a_batch_of_more_scans_than_slots_is_cut_by_what_the_pieces_cost():
    """46 x 46 elements x 40 alphas: 2 116 scans for 2 048 slots.  The rule of the BASELINE batch -- two pieces per slot -- leaves such
    a batch one piece per scan, and the slots that take a second scan run twice as long as the others (48 x 48 x 100: 9.6 ms where
    six pieces per scan take 5.6; round 5).  The count now weighs a piece's cold start against the imbalance of the queue: more,
    shorter pieces -- the same answers, every alpha through the audit, and a shorter launch than the scans left whole."""
    batch = bench.build_batch(46, 40, 100, 40, 0)
    ctx = bench.stage(batch, 0)
    n = len(batch['elems'])
    res = {}
    for name, split in (('auto', 0), ('whole scans', 1)):
        ctx.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts(alpha_split=split))
        ts = []
        for _ in range(5):
            ctx.launch(); ctx.sync(); ts.append(ctx.last_kernel_ms())
        left = ctx.finish()
        out = ctx.fetch(want_v=False)
        assert out['converged'].all() and left == 0, name
        assert np.nanmax(ctx.audit()['corr']) < 1e-6, name
        res[name] = (min(ts), np.array(out['H']), float(out['n_evals'].mean()))
    ctx.close()
    assert res['auto'][2] > res['whole scans'][2]                       # (more cold starts ...)
    assert res['auto'][0] < 0.85 * res['whole scans'][0], (res['auto'][0], res['whole scans'][0])     # (... and a shorter launch)
    d = np.linalg.norm(res['auto'][1] - res['whole scans'][1], axis=-1) / np.linalg.norm(res['whole scans'][1], axis=-1)
    assert d.max() < 1e-6, d.max()


def test_several_data_sets_pieces_of_like_cost_share_a_workgroup(cfg4, monkeypatch):
    """The BASELINE batch with TWO data sets (two error bars: elements alternate): a workgroup streams one basis, so its four pieces
    come from one data set and it takes no others (static layout, no queue).  Its slots run in lock-step until the longest piece is
    through: pieces of like cost go together and the longest workgroups first (round 5: 1.42 -> 0.95 ms; a data set per element
    2.13 -> 1.29 ms, tools/many_datasets.py).  A piece does not depend on its company: the answers are the same bits either way."""
    batch = cfg4[0]
    K = batch['K']
    n = len(batch['elems'])
    res = {}
    for name, env in (('sorted', None), ('scan order', '1')):
        if env:
            monkeypatch.setenv('MXE_NO_SORTED_STATIC', env)
        ctx = device.DeviceContext(K.U, K.S, K.V, device=0)
        ds = [ctx.add_dataset(batch['err'] * (1.0 + 1e-3 * k)) for k in range(2)]
        ctx.set_elements([ds[e % 2] for e in range(n)], [batch['Gmat'][batch['elems'][e]] for e in range(n)],
                         np.tile(batch['D'], (n, 1)), batch['kinds'])
        ctx.upload_chains(np.arange(n, dtype=np.int32), batch['alphas'], batch['v0'])
        ts = []
        for _ in range(5):
            ctx.launch(); ctx.sync(); ts.append(ctx.last_kernel_ms())
        assert ctx.finish() == 0
        out = ctx.fetch(want_v=False)
        assert out['converged'].all() and np.nanmax(ctx.audit()['corr']) < 1e-6
        res[name] = (min(ts), np.array(out['H']), np.array(out['chi2']), np.array(out['n_evals']))
        ctx.close()
        if env:
            monkeypatch.delenv('MXE_NO_SORTED_STATIC')
    for k in (1, 2, 3):
        assert np.array_equal(res['sorted'][k], res['scan order'][k])
    assert res['sorted'][0] < 0.85 * res['scan order'][0], (res['sorted'][0], res['scan order'][0])
