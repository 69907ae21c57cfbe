"""Jobs in flight behind the reference's API: ``ElementwiseMaxEnt.run_async()`` / ``maxent_amd.run_many``.

What they replace: ``ElementwiseMaxEnt.run()`` called once per job of a self-consistency loop (reference
elementwise_maxent.py:270-285).  Every object is prepared, staged and launched before the first is waited for; the results
must be those of the sequential ``run()`` calls."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maxent_amd as mx                                      # noqa: E402
from maxent_amd import device, synthetic                     # noqa: E402

pytestmark = pytest.mark.gpu
FIELDS = ('chi2', 'S', 'Q', 'H', 'A')


def make(n_orb, n_tau, n_w, n_alpha, k, **kw):
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_w, noise_seed=4000 + k)
    ew = mx.ElementwiseMaxEnt(use_hermiticity=False, **kw)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(tau, Gmat)
    ew.omega = omega
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=n_alpha)
    ew.set_error(synthetic.SIGMA)
    return ew


def snapshot(res):
    out = {f: np.array(getattr(res, f)) for f in FIELDS}
    out['A_out'] = np.array(res.A_out)
    out['converged'] = np.array(res.converged)
    out['alpha'] = np.array(res.alpha)
    out['picks'] = {name: np.array([[res.analyzer_results[i][j][name]['alpha_index'] for j in range(res.A.shape[1])]
                                    for i in range(res.A.shape[0])])
                    for name in ('LineFitAnalyzer', 'Chi2CurvatureAnalyzer', 'EntropyAnalyzer')}
    return out


def close(a, b, tol):
    den = np.maximum(np.abs(b), 1e-300)
    return np.nanmax(np.abs(a - b) / np.maximum(den, tol * np.nanmax(np.abs(b)))) if a.size else 0.0


@pytest.mark.parametrize('n_jobs', [2, 4])
def test_run_many_returns_what_the_sequential_runs_return(n_jobs):
    jobs = [make(3, 100, 200, 30, k) for k in range(n_jobs)]
    seq = [snapshot(ew.run()) for ew in jobs]
    for ew in jobs:
        ew.maxent_result = None
    res = mx.run_many(jobs, same_cut=False)
    assert len(res) == n_jobs and all(r is ew.maxent_result for r, ew in zip(res, jobs))
    for k, (r, s) in enumerate(zip(res, seq)):
        got = snapshot(r)
        assert np.all(got['converged'] == 1) and np.array_equal(got['alpha'], s['alpha'])
        for f in FIELDS + ('A_out',):
            assert got[f].shape == s[f].shape, (k, f)
            # (same_cut=False: jobs in flight are cut into fewer cold-started pieces -- mxe_opts.in_flight --: other iterates, the
            #  same minimisers to the stopping tolerance; the gate against the truth is 1e-6)
            assert np.all(np.isfinite(got[f]))
            rel = np.linalg.norm(got[f] - s[f]) / np.linalg.norm(s[f])
            assert rel < 1e-7, (k, f, rel)
        for name in got['picks']:
            assert np.array_equal(got['picks'][name], s['picks'][name]), (k, name)
    # different data really were different jobs
    assert np.linalg.norm(np.asarray(res[0].chi2) - np.asarray(res[1].chi2)) > 0
    # every object had contexts of its own while in flight, and the launches of the jobs were cut for n_jobs in flight
    assert all(len(ew.last_launches) >= 2 for ew in jobs)


@pytest.mark.parametrize('shape', [(3, 100, 200, 30), (16, 200, 500, 100)])
def test_run_many_is_the_sequential_runs_bit_for_bit(shape):
    """the default (``same_cut=True``): the jobs are in flight together and each is cut as ``run()`` cuts it -- every field, the
    spectra and the analyzers' choices are bitwise those of the sequential calls.  The second shape is
    the BASELINE batch (chain_kernel_mc<32, 2>, 512 workgroups): until round 5 its launch did not repeat bit for bit even when
    run() was called twice -- four atomic additions per partial h in the order the waves arrived; they are summed in pairs now"""
    jobs = [make(*shape, 20 + k) for k in range(4)]
    seq = [snapshot(ew.run()) for ew in jobs]
    wgs = [ew.last_launches[-1]['n_workgroups'] for ew in jobs]
    for ew in jobs:
        ew.maxent_result = None
    res = mx.run_many(jobs)
    for k, (r, s) in enumerate(zip(res, seq)):
        got = snapshot(r)
        for f in FIELDS + ('A_out', 'converged', 'alpha'):
            assert np.array_equal(got[f], s[f], equal_nan=True), (k, f)
        for name in got['picks']:
            assert np.array_equal(got['picks'][name], s['picks'][name]), (k, name)
    assert [ew.last_launches[-1]['n_workgroups'] for ew in jobs] == wgs      # (the cut of run(), not the one for four in flight)


def test_without_page_locked_blocks_the_results_are_copied_when_waited_for(monkeypatch):
    """the scalars and rows of a job go into one page-locked block behind the kernels (mxe_chains_prefetch); results a caller keeps hold
    their blocks, so beyond ``device.SMALL_PINNED_LIMIT`` bytes a job gets ordinary memory and the copies of round 4: the same bits"""
    jobs = [make(3, 100, 200, 30, 40 + k) for k in range(2)]
    res = mx.run_many(jobs)
    a = [snapshot(r) for r in res]
    assert device._small_pinned[0] > 0                  # (the blocks of the two results)
    del res
    for ew in jobs:
        ew.maxent_result = None
    import gc
    gc.collect()
    monkeypatch.setattr(device, 'SMALL_PINNED_LIMIT', 0)
    before = device._small_pinned[0]
    res = mx.run_many(jobs)
    assert device._small_pinned[0] == before
    b = [snapshot(r) for r in res]
    for x, y in zip(a, b):
        for f in FIELDS + ('A_out', 'converged'):
            assert np.array_equal(x[f], y[f], equal_nan=True), f


def test_run_async_with_the_cut_of_one_job_is_run():
    """``run()`` IS ``run_async().result()``; a handle gives its result once and again; an object cannot be started twice"""
    ew = make(2, 100, 200, 20, 7)
    a = snapshot(ew.run())
    ew.maxent_result = None
    h = ew.run_async()
    with pytest.raises(RuntimeError):
        ew.run_async()
    r = h.result()
    assert h.done and h.result() is r and r is ew.maxent_result
    b = snapshot(r)
    for f in FIELDS + ('A_out',):
        assert np.linalg.norm(a[f] - b[f]) <= 1e-9 * np.linalg.norm(a[f]), f
    ew.maxent_result = None
    ew.run_async().result()                       # (free again)


def test_jobs_in_flight_hold_different_device_contexts_and_give_them_back():
    from maxent_amd.batch_solver import BatchSolver
    jobs = [make(2, 100, 200, 20, 10 + k) for k in range(3)]
    handles = [ew.run_async(in_flight=3) for ew in jobs]
    solvers = [ew.maxent_diagonal.K.__dict__['_batch_solvers'][(0,)] for ew in jobs]
    assert len(set(map(id, solvers))) == 3 and all(s._busy for s in solvers)
    res = [h.result() for h in handles]
    assert not any(s._busy for s in solvers)
    assert all(np.all(np.asarray(r.converged) == 1) for r in res)
    # a fourth object on the same grids, after the results are dropped, takes one of the three over
    del res, handles
    for ew in jobs:
        ew.maxent_result = None
    n_before = len(BatchSolver._pooled)
    ew = make(2, 100, 200, 20, 20)
    ew.run()
    assert len(BatchSolver._pooled) == n_before


def test_poorman_runs_to_the_end_in_run_async():
    tau, omega, K, Gmat, _ = synthetic.matrix_G(2, 100, 200, noise_seed=5)
    pm = mx.PoormanMaxEnt(use_hermiticity=True)
    pm.set_verbosity(mx.VerbosityFlags.Quiet)
    pm.set_G_tau_data(tau, Gmat)
    pm.omega = omega
    pm.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=20)
    pm.set_error(synthetic.SIGMA)
    h = pm.run_async()
    assert h.done and h.result() is pm.maxent_result and np.all(np.isfinite(h.result().A_out))


@pytest.mark.parametrize('herm', [False, True])
def test_fields_served_from_the_launch_arrays_are_those_assembled_from_records(herm):
    """``run()`` on array input hands the result ONE object per launch (DeferredLaunch); A_out, chi2, S, Q, H, A, n_iter,
    converged, alpha are served from the launch's arrays.  Looking at anything per element settles the launch -- the records and
    analyses ``run()`` always built -- and the same fields assembled from those must be the same bits."""
    tau, omega, K, Gmat, _ = synthetic.matrix_G(3, 100, 200, noise_seed=77)
    ew = mx.ElementwiseMaxEnt(use_hermiticity=herm)
    ew.set_verbosity(mx.VerbosityFlags.Quiet)
    ew.set_G_tau_data(tau, Gmat)
    ew.omega = omega
    ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=25)
    ew.set_error(synthetic.SIGMA)
    res = ew.run()
    assert len(res.__dict__['_deferred']) == 1 and not res.__dict__['_records_store']
    if herm:
        assert res._whole() is None              # (only the upper triangle was solved: the mirror is made the general way)
    else:
        assert res._whole() is not None
    names = ('alpha', 'chi2', 'S', 'Q', 'n_iter', 'converged', 'H', 'A', 'A_out')
    fast = {n: np.array(getattr(res, n)) for n in names}
    fast_other = {a: np.array(res.get_A_out(a)) for a in ('LineFitAnalyzer', 'Chi2CurvatureAnalyzer', 'EntropyAnalyzer')}
    if not herm:
        assert len(res.__dict__['_deferred']) == 1          # (none of that built a record)
    picks = res.analyzer_results[0][1]['LineFitAnalyzer']['alpha_index']          # per element: settles
    assert not res.__dict__['_deferred'] and len(res.__dict__['_records_store']) == (6 if herm else 9)
    assert isinstance(picks, (int, np.integer))
    res._cache = dict()
    for n in names:
        slow = np.array(getattr(res, n))
        assert slow.shape == fast[n].shape and slow.dtype == fast[n].dtype, n
        assert np.array_equal(slow, fast[n], equal_nan=True), n
    for a, val in fast_other.items():
        assert np.array_equal(np.array(res.get_A_out(a)), val, equal_nan=True), a
    assert res.omega is ew.omega or np.array_equal(np.asarray(res.omega), np.asarray(ew.omega))
    # a second run into the same result object: everything the general way
    res2 = ew.run()
    assert res2 is res and not res.__dict__['_deferred']
    assert np.allclose(np.array(res.chi2), fast['chi2'], rtol=1e-9, equal_nan=True)


def test_new_data_on_the_same_object_updates_only_the_data_on_the_device(monkeypatch):
    """set_G_tau_data again (every iteration of a self-consistency loop): mxe_elements_update_data -- the chains stay staged --
    and the answers are those of a fresh object on those data, to the bit: the projection of the new data runs on the device
    (project_kernel) with the operations of the host's in the host's order, no fused multiply-adds"""
    tau, omega, K, G1, _ = synthetic.matrix_G(3, 100, 200, noise_seed=1)
    _, _, _, G2, _ = synthetic.matrix_G(3, 100, 200, noise_seed=2)

    def make(G):
        ew = mx.ElementwiseMaxEnt(use_hermiticity=False)
        ew.set_verbosity(mx.VerbosityFlags.Quiet)
        ew.set_G_tau_data(tau, G)
        ew.omega = omega
        ew.alpha_mesh = mx.LogAlphaMesh(alpha_min=1e-2, alpha_max=1e4, n_points=25)
        ew.set_error(synthetic.SIGMA)
        return ew
    ew = make(G1)
    r1 = np.array(ew.run().chi2)
    ew.set_G_tau_data(tau, G2)
    r2 = np.array(ew.run().chi2)
    assert np.all(np.isfinite(r2)) and np.linalg.norm(r2 - r1) > 1e-3 * np.linalg.norm(r1)
    ref = np.array(make(G2).run().chi2)
    assert np.array_equal(r2, ref)
    monkeypatch.setenv('MXE_HOST_PROJECTION', '1')      # (the host's projection, threaded: what other layouts of G fall back to)
    ew.set_G_tau_data(tau, G1)
    assert np.array_equal(np.array(ew.run().chi2), r1)
    ew.set_G_tau_data(tau, G2)
    assert np.array_equal(np.array(ew.run().chi2), ref)
