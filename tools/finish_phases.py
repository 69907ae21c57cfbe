"""Where the hand-over pass (mxe_chains_finish: one-chain kernel on the alphas the lock-step launch left open) spends its time, for
cases of tools/stress.py:   MAXENT_AMD_LIB=maxent_amd/lib/libmaxent_hip_prof.so python tools/finish_phases.py 25 70 5 17
(stamp build: make -C maxent_amd/csrc prof; shares, not times)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MAXENT_AMD_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'maxent_amd', 'lib', 'libmaxent_hip_prof.so'))
from maxent_amd import device
import stress

want = [int(x) for x in sys.argv[1:]] or [25]
lib = device.load_library()
lib.mxe_prof_fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
names = ['prep', 'gram', 'solve', 'step/norm', 'eval', 'accept', 'output']
for c in stress.cases(max(want) + 1, 7):
    if c['case'] not in want:
        continue
    tau, omega, K, Gmat, D, err, alphas, elems, kinds, v0 = stress.inputs(c)
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    n = len(elems)
    ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
    ctx.upload_chains(np.arange(n), alphas, v0, None)
    ctx.launch(); ctx.sync()
    before = ctx.fetch(want_v=False, want_H=False)
    t0 = time.perf_counter()
    n_fin = ctx.finish()
    dt = time.perf_counter() - t0
    after = ctx.fetch(want_v=False, want_H=False)
    prof = np.zeros((n * len(alphas) + 8192, 8), dtype=np.int64)
    assert lib.mxe_prof_fetch(ctx._h, prof.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
    rows = prof[prof[:, :7].sum(axis=1) > 0][:, :7]           # (the stamp buffer is cleared before the pass: its chains are the rows that are not zero)
    d_it = int(after['n_iter'].sum() - before['n_iter'].sum()); d_ev = int(after['n_evals'].sum() - before['n_evals'].sum())
    tot = rows.sum(axis=1)
    print('case %d: n_s %d n_omega %d, %d alphas handed over in %d chains, finish %.1f ms (stamp build); iterations +%d, evaluations +%d; '
          'longest chain %.3e cycles (%.1f ms at 2.4 GHz) of %.3e in all' % (c['case'], len(K.S), c['n_omega'], n_fin, len(rows), 1e3 * dt, d_it, d_ev, tot.max() if len(tot) else 0, (tot.max() if len(tot) else 0) / 2.4e6, tot.sum()))
    if len(rows):
        long = rows[np.argmax(tot)]
        print('   ' + '  '.join('%s %.1f %%' % (nm, 100.0 * rows[:, q].sum() / tot.sum()) for q, nm in enumerate(names)))
        print('   longest chain: ' + '  '.join('%s %.1f %%' % (nm, 100.0 * long[q] / long.sum()) for q, nm in enumerate(names)))
        print('   cycles per iteration %.0f, per evaluation %.0f' % (tot.sum() / max(d_it, 1), tot.sum() / max(d_ev, 1)))
    ctx.close()
