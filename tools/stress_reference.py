#!/usr/bin/env python3
"""The reference's algorithm on the hard cases of tools/stress.py: for every case of a device
run (``python tools/stress.py 100 7 gpurun_out/.../stress_device.npz``) in which the device left alphas unconverged or
needed more than 300 evaluations for one, the scan of the worst element is solved with oracle/ref_numpy.py -- the
step-faithful port of LevenbergMinimizer + MaxEntCostFunction with the reference's defaults (maxiter 1000,
max|dQ| < 1e-4 or relative change < 1e-16) -- and its per-alpha iteration counts and convergence flags are written to
a fixture.  CPU only (build container), one process per case.

    python tools/stress_reference.py gpurun_out/r03/stress_device.npz tests/golden/stress_reference.npz [n_workers]
"""
import os
import sys
import time
import multiprocessing as mp

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def solve(job):
    case, elem = job
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    import stress
    from maxent_amd import device
    from oracle import ref_numpy as R
    c = [x for x in stress.cases(case + 1, SEED) if x['case'] == case][0]
    tau, omega, K, Gmat, D, err, alphas, elems, kinds, v0 = stress.inputs(c)
    i, j = elems[elem]
    ent = 'normal' if kinds[elem] == device.ENTROPY_NORMAL else 'plusminus'
    p = R.Problem(np.array(K.K), K.U, K.S, K.V, Gmat[i, j], err, D, entropy=ent)
    t0 = time.perf_counter()
    out = R.alpha_loop(p, omega.delta, alphas / c['n_tau'])
    return case, elem, out['n_iter'], out['converged'], time.perf_counter() - t0


def main():
    global SEED
    dev = np.load(sys.argv[1])
    dst = sys.argv[2]
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    SEED = int(dev['seed'])
    jobs = []
    for case in range(int(dev['n_cases'])):
        conv, ne = dev['case%d_converged' % case], dev['case%d_n_evals' % case]
        bad = (conv == 0).sum(axis=1) * 10000 + ne.max(axis=1)
        if (conv == 0).any() or ne.max() > 300:
            jobs.append((case, int(np.argmax(bad))))
    print('%d cases with unconverged alphas or > 300 evaluations; worst element of each through the oracle port' % len(jobs), flush=True)
    rec = dict(seed=SEED, cases=np.array([j[0] for j in jobs]), elements=np.array([j[1] for j in jobs]))
    with mp.get_context('fork').Pool(workers) as pool:
        for case, elem, n_iter, conv, dt in pool.imap_unordered(solve, jobs):
            dconv, dne = dev['case%d_converged' % case][elem], dev['case%d_n_evals' % case][elem]
            rec['case%d_ref_n_iter' % case] = n_iter.astype(np.int32)
            rec['case%d_ref_converged' % case] = conv.astype(np.int8)
            rec['case%d_dev_converged' % case] = dconv
            rec['case%d_dev_n_evals' % case] = dne
            print('case %2d element %3d: reference unconverged %3d / %3d (max %4d iterations, %5.1f s)   device unconverged %3d (max %4d evaluations)'
                  % (case, elem, int((~conv).sum()), len(conv), int(n_iter.max()), dt, int((dconv == 0).sum()), int(dne.max())), flush=True)
    np.savez_compressed(dst, **rec)
    ref_u = sum(int((rec['case%d_ref_converged' % c] == 0).sum()) for c in rec['cases'])
    dev_u = sum(int((rec['case%d_dev_converged' % c] == 0).sum()) for c in rec['cases'])
    print('on these %d scans: reference unconverged %d, device unconverged %d' % (len(jobs), ref_u, dev_u))


if __name__ == '__main__':
    main()
