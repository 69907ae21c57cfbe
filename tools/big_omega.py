"""how far n_omega goes: kernel chosen, time, convergence, audit (or the error the library gives)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maxent_amd as mx
from maxent_amd import synthetic, device, hostprep
for n_omega in (2000, 3000, 5000, 7000):
    for n_orb in (1, 16):
        try:
            tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, 100, n_omega)
            K.reduce_singular_space(1e-14)
            D = synthetic.flat_D(omega)
            err = synthetic.SIGMA * np.ones(100)
            alphas = np.array(synthetic.alpha_mesh(20)) * 100
            elems = [(i, j) for i in range(n_orb) for j in range(n_orb)]
            kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
            v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
            ctx = device.DeviceContext(K.U, K.S, K.V)
            ds = ctx.add_dataset(err)
            n = len(elems)
            ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
            out = ctx.solve_chains(np.arange(n), alphas, v0, want_v=False, want_H=False)
            au = ctx.audit()
            print('n_omega %5d, %3d scans: %s, %.2f ms, converged %d/%d, audit max %.1e' % (
                n_omega, n, ctx.last_launch_info()['kernel'], ctx.last_kernel_ms(), out['converged'].sum(), out['converged'].size, au['corr'].max()), flush=True)
            ctx.close()
        except Exception as e:
            print('n_omega %5d, %3d scans: %s: %s' % (n_omega, n_orb * n_orb, type(e).__name__, str(e)[:150]), flush=True)
