"""kernel time of small batches (1, 2, 4, 16 alpha scans of 100 alphas, n_tau 200, n_omega 500; 1 = BASELINE config 2) in
the default (lock-step) layout and in the one-chain layout"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maxent_amd import device, synthetic, hostprep
for n_scan in (1, 2, 4, 16):
    tau, omega, K, Gmat, _ = synthetic.matrix_G(4, 200, 500)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = synthetic.SIGMA * np.ones(200)
    alphas = np.array(synthetic.alpha_mesh(100)) * 200
    elems = [(i, i) for i in range(4)] + [(i, j) for i in range(4) for j in range(4) if i != j]
    elems = elems[:n_scan]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    n = len(elems)
    ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
    for split, cpw in ((0, 0), (0, 1)):
        ms = []
        for rep in range(4):
            out = ctx.solve_chains(np.arange(n), alphas, v0, device.default_opts(alpha_split=split, chains_per_wg=cpw), want_v=False, want_H=False)
            ms.append(ctx.last_kernel_ms())
        print('%2d scan(s) x 100 alpha, alpha_split %d cpw %d: %s, kernel %.3f ms (best of 3), evals %d, converged %d/%d, workgroups %d' % (
            n, split, cpw, ctx.last_launch_info()['kernel'], min(ms[1:]), out['n_evals'].sum(), out['converged'].sum(), out['converged'].size,
            ctx.last_launch_info()['n_workgroups']), flush=True)
    ctx.close()
