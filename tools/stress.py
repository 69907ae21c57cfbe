"""Randomised shapes and alpha meshes through the default schedule, checked with the device audit (exact Newton
correction at the returned v of EVERY problem): convergence, accuracy and the cost of the most expensive alpha.
    python tools/stress.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maxent_amd import device, synthetic, hostprep
import maxent_amd as mx

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = dict(corr=0.0, evals=0)
for case in range(n_cases):
    n_orb = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16]))
    n_tau = int(rng.choice([40, 100, 200]))
    n_omega = int(rng.choice([60, 100, 257, 500]))
    n_alpha = int(rng.choice([3, 8, 20, 50, 100, 150]))
    lo, hi = 10.0 ** rng.uniform(-3, 0), 10.0 ** rng.uniform(2, 5)
    ascending = rng.rand() < 0.2
    sigma = 10.0 ** rng.uniform(-5, -2.5)
    tau, omega, K, Gmat, _ = synthetic.matrix_G(n_orb, n_tau, n_omega, seed=int(rng.randint(1 << 30)))
    Gmat = Gmat + (sigma - synthetic.SIGMA) * 0     # (noise of the generator stays; the error bar varies)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = sigma * np.ones(n_tau)
    alphas = np.array(mx.LogAlphaMesh(alpha_min=lo, alpha_max=hi, n_points=n_alpha)) * n_tau
    if ascending:
        alphas = alphas[::-1].copy()
    elems = [(i, j) for i in range(n_orb) for j in range(n_orb)]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    n = len(elems)
    ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
    t0 = time.perf_counter()
    ctx.upload_chains(np.arange(n), alphas, v0)
    ctx.launch()
    n_fin = ctx.finish()
    out = ctx.fetch(want_v=False, want_H=True)
    dt = time.perf_counter() - t0
    info = ctx.last_launch_info()
    au = ctx.audit()
    conv = out['converged']
    corr = np.where(conv, au['corr'], 0.0)
    line = ('case %2d: %2dx%-2d n_tau %3d n_omega %3d n_s %2d n_alpha %3d alpha %.1e..%.1e%s sigma %.0e | %s | kernel %.2f ms (solve incl. upload / finish / fetch %.1f ms, %d alphas finished in the one-chain layout) | '
            'converged %d/%d, audit max %.1e, evals/alpha max %d, H finite %s' % (
                case, n_orb, n_orb, n_tau, n_omega, len(K.S), n_alpha, lo, hi, ' asc' if ascending else '', sigma,
                info['kernel'].replace('mxe::', ''), ctx.last_kernel_ms(), 1e3 * dt, n_fin, conv.sum(), conv.size, corr.max(),
                out['n_evals'].max(), bool(np.all(np.isfinite(out['H'][conv.astype(bool)])))))
    print(line, flush=True)
    worst['corr'] = max(worst['corr'], corr.max()); worst['evals'] = max(worst['evals'], int(out['n_evals'].max()))
    ctx.close()
print('worst audit correction %.2e, most evaluations for one alpha %d' % (worst['corr'], worst['evals']))
