"""Randomised shapes and alpha meshes through the default schedule, checked with the device audit (exact Newton
correction at the returned v of EVERY problem): convergence, accuracy and the cost of the most expensive alpha.
    python tools/stress.py [n_cases] [seed]        (STRESS_F32=1: the binary32 launches, mxe_opts.precision)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maxent_amd import device, synthetic, hostprep
import maxent_amd as mx

def cases(n_cases, seed=7):
    """the random cases of this tool, reproducibly (tools/stress_reference.py replays them for the oracle port)"""
    rng = np.random.RandomState(seed)
    for case in range(n_cases):
        n_orb = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16]))
        n_tau = int(rng.choice([40, 100, 200]))
        n_omega = int(rng.choice([60, 100, 257, 500]))
        n_alpha = int(rng.choice([3, 8, 20, 50, 100, 150]))
        lo, hi = 10.0 ** rng.uniform(-3, 0), 10.0 ** rng.uniform(2, 5)
        ascending = bool(rng.rand() < 0.2)
        sigma = 10.0 ** rng.uniform(float(os.environ.get('STRESS_LOG_SIGMA_MIN', '-5')), float(os.environ.get('STRESS_LOG_SIGMA_MAX', '-2.5')))     # (defaults: the range of the recorded runs)
        yield dict(case=case, n_orb=n_orb, n_tau=n_tau, n_omega=n_omega, n_alpha=n_alpha, lo=lo, hi=hi,
                   ascending=ascending, sigma=sigma, seed=int(rng.randint(1 << 30)))


def inputs(c):
    """(tau, omega, K, Gmat, D, err, alphas, elems, kinds, v0) of a case"""
    tau, omega, K, Gmat, _ = synthetic.matrix_G(c['n_orb'], c['n_tau'], c['n_omega'], seed=c['seed'])
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    err = c['sigma'] * np.ones(c['n_tau'])          # (the noise of the generator stays 1e-4; the error bar varies)
    alphas = np.array(mx.LogAlphaMesh(alpha_min=c['lo'], alpha_max=c['hi'], n_points=c['n_alpha'])) * c['n_tau']
    if c['ascending']:
        alphas = alphas[::-1].copy()
    elems = [(i, j) for i in range(c['n_orb']) for j in range(c['n_orb'])]
    kinds = [device.ENTROPY_NORMAL if i == j else device.ENTROPY_PLUSMINUS for i, j in elems]
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    return tau, omega, K, Gmat, D, err, alphas, elems, kinds, v0


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dump = sys.argv[3] if len(sys.argv) > 3 else None        # .npz: per case the converged flags and evaluation counts of every alpha
    worst = dict(corr=0.0, evals=0)
    record = {}
    for c in cases(n_cases, seed):
        case, n_orb, n_tau, n_omega, n_alpha, lo, hi, ascending, sigma = (c[k] for k in (
            'case', 'n_orb', 'n_tau', 'n_omega', 'n_alpha', 'lo', 'hi', 'ascending', 'sigma'))
        tau, omega, K, Gmat, D, err, alphas, elems, kinds, v0 = inputs(c)
        ctx = device.DeviceContext(K.U, K.S, K.V)
        ds = ctx.add_dataset(err)
        n = len(elems)
        ctx.set_elements([ds] * n, [Gmat[i, j] for i, j in elems], np.tile(D, (n, 1)), kinds)
        t0 = time.perf_counter()
        ctx.upload_chains(np.arange(n), alphas, v0, device.default_opts(precision=device.PRECISION_F32) if os.environ.get('STRESS_F32') else None)
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
        record['case%d_converged' % case] = conv.astype(np.int8)
        record['case%d_n_evals' % case] = out['n_evals'].astype(np.int32)
        record['case%d_ms' % case] = np.array([ctx.last_kernel_ms(), 1e3 * dt, n_fin])
        ctx.close()
    print('worst audit correction %.2e, most evaluations for one alpha %d' % (worst['corr'], worst['evals']))
    if dump:
        np.savez_compressed(dump, n_cases=n_cases, seed=seed, **record)


if __name__ == '__main__':
    main()
