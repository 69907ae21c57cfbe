"""Coarse alpha meshes (3-8 alphas over 4-6 decades: the reference's own tests and defaults): where the rounds go.
smoke()'s launch (2 scans x 8 alphas, n_omega 120), the reference's test mesh (LogAlphaMesh(alpha_min=0.08, n_points=5), test/python/
tau_maxent.py:44) on the cfg2 grids, and the coarse cases of tools/stress.py: kernel ms, rounds of the deepest slot, evaluations per alpha.
    python tools/coarse_mesh.py [case numbers of tools/stress.py 100 7 ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import maxent_amd as mx
from maxent_amd import device, synthetic, hostprep
import stress


def run(tag, K, err, Gs, D, kinds, alphas, omega, opts=None):
    ctx = device.DeviceContext(K.U, K.S, K.V)
    ds = ctx.add_dataset(err)
    n = len(Gs)
    ctx.set_elements([ds] * n, list(Gs), np.tile(D, (n, 1)), kinds)
    v0 = np.stack([hostprep.initial_v(K.V, D, omega.delta, k) for k in kinds])
    ctx.upload_chains(np.arange(n), alphas, v0, opts)
    ctx.launch(); ctx.sync()
    ms = []
    for _ in range(5):
        ctx.launch(); ctx.sync(); ms.append(ctx.last_kernel_ms())
    left = ctx.finish()
    out = ctx.fetch(want_v=False, want_H=False)
    info = ctx.last_launch_info()
    depth = ctx.launch_depth()
    a = ctx.audit()['corr']
    ev = out['n_evals']
    nk = np.array(kinds)
    def per(kind):
        sel = nk == kind
        return ' '.join('%d' % x for x in ev[sel].max(axis=0)) if sel.any() else '-'
    print('%-28s %s wg %d: kernel %.3f ms, depth %s rounds, left to finish %d, converged %d/%d, audit max %.1e\n'
          '    evaluations per alpha (max over scans) normal: %s\n    plus-minus: %s\n    alphas: %s' % (
              tag, info['kernel'].replace('mxe::', ''), info['n_workgroups'], min(ms), depth['max_rounds'], left, int(out['converged'].sum()), out['converged'].size,
              np.nanmax(np.where(out['converged'], a, 0)), per(device.ENTROPY_NORMAL), per(device.ENTROPY_PLUSMINUS),
              ' '.join('%.2g' % x for x in np.atleast_2d(alphas)[0])), flush=True)
    ctx.close()


def main():
    # smoke()
    tau, omega, K, G = synthetic.single_G(60, 120)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    run('smoke (2 x 8 alphas)', K, synthetic.SIGMA * np.ones(60), [G, G], D, [device.ENTROPY_NORMAL, device.ENTROPY_PLUSMINUS],
        np.array(synthetic.alpha_mesh(8)) * 60, omega)
    # the reference's own test mesh on the cfg2 grids
    tau, omega, K, G = synthetic.single_G(200, 500)
    K.reduce_singular_space(1e-14)
    D = synthetic.flat_D(omega)
    for n_pts, amin in ((5, 0.08), (20, 1e-4)):
        al = np.array(mx.LogAlphaMesh(alpha_min=amin, n_points=n_pts) if amin != 1e-4 else mx.LogAlphaMesh()) * 200
        run('LogAlphaMesh(%g, n=%d)' % (amin, n_pts), K, synthetic.SIGMA * np.ones(200), [G], D, [device.ENTROPY_NORMAL], al, omega)
    want = [int(x) for x in sys.argv[1:]] or [1, 4, 8, 19, 22, 27, 92]
    for c in stress.cases(100, 7):
        if c['case'] in want:
            tau, omega, K, Gmat, D, err, alphas, elems, kinds, v0 = stress.inputs(c)
            run('stress case %d (%dx%d, %d alphas)' % (c['case'], c['n_orb'], c['n_orb'], c['n_alpha']), K, err,
                [Gmat[i, j] for i, j in elems], D, kinds, alphas, omega)


if __name__ == '__main__':
    main()
