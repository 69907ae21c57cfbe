"""Where the reference's own iterate is NOT at the minimiser: small alphas of the diagonal elements of BASELINE config 4.

For every diagonal element (and a few off-diagonal ones) of the cfg4 batch: the oracle port runs the alpha scan, the
extended-precision polish (oracle/hp_truth.py) is started from ITS iterates at the last alphas of the mesh, and the device's H
is compared with that truth and with the port's H.  TEST INFRASTRUCTURE / measurement script (imports oracle/).

    python tools/small_alpha_truth.py [out.txt]
"""
import os
import sys
import multiprocessing as mp

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROWS = (0, 50, 90, 93, 95, 96, 97, 98, 99)


def one(args):
    c, = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)               # (16 forked workers x a BLAS pool each stall one another)
    except Exception:
        pass
    import bench
    from oracle import ref_numpy as R, hp_truth
    batch = bench.build_batch(16, 200, 500, 100, 0)
    K = batch['K']
    i, j = batch['elems'][c]
    ent = 'normal' if i == j else 'plusminus'
    p = R.Problem(np.array(K.K), K.U, K.S, K.V, batch['Gmat'][i, j], batch['err'], batch['D'], entropy=ent)
    ref = R.alpha_loop(p, batch['omega'].delta, np.asarray(batch['alphas']) / len(batch['tau']))
    rows = {}
    for ia in ROWS:
        info = {}
        _, Ht = hp_truth.polish(p.K, p.G, p.err, p.D, p.V, p.S, batch['alphas'][ia], ref['v'][ia], ent, iters=6, info=info)
        rows[ia] = (Ht, ref['H'][ia], info)
    return c, rows


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    import bench
    batch = bench.build_batch(16, 200, 500, 100, 0)
    chains = [c for c in range(256) if batch['elems'][c][0] == batch['elems'][c][1]] + [1, 17, 100]
    truths = {}
    with mp.get_context('fork').Pool(min(16, os.cpu_count() or 1)) as pool:        # before anything touches the GPU
        for c, rows in pool.imap_unordered(one, [(c,) for c in chains]):
            truths[c] = rows
            print('truth of chain %d done (%d of %d)' % (c, len(truths), len(chains)), flush=True)
    ctx = bench.stage(batch, 0)
    out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'])
    corr = ctx.audit()['corr']
    lines = ['# chain (i,j) alpha_index  polish(damped,iters,converged)  |H_ref-H_truth|/|H_truth|  |H_gpu-H_truth|/|H_truth|  '
             '|H_gpu-H_ref|/|H_ref|  device audit  n_iter']
    worst = 0.0
    for c in chains:
        i, j = batch['elems'][c]
        for ia in ROWS:
            Ht, Hr, info = truths[c][ia]
            n = np.linalg.norm
            e_ref = n(Hr - Ht) / n(Ht)
            e_gpu = n(out['H'][c, ia] - Ht) / n(Ht)
            e_gr = n(out['H'][c, ia] - Hr) / n(Hr)
            worst = max(worst, e_gpu) if np.isfinite(e_gpu) else np.inf
            lines.append('%3d (%2d,%2d) %3d  %d %3d %d  %.3e  %.3e  %.3e  %.2e  %d' % (
                c, i, j, ia, info['damped'], info['iterations'], info['converged'], e_ref, e_gpu, e_gr, corr[c, ia],
                out['n_iter'][c, ia]))
    lines.append('# worst |H_gpu - H_truth| / |H_truth| over the table: %.3e' % worst)
    text = '\n'.join(lines)
    print(text)
    if out_path:
        with open(out_path, 'w') as f:
            f.write(text + '\n')
    ctx.close()


if __name__ == '__main__':
    main()
