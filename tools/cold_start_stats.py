import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
for split in (4, 8):
    out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'],
                           device.default_opts(chains_per_wg=4, alpha_split=split), want_v=False, want_H=False)
    L = 100 // split
    starts = [int(100*s/split) for s in range(split)]
    ni = out['n_iter']
    print('split', split, 'kernel ms', ctx.last_kernel_ms())
    for s0 in starts:
        col = ni[:, s0]
        print('  piece start alpha idx %3d: cold-start iters mean %.1f max %d (elem %d kind %d) p99 %.0f' % (s0, col.mean(), col.max(), col.argmax(), batch['kinds'][col.argmax()], np.percentile(col, 99)))
    tot = np.array([[ni[c, int(100*s/split):int(100*(s+1)/split)].sum() for s in range(split)] for c in range(256)])
    print('  per-piece total iterations: mean %.1f max %d ; nevals max per problem %d' % (tot.mean(), tot.max(), out['n_evals'].max()))
    big = np.argwhere(ni > 60)
    print('  problems with >60 iterations:', len(big), big[:10].tolist())
