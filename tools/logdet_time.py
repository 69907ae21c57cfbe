"""time of mxe_logdet on the cfg4 batch (25 600 problems) next to the host evaluation it replaces"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
out = ctx.solve_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], want_v=False)
ctx.logdet()
t0 = time.perf_counter(); ld = ctx.logdet(); t1 = time.perf_counter()
print('mxe_logdet, 25600 problems, incl. D2H of the result: %.2f ms' % (1e3 * (t1 - t0)))
K = batch['K']
M = np.diag((K.S / batch['err'][0]) ** 2)
t0 = time.perf_counter()
n = 0
for c in (0, 1):
    for ia in range(100):
        H = out['H'][c, ia]
        w = H if batch['kinds'][c] == 0 else np.sqrt(H * H + 4 * batch['D'] ** 2)
        ref = np.linalg.slogdet(np.eye(len(K.S)) + M @ ((K.V.T * w) @ K.V) / batch['alphas'][ia])[1]
        assert abs(ref - ld[c, ia]) < 1e-8 * max(1, abs(ref))
        n += 1
t1 = time.perf_counter()
print('host numpy (n_s x n_s form): %.3f ms per problem -> %.1f s for 25600' % (1e3 * (t1 - t0) / n, (t1 - t0) / n * 25600))
