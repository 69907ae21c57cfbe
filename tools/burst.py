"""a BURST of n jobs (one launch per context, enqueued back to back, then waited for) against the steady rate of bench.py's region
with launches in flight: what fill and drain cost a caller that has n jobs and no more"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
mine = list(range(256))
for n, cut in ((4, 4), (4, 2), (4, 1), (2, 2), (2, 1), (8, 8), (8, 4), (1, 1)):
    opts = device.default_opts(in_flight=cut)
    lanes = []
    for _ in range(n):
        c = bench.stage(batch, 0, mine)
        c.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], opts)
        lanes.append(c)
    for c in lanes:
        c.launch(); c.select_launch(0)
    for c in lanes:
        c.sync()
    best, gap_best = 1e9, 1e9
    for rep in range(30):
        t0 = time.perf_counter()
        for c in lanes:
            c.launch(); c.select_launch(0)
        for c in lanes:
            c.sync()
        best = min(best, time.perf_counter() - t0)
    for rep in range(30):            # with an idle gap in front (clocks)
        time.sleep(0.003)
        t0 = time.perf_counter()
        for c in lanes:
            c.launch(); c.select_launch(0)
        for c in lanes:
            c.sync()
        gap_best = min(gap_best, time.perf_counter() - t0)
    # steady: 50 rounds back to back
    t0 = time.perf_counter()
    for rep in range(50):
        for c in lanes:
            c.launch(); c.select_launch(0)
    for c in lanes:
        c.sync()
    steady = (time.perf_counter() - t0) / (50 * n)
    print('%d jobs, cut for %d in flight: burst %.3f ms (%.3f per job; after a 3 ms pause %.3f), steady %.3f ms per job, kernel alone %.3f ms' % (
        n, cut, 1e3 * best, 1e3 * best / n, 1e3 * gap_best, 1e3 * steady, lanes[0].last_kernel_ms()), flush=True)
    for c in lanes:
        c.close()
