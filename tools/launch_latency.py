"""host-side duration of mxe_chains_launch and device-side step time, several processes in a row
(looking for the runs in which a step takes 4-5 ms although the kernel takes 1.7)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maxent_amd import device
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts())
for _ in range(3):
    ctx.launch()
ctx.sync()
host = []
ctx.timing_mark()
t0 = time.perf_counter()
for _ in range(20):
    ta = time.perf_counter(); ctx.launch(); host.append(time.perf_counter() - ta)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print('pid %d: host launch call mean %.3f ms max %.3f ms; enqueue loop %.2f ms, + sync %.2f ms; device ms per step %.3f; last kernel %.3f ms'
      % (os.getpid(), 1e3 * np.mean(host), 1e3 * np.max(host), 1e3 * (t1 - t0), 1e3 * (t2 - t1), ctx.ms_since_mark() / 20, ctx.last_kernel_ms()))
