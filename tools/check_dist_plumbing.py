"""one-rank check of the device-buffer -> torch -> RCCL gather plumbing used by bench.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from maxent_amd import device
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
batch = bench.build_batch(16, 200, 500, 100, 0)
ctx = bench.stage(batch, 0)
ctx.upload_chains(np.arange(256, dtype=np.int32), batch['alphas'], batch['v0'], device.default_opts())
ptrs = ctx.result_device_ptrs()
P = 25600
tH = torch.as_tensor(bench._DevArray(ptrs['H'], (P, 500), '<f8'), device='cuda')
print('tensor data_ptr == library pointer:', tH.data_ptr() == ptrs['H'])
ctx.launch(); ctx.sync()
out = ctx.fetch(want_v=False)
print('zero-copy view sees the new results:', np.array_equal(tH.cpu().numpy().reshape(256, 100, 500), out['H']))
gH = [torch.empty_like(tH)]
for name, fn in (('gather 102 MB', lambda: dist.gather(tH, gH, dst=0)), ('barrier', lambda: dist.barrier()),
                 ('all_gather 102 MB', lambda: dist.all_gather(gH, tH))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print('%-18s %.3f ms' % (name, (time.perf_counter() - t0) / 5 * 1e3))
t0 = time.perf_counter()
for _ in range(5):
    ctx.launch(); ctx.sync()
print('launch+sync        %.3f ms (kernel %.3f)' % ((time.perf_counter() - t0) / 5 * 1e3, ctx.last_kernel_ms()))
dist.destroy_process_group()
