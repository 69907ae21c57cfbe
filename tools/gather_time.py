"""time of one torch.distributed (RCCL) gather / all_gather of a 103 MB double tensor; run under
torch.distributed.run (any number of ranks, one per GPU)"""
import os, time
import torch
import torch.distributed as dist
dist.init_process_group('nccl', device_id=torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0))))
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
n = 25600 * 503
x = torch.randn(n, dtype=torch.float64, device='cuda')
outs = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
flat = torch.empty(world * n, dtype=torch.float64, device='cuda')
for name, fn in (('gather', lambda: dist.gather(x, outs, dst=0)),
                 ('all_gather_into_tensor', lambda: dist.all_gather_into_tensor(flat, x)),
                 ('copy_', lambda: flat[:n].copy_(x))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if rank == 0:
        print('%s: %.3f ms per call (%d ranks, %.1f MB per rank)' % (name, 1e2 * (t1 - t0), world, n * 8 / 1e6))
dist.destroy_process_group()
