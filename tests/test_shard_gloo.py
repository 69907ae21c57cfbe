"""The N > 1 host path with two REAL processes on the CPU (world_size 2, gloo): every rank derives its shard from
``mxe_shard_plan`` (the C entry the devices are driven through), packs its elements' results in the layout of the
library's compact result pack, rank 0 receives the packs in rank order -- what ``mxe_gather`` delivers over RCCL on
the devices -- and takes them apart with the product's own unpacking code.  torch.distributed is test plumbing
here; the package and bench.py do not import it (tests/test_gpu_multi.py runs the gather itself on a device)."""
import multiprocessing as mp
import socket

import numpy as np

N_ALPHA, N_OMEGA, N_ELEM, WORLD = 6, 9, 7, 2


def _truth():
    rng = np.random.RandomState(42)
    return dict(chi2=rng.rand(N_ELEM, N_ALPHA), S=rng.rand(N_ELEM, N_ALPHA), Q=rng.rand(N_ELEM, N_ALPHA),
                H=rng.rand(N_ELEM, N_OMEGA), idx=rng.randint(0, N_ALPHA, N_ELEM))


def _rank(rank, port, q):
    try:
        import torch
        import torch.distributed as dist
        from maxent_amd import device
        from maxent_amd.batch_solver import BatchSolver
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=WORLD)
        rank_of, local_of, n_local = device.shard_plan(N_ELEM, WORLD)
        t = _truth()
        mine = [e for e in range(N_ELEM) if rank_of[e] == rank]
        assert [int(local_of[e]) for e in mine] == list(range(len(mine)))
        pack = np.concatenate([t['chi2'][mine].ravel(), t['S'][mine].ravel(), t['Q'][mine].ravel(),
                               t['H'][mine].ravel(), t['idx'][mine].astype(float)])
        counts = [3 * int(n) * N_ALPHA + int(n) * (N_OMEGA + 1) for n in n_local]      # bench.py: per()
        assert len(pack) == counts[rank]
        width = max(counts)
        send = torch.zeros(width, dtype=torch.float64)
        send[:len(pack)] = torch.from_numpy(pack)
        recv = [torch.zeros(width, dtype=torch.float64) for _ in range(WORLD)] if rank == 0 else None
        dist.gather(send, recv, dst=0)
        # the timing reduction of bench.py: maximum over the ranks
        el = torch.tensor([0.5 + rank], dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        assert float(el[0]) == 0.5 + WORLD - 1
        if rank == 0:
            class Shell(BatchSolver):         # the unpacking arithmetic without device contexts
                def __init__(self):
                    self.n_omega = N_OMEGA
            sh = Shell()
            for r in range(WORLD):
                out = sh._unpack_compact(recv[r][:counts[r]].numpy(), int(n_local[r]), N_ALPHA)
                for e in range(N_ELEM):
                    if rank_of[e] == r:
                        c = int(local_of[e])
                        assert np.array_equal(out['chi2'][c], t['chi2'][e]) and np.array_equal(out['S'][c], t['S'][e])
                        assert np.array_equal(out['Q'][c], t['Q'][e]) and np.array_equal(out['linefit_H'][c], t['H'][e])
                        assert out['linefit_index'][c] == t['idx'][e]
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as exc:          # pragma: no cover
        q.put((rank, 'failed: %r' % (exc,)))


def test_two_processes_shard_pack_gather_unpack():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
    assert results == {0: 'ok', 1: 'ok'}, results
