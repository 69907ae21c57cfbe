"""world_size-2 CPU test of the multi-GPU path (gloo): element sharding and the
result gather of maxent_amd.distributed, with the device solve replaced by a
deterministic stand-in (no GPU here)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_solve(K, specs, minimizer, device_id=0, **kw):
    """stands in for solve_elements: results are a pure function of the spec."""
    out = []
    for s in specs:
        a = np.asarray(s['alpha'], dtype=float)
        tag = float(np.sum(s['G']))
        H = np.outer(1.0 / (1.0 + a), np.asarray(s['D'])) * (1 + tag)
        out.append(dict(alpha=a, v=np.outer(a, s['v0']), H=H, chi2=a * tag,
                        S=-a, Q=a * 2, n_iter=np.full(len(a), 3, dtype=np.int32),
                        converged=np.ones(len(a), dtype=bool),
                        n_evals=np.full(len(a), 4, dtype=np.int32)))
    return out, dict(kernel_ms=0.0, device=device_id)


def _specs(n_elem, n_alpha=5, n_omega=7, n_s=3, n_tau=4):
    rng = np.random.RandomState(0)
    return [dict(G=rng.randn(n_tau), err=np.ones(n_tau), D=rng.rand(n_omega),
                 kind=e % 2, v0=rng.randn(n_s),
                 alpha=np.logspace(2, 0, n_alpha)) for e in range(n_elem)]


def _worker(rank, world, port, n_elem, to_all, q):
    sys.path.insert(0, ROOT)
    from maxent_amd import distributed as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    specs = _specs(n_elem)
    res, info = D.solve_elements_sharded(None, specs, None, solve_fn=_fake_solve,
                                         gather_to_all=to_all)
    ref, _ = _fake_solve(None, specs, None)
    ok = True
    if res is None:
        ok = (rank != 0 and not to_all)
    else:
        for a, b in zip(res, ref):
            for k in ('H', 'v', 'chi2', 'S', 'Q', 'n_iter', 'converged', 'n_evals', 'alpha'):
                ok = ok and np.array_equal(np.asarray(a[k]), np.asarray(b[k]))
    q.put((rank, ok, info['device']))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n_elem, to_all):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_elem, to_all, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(got)


def test_shard_indices_cover_everything_once():
    sys.path.insert(0, ROOT)
    from maxent_amd.distributed import shard_indices
    for n, w in ((7, 2), (16, 8), (3, 4), (256, 8)):
        seen = sorted(i for r in range(w) for i in shard_indices(n, w, r))
        assert seen == list(range(n))
    assert shard_indices(256, 8, 3)[:3] == [3, 11, 19]


def test_two_ranks_allgather_odd_element_count():
    got = _run(7, True)
    assert got == [(0, True, 0), (1, True, 1)]


def test_two_ranks_gather_to_rank0():
    got = _run(4, False)
    assert [g[1] for g in got] == [True, True]
