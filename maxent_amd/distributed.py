"""Multi-GPU driver: one process per GPU, elements sharded across ranks.

The (element, alpha) problems are independent given the shared SVD of the
kernel, so the element-wise batch shards with no exchange during the solve
(SURVEY.md 8e).  Elements are dealt round-robin (element e -> rank e mod N),
which interleaves the cheaper plus-minus off-diagonals with the diagonals;
every rank stages U/S/V itself (0.3 MB).  The only collective is the gather
of the per-alpha results, done once after the solve with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU node, "gloo"
in the CPU tests).  Nothing here touches the solver; the local solve is the
same :func:`maxent_amd.maxent_loop.solve_elements`.
"""

import numpy as np


def shard_indices(n_items, world_size, rank):
    """indices of the items rank ``rank`` owns (round-robin)."""
    return list(range(rank, n_items, world_size))


def _dist():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist


def solve_elements_sharded(K, specs, minimizer, solve_fn=None, device_id=None,
                           gather_to_all=True):
    """Shard ``specs`` over the initialised process group, solve the local
    shard on the local GPU and gather the results.

    Returns the same ``(results, info)`` as ``solve_elements`` with the
    results of ALL elements, in the order of ``specs`` (on every rank if
    ``gather_to_all`` else only on rank 0, where other ranks get ``None``).
    With no process group it is a plain local solve.
    """
    if solve_fn is None:
        from .maxent_loop import solve_elements as solve_fn
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return solve_fn(K, specs, minimizer, device_id=device_id or 0)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = shard_indices(len(specs), world, rank)
    if device_id is None:
        device_id = rank if not torch.cuda.is_available() \
            else rank % max(torch.cuda.device_count(), 1)
    local, info = solve_fn(K, [specs[i] for i in mine], minimizer,
                           device_id=device_id)
    use_cuda = dist.get_backend() == 'nccl'
    dev = torch.device('cuda', device_id) if use_cuda else torch.device('cpu')
    n_alpha = len(specs[0]['alpha'])
    n_omega = len(specs[0]['D'])
    n_s = len(specs[0]['v0'])
    per = (len(specs) + world - 1) // world          # padded shard size
    width = n_omega + n_s + 6

    # one packed float64 buffer per rank: [per][n_alpha][H | v | chi2 S Q n_iter conv n_evals]
    buf = np.zeros((per, n_alpha, width))
    for n, r in enumerate(local):
        buf[n, :, :n_omega] = r['H']
        buf[n, :, n_omega:n_omega + n_s] = r['v']
        buf[n, :, n_omega + n_s + 0] = r['chi2']
        buf[n, :, n_omega + n_s + 1] = r['S']
        buf[n, :, n_omega + n_s + 2] = r['Q']
        buf[n, :, n_omega + n_s + 3] = r['n_iter']
        buf[n, :, n_omega + n_s + 4] = r['converged']
        buf[n, :, n_omega + n_s + 5] = r['n_evals']
    t = torch.from_numpy(buf).to(dev)
    if gather_to_all:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
    else:
        parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, parts, dst=0)
        if rank != 0:
            return None, info
    out = [None] * len(specs)
    for r in range(world):
        block = parts[r].cpu().numpy()
        for n, i in enumerate(shard_indices(len(specs), world, r)):
            b = block[n]
            out[i] = dict(alpha=np.asarray(specs[i]['alpha'], dtype=float), A=None,
                          H=b[:, :n_omega].copy(),
                          v=b[:, n_omega:n_omega + n_s].copy(),
                          chi2=b[:, n_omega + n_s + 0].copy(),
                          S=b[:, n_omega + n_s + 1].copy(),
                          Q=b[:, n_omega + n_s + 2].copy(),
                          n_iter=b[:, n_omega + n_s + 3].astype(np.int32),
                          converged=b[:, n_omega + n_s + 4].astype(bool),
                          n_evals=b[:, n_omega + n_s + 5].astype(np.int32))
    return out, info
