"""Host side of the multi-GPU path without a GPU: the shard plan comes out of the C entry the devices are
driven through (``mxe_shard_plan``), the result packs are split back by the same arithmetic the gather
uses, and no framework is involved (the gather itself needs devices: tests/test_gpu_multi.py)."""
import numpy as np

from maxent_amd import device
from maxent_amd.batch_solver import BatchSolver


def test_shard_plan_is_round_robin_and_complete():
    for n_elem, n_ranks in ((256, 8), (7, 2), (3, 4), (16, 1), (0, 3)):
        rank_of, local_of, n_local = device.shard_plan(n_elem, n_ranks)
        assert list(rank_of) == [e % n_ranks for e in range(n_elem)]
        assert int(n_local.sum()) == n_elem
        # (rank, local index) is a bijection onto the shards, in element order inside a rank
        seen = set()
        for e in range(n_elem):
            assert local_of[e] == e // n_ranks
            seen.add((int(rank_of[e]), int(local_of[e])))
        assert len(seen) == n_elem
        assert all(n_local[r] == len([e for e in range(n_elem) if e % n_ranks == r]) for r in range(n_ranks))
    # cfg4 on a node: 32 elements per GPU, diagonal (expensive) elements spread over all of them
    rank_of, _, n_local = device.shard_plan(256, 8)
    assert list(n_local) == [32] * 8
    diag = [17 * i for i in range(16)]          # (i, i) of a 16 x 16 matrix in row-major order
    assert sorted(np.bincount(rank_of[diag], minlength=8)) == [2] * 8


def test_two_fake_ranks_pack_and_unpack_through_the_gather_layout():
    """two ranks' compact packs, concatenated in rank order as the root receives them, come apart into
    the per-element arrays of the right elements"""
    n_alpha, n_omega, n_elem, n_ranks = 5, 7, 5, 2
    rank_of, local_of, n_local = device.shard_plan(n_elem, n_ranks)
    rng = np.random.RandomState(0)
    truth = dict(chi2=rng.rand(n_elem, n_alpha), S=rng.rand(n_elem, n_alpha), Q=rng.rand(n_elem, n_alpha),
                 H=rng.rand(n_elem, n_omega), idx=rng.randint(0, n_alpha, n_elem))
    packs = []
    for r in range(n_ranks):
        mine = [e for e in range(n_elem) if rank_of[e] == r]
        packs.append(np.concatenate([truth['chi2'][mine].ravel(), truth['S'][mine].ravel(), truth['Q'][mine].ravel(),
                                     truth['H'][mine].ravel(), truth['idx'][mine].astype(float)]))
        assert len(packs[-1]) == 3 * len(mine) * n_alpha + len(mine) * (n_omega + 1)
    recv = np.concatenate(packs)

    class Shell(BatchSolver):             # the unpacking arithmetic without contexts
        def __init__(self):
            self.n_omega = n_omega
    sh, off = Shell(), 0
    for r in range(n_ranks):
        cnt = len(packs[r])
        out = sh._unpack_compact(recv[off:off + cnt], int(n_local[r]), n_alpha)
        off += cnt
        for e in range(n_elem):
            if rank_of[e] == r:
                c = local_of[e]
                assert np.array_equal(out['chi2'][c], truth['chi2'][e]) and np.array_equal(out['S'][c], truth['S'][e])
                assert np.array_equal(out['Q'][c], truth['Q'][e]) and np.array_equal(out['linefit_H'][c], truth['H'][e])
                assert out['linefit_index'][c] == truth['idx'][e]


def test_package_does_not_import_a_framework():
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = 'import sys; sys.path.insert(0, %r); import maxent_amd, maxent_amd.batch_solver; ' \
           'assert "torch" not in sys.modules, "torch imported"' % root
    subprocess.check_call([sys.executable, '-c', code])
