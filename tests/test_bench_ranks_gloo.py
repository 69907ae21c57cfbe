"""The rank choreography of ``bench.py --gpus N`` with two REAL processes on the CPU (gloo): the communicator hand-shake through a
file (one per context in flight), the settling passes whose number the ranks agree on by ONE all-reduce per pass -- every step ends
with a gather that pairs the ranks up: a rank that decided on its own clock would leave its peer waiting --, the timed region with
four contexts in flight and the maximum over the ranks.  The device is a stand-in whose gather and all-reduce are torch.distributed
calls (they block until both ranks have issued them, like the RCCL ones); torch.distributed is test plumbing here."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _rank(rank, port, q):
    try:
        import torch
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=WORLD)
        import bench
        from maxent_amd import device
        log = []

        class FakeCtx(object):
            made = 0

            def __init__(self):
                FakeCtx.made += 1
                self.uid, self.gathers, self.launches = None, 0, 0

            def upload_chains(self, elems, alphas, v0, opts):
                self.opts = opts

            def comm_init(self, world, r, uid):
                self.uid = bytes(uid)

            def comm_set_loopback(self, on):
                raise AssertionError('two ranks: no loop-back')

            def allreduce(self, values, op='sum'):
                t = torch.tensor(np.atleast_1d(values).astype(float))
                dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 'sum' else dist.ReduceOp.MAX)
                return t.numpy()

            def launch(self):
                self.launches += 1

            def select_launch(self, deg):
                pass

            def gather(self, root, counts, full=False, recv=None):
                # (blocks until every rank has issued it, as the send / recv pairs of mxe_gather do)
                send = torch.full((4,), float(self.gathers))
                out = [torch.zeros(4) for _ in range(WORLD)] if rank == root else None
                dist.gather(send, out, dst=root)
                if rank == root:
                    assert all(float(o[0]) == self.gathers for o in out), 'the ranks are not at the same step of this context'
                self.gathers += 1

            def sync(self):
                pass

            def finish(self):
                return 0

            def fetch(self, want_v=True, want_H=True):
                return dict(converged=np.ones((3, 5), dtype=np.int32), n_evals=np.full((3, 5), 2, dtype=np.int32))

            def audit(self):
                return dict(corr=np.full((3, 5), 1e-9))

            def last_launch_info(self):
                return dict(kernel='fake', n_workgroups=1)

            def comm_destroy(self):
                log.append('destroyed')

            def close(self):
                log.append('closed')

        bench.stage = lambda batch, dev, which=None: FakeCtx()
        device.comm_unique_id = lambda: os.urandom(128)
        device.default_opts = lambda **kw: dict(kw)

        class Args(object):
            waves_per_chain = chains_per_wg = alpha_split = wg_per_cu = 0
            warmup, steps = 3, 37 + 0 * rank
        # the rank with the slow clock: its own clock would end the settling passes earlier
        if rank == 1:
            real = bench.time.perf_counter
            bench.time.perf_counter = lambda: real() * 1.7
        batch = dict(alphas=np.ones(5), v0=np.zeros((3, 4)))
        elapsed, check, t_comm = bench.in_flight_comm_region(batch, [0, 1, 2], 0, rank, WORLD, [10, 10], False, Args, 4)
        assert t_comm > 0                      # (the four communicators are made one after the other BEFORE any launch, timed apart)
        assert FakeCtx.made == 4 and log.count('destroyed') == 4 and log.count('closed') == 4
        assert elapsed > 0 and ((check is not None and check['converged'] == 15) if rank == 0 else check is None)
        # the hand-shake of ONE context (the first region of bench.py) with its own file, behind the four above
        c = FakeCtx()
        bench.comm_setup(c, rank, WORLD)
        uids = [None, None]
        dist.all_gather_object(uids, c.uid)
        assert uids[0] == uids[1] and len(uids[0]) == 128
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as exc:          # pragma: no cover
        import traceback
        q.put((rank, 'failed: %r\n%s' % (exc, traceback.format_exc())))


def test_two_ranks_agree_on_every_pass_of_the_region_with_steps_in_flight():
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
