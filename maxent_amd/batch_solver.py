"""The alpha scans of many matrix elements on one or several GPUs of this process.

This is the host side of SURVEY.md 8(e): the (element, alpha) problems are independent given the
shared SVD of the kernel, so a batch shards by element -- element e of the batch on rank e mod N
(``mxe_shard_plan``), its whole alpha scan on one device --, every device stages U / S / V itself,
solves its shard with the same chain kernel, and ONE gather (``mxe_gather_local``: RCCL send / recv over
xGMI between distinct devices, called from libmaxent_hip.so directly) brings the per-alpha scalars and,
on request, the hidden images to the first device and from there to the host.

Contexts are kept: ``BatchSolver.for_kernel(K, device_ids)`` returns the solver that already holds the
staged basis of ``K`` on those devices (a new one only when the kernel was refilled or re-decomposed),
so that ``ElementwiseMaxEnt.run`` does not pay a hipMalloc / upload / hipFree cycle per phase.

H is large (n_alpha x n_omega per element; 102 MB for a 16 x 16 x 100 x 500 job) and most of it is never
looked at: the scalars chi2, S, Q, n_iter, converged and the vectors v come back eagerly, H of an element
is fetched when it is first asked for (:class:`LazyH`), single rows -- what an analyzer needs -- through
``rows()`` without touching the rest.
"""

import weakref

import numpy as np

from . import device


class LazyH(object):
    """hidden images of one alpha scan, still on the device until somebody looks"""
    _what = 'H'

    def __init__(self, owner, rank, chain, n_alpha, n_omega):
        self._owner, self._rank, self._chain = owner, rank, chain
        self.shape = (n_alpha, n_omega)
        self.dtype = np.dtype(float)
        self.ndim = 2
        self._val = None

    def materialize(self):
        if self._val is None:
            self._owner._materialize_rank(self._rank)
        return self._val

    def __array__(self, dtype=None, copy=None):
        val = self.materialize()
        return val if dtype is None else val.astype(dtype, copy=False)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, item):
        if self._val is None and isinstance(item, (int, np.integer)):
            i = int(item)
            if not -self.shape[0] <= i < self.shape[0]:
                raise IndexError('index %d is out of bounds for axis 0 with size %d' % (i, self.shape[0]))
            return self._owner.rows([(self._rank, self._chain, i % self.shape[0])])[0]
        return self.materialize()[item]


    @property
    def on_host(self):
        return self._val is not None

    def row_request(self, i):
        return (self._rank, self._chain, int(i) % self.shape[0])


class LazyV(LazyH):
    """the singular-space vectors v of one alpha scan (n_alpha x n_s), on the device until somebody looks: 13 MB of
    a 16 x 16 x 100 job that the result object hardly ever shows"""
    _what = 'v'

    def __getitem__(self, item):
        return self.materialize()[item]


class LazyA(object):
    """A = A_of_H(H) of one alpha scan (H / delta, or B H with a preblur), formed when looked at"""

    def __init__(self, H, A_of_H):
        self._H, self._map = H, A_of_H
        self.shape, self.dtype, self.ndim = H.shape, np.dtype(float), 2
        self._val = None

    def __array__(self, dtype=None, copy=None):
        if self._val is None:
            self._val = np.asarray(self._map.f(np.asarray(self._H)))
        return self._val if dtype is None else self._val.astype(dtype, copy=False)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, item):
        if self._val is None and isinstance(item, (int, np.integer)):
            return np.asarray(self._map.f(np.asarray(self._H[int(item)])))       # (LazyH normalises a negative index)
        return np.asarray(self)[item]

    @property
    def on_host(self):
        return self._val is not None or getattr(self._H, 'on_host', True)

    def row_request(self, i):
        return self._H.row_request(i)

    def from_H_row(self, row):
        return np.asarray(self._map.f(row))


class BatchSolver(object):
    def __init__(self, K, device_ids=(0,)):
        K.S                                         # decompose if needed
        self.device_ids = tuple(int(d) for d in device_ids)
        if not self.device_ids:
            raise ValueError('at least one device is needed')
        self._token = self._kernel_token(K)
        U = None if K.rotation is not None else K.U
        self.ctxs = [device.DeviceContext(U, K.S, K.V, device=d) for d in self.device_ids]
        self.n_s, self.n_omega = self.ctxs[0].n_s, self.ctxs[0].n_omega
        if len(self.ctxs) > 1:
            device.comm_init_local(self.ctxs)
        self._pending = []                          # weak references to the LazyH whose data still live in a result buffer
        self._layout = None
        self.last_info = None

    # ---- reuse -------------------------------------------------------------
    @staticmethod
    def _kernel_token(K):
        """what the staged basis depends on, as the OBJECTS themselves (held, so that their addresses cannot be
        recycled by a re-decomposition): U too -- a covariance rotation with truncated eigenvalues followed by
        set_error leaves another U beside the same S and V"""
        return (K._U, K._S, K._V, K.rotation is None)

    @staticmethod
    def _same_token(a, b):
        return a is not None and b is not None and all((x is y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3]

    @classmethod
    def for_kernel(cls, K, device_ids=(0,)):
        K.S
        device_ids = tuple(int(d) for d in device_ids)
        held = K.__dict__.get('_batch_solvers')
        if held is None:
            held = K.__dict__['_batch_solvers'] = {}
        s = held.get(device_ids)
        if s is not None and cls._same_token(s._token, cls._kernel_token(K)) and s.ctxs[0]._h:
            return s
        if s is not None:
            s.close()
        s = held[device_ids] = cls(K, device_ids)
        return s

    def close(self):
        self.materialize_pending()
        pool = self.__dict__.pop('_pool', None)
        if pool is not None:
            pool.shutdown(wait=True)
        for c in self.ctxs:
            c.close()

    # ---- one batch -----------------------------------------------------------
    def solve(self, K, specs, opts, want_logdet=False, want_H='lazy', output_map=None, select=(0, 0.2)):
        """``specs``: dicts with G, err, U_rot (or None), D, kind, v0, alpha (equal lengths).  Returns
        (list of per-spec result dicts in the order of ``specs``, info).  ``want_H``: 'lazy' (default),
        True (fetched now) or False.  ``select`` = (linefit_deg, gamma): the three default analyzers' alphas are picked
        on the device behind the solve (``mxe_select3_launch``) and come back with their H rows as ``device_select``
        of every result; None: not."""
        self.materialize_pending()                  # the result buffers are about to be overwritten
        n_alpha = len(specs[0]['alpha'])
        for s in specs:
            if len(s['alpha']) != n_alpha:
                raise ValueError('all elements of a batch need the same number of alpha values')
        N = len(self.ctxs)
        rank_of, local_of, n_local = device.shard_plan(len(specs), N)
        per_rank = [[i for i in range(len(specs)) if rank_of[i] == r] for r in range(N)]
        active = [r for r in range(N) if per_rank[r]]
        eta = float(getattr(opts, 'chi2_factor', 1.0)) if opts is not None else 1.0
        gather = len(active) == N and N > 1
        outs = [None] * N
        picks = [None] * N

        def work(r):
            # stage -> launch -> finish -> (fetch) of ONE device; with several devices each runs on a thread of its own
            # (ctypes releases the GIL, the library promises one thread per context: include/maxent_hip.h)
            c = self.ctxs[r]
            self._stage(c, K, [specs[i] for i in per_rank[r]], opts)
            c.launch()
            c.finish()                              # (alphas the lock-step layout gave up on: one-chain layout)
            if select is not None:
                c.select3_launch(select[0], select[1])
            elif gather:
                c.select_launch(0)
            if not gather:
                outs[r] = c.fetch(want_v=False, want_H=False)
            if select is not None:
                picks[r] = c.select3_fetch()

        self._on_devices(work, active)
        info = None
        if gather:
            # ONE gather of the per-alpha scalars (and of the analyzer's rows) to the first device
            counts = [self.ctxs[r].compact_count() for r in range(N)]
            recv = np.empty(int(np.sum(counts)))
            device.gather_local(self.ctxs, 0, counts, full=False, recv=recv)
            offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)

            def rest(r):
                outs[r] = self._unpack_compact(recv[offs[r]:offs[r] + counts[r]], len(per_rank[r]), n_alpha, eta)
                extra = self.ctxs[r].fetch(want_v=False, want_H=False)
                for k in ('n_iter', 'converged', 'n_evals'):
                    outs[r][k] = extra[k]
            self._on_devices(rest, list(range(N)))
        logdets = {r: self.ctxs[r].logdet() for r in active} if want_logdet else {}
        maps = {r: self.ctxs[r].apply_output_map(output_map) for r in active} if output_map is not None else {}
        ms = [self.ctxs[r].last_kernel_ms() for r in active]
        info = dict(kernel_ms=max(ms), kernel_ms_per_device=ms, devices=[self.device_ids[r] for r in active],
                    n_datasets=[self._n_datasets.get(id(self.ctxs[r]), 0) for r in active])
        info.update(self.ctxs[active[0]].last_launch_info())
        self.last_info = info
        res = []
        for i, s in enumerate(specs):
            r, c = int(rank_of[i]), int(local_of[i])
            o = outs[r]
            H = LazyH(self, r, c, n_alpha, self.n_omega)
            v = LazyV(self, r, c, n_alpha, self.n_s)
            self._pending.append(weakref.ref(H))
            self._pending.append(weakref.ref(v))
            d = dict(alpha=np.asarray(s['alpha'], dtype=float), H=H,
                     A=(maps[r][c] if r in maps else None),
                     v=v, chi2=o['chi2'][c], S=o['S'][c], Q=o['Q'][c],
                     n_iter=o['n_iter'][c], converged=o['converged'][c].astype(bool), n_evals=o['n_evals'][c])
            if 'linefit_index' in o:
                d['device_linefit_index'] = int(o['linefit_index'][c])
                d['device_linefit_H'] = o['linefit_H'][c]
            if picks[r] is not None:
                d['device_select'] = dict(params=(int(select[0]), float(select[1])), index=picks[r][0][:, c], H=picks[r][1][:, c])
                if 'device_linefit_index' not in d:
                    d['device_linefit_index'], d['device_linefit_H'] = int(picks[r][0][0, c]), picks[r][1][0, c]
            if r in logdets:
                d['logdet'] = logdets[r][c]
            res.append(d)
        if want_H is True:
            self.materialize_pending()
        elif want_H is False:
            self._pending = [ref for ref in self._pending if isinstance(ref(), LazyV)]
            for d in res:
                d['H'] = None
        return res, info

    def _stage(self, ctx, K, specs, opts):
        """data sets, elements and chains of one device.  What is staged is remembered (contents, not identities): the
        same job again -- the same object run twice, a parameter of the analyzers changed -- uploads nothing"""
        n_tau = len(specs[0]['G'])
        errs = np.stack([np.asarray(s['err'], dtype=float) * np.ones(len(s['G'])) for s in specs]) \
            if all(len(s['G']) == n_tau for s in specs) else None
        staged = dict(
            G=np.stack([np.asarray(s['G'], dtype=float) for s in specs]) if errs is not None else None,
            err=errs,
            D=np.stack([np.asarray(s['D'], dtype=float) for s in specs]),
            alpha=np.stack([np.asarray(s['alpha'], dtype=float) for s in specs]),
            v0=np.stack([np.asarray(s['v0'], dtype=float) for s in specs]),
            kinds=np.array([s['kind'] for s in specs]),
            U_rot=[s.get('U_rot') for s in specs],
            opts=(bytes(opts) if opts is not None else b''), rotated=K.rotation is not None)
        held = self.__dict__.setdefault('_staged', {})
        old = held.get(id(ctx))
        if old is not None and errs is not None and old['G'] is not None and ctx._n_chain == len(specs) and \
                old['opts'] == staged['opts'] and old['rotated'] == staged['rotated'] and \
                len(old['U_rot']) == len(staged['U_rot']) and all(a is b for a, b in zip(old['U_rot'], staged['U_rot'])) and \
                all(old[k].shape == staged[k].shape and np.array_equal(old[k], staged[k])
                    for k in ('G', 'err', 'D', 'alpha', 'v0', 'kinds')):
            return
        held.pop(id(ctx), None)
        ctx.clear_datasets()
        ds_ids, seen = [], []
        self.__dict__.setdefault('_n_datasets', {})
        for n, s in enumerate(specs):
            err = errs[n] if errs is not None else np.asarray(s['err'], dtype=float) * np.ones(len(s['G']))
            U_rot = s.get('U_rot')
            found = None
            for (e0, u0, i0) in seen:
                if u0 is U_rot and e0.shape == err.shape and np.array_equal(e0, err):
                    found = i0
                    break
            if found is None:
                found = ctx.add_dataset(err, K.U if (U_rot is None and K.rotation is not None) else U_rot)
                seen.append((err, U_rot, found))
            ds_ids.append(found)
        self._n_datasets[id(ctx)] = len(seen)
        ctx.set_elements(ds_ids, [s['G'] for s in specs], staged['D'], [s['kind'] for s in specs])
        ctx.upload_chains(np.arange(len(specs), dtype=np.int32), staged['alpha'], staged['v0'], opts)
        held[id(ctx)] = staged

    def _on_devices(self, fn, ranks):
        """fn(rank) for every rank: in this thread for one device, one thread per device otherwise"""
        if len(ranks) <= 1:
            for r in ranks:
                fn(r)
            return
        pool = self.__dict__.get('_pool')
        if pool is None or pool._max_workers < len(ranks):
            from concurrent.futures import ThreadPoolExecutor
            pool = self.__dict__['_pool'] = ThreadPoolExecutor(max_workers=len(self.ctxs))
        for f in [pool.submit(fn, r) for r in ranks]:
            f.result()                              # (re-raises what a worker raised)

    def _unpack_compact(self, pack, n_chain, n_alpha, eta=1.0):
        """the compact result pack of one rank (include/maxent_hip.h: MXE_GATHER_COMPACT).  The kernel iterates on
        alpha / eta and stores Q / eta (eta = chi2_factor); mxe_chains_fetch scales Q on the host, and so does this"""
        P, nw = n_chain * n_alpha, self.n_omega
        out = dict(chi2=pack[:P].reshape(n_chain, n_alpha), S=pack[P:2 * P].reshape(n_chain, n_alpha),
                   Q=(pack[2 * P:3 * P] * eta).reshape(n_chain, n_alpha))
        out['linefit_H'] = pack[3 * P:3 * P + n_chain * nw].reshape(n_chain, nw)
        out['linefit_index'] = pack[3 * P + n_chain * nw:].astype(np.int32)
        return out

    # ---- H on demand ----------------------------------------------------------
    def _alive(self):
        """the LazyH somebody still holds and that have not been fetched (a result that was dropped takes
        its claim on the device buffer with it)"""
        out = []
        for ref in self._pending:
            h = ref()
            if h is not None and h._val is None:
                out.append(h)
        return out

    def _materialize_rank(self, rank):
        alive = self._alive()
        mine = [h for h in alive if h._rank == rank]
        if mine:
            got = self.ctxs[rank].fetch(want_v=any(h._what == 'v' for h in mine), want_H=any(h._what == 'H' for h in mine))
            for h in mine:
                h._val = got[h._what][h._chain]
        self._pending = [weakref.ref(h) for h in alive if h._val is None]

    def materialize_pending(self):
        for r in sorted(set(h._rank for h in self._alive())):
            self._materialize_rank(r)
        self._pending = []

    def rows(self, wanted):
        """``wanted``: list of (rank, chain, alpha index) of the LAST batch -> array (len, n_omega); one
        device-to-host copy per row, nothing else moves"""
        out = np.empty((len(wanted), self.n_omega))
        by_rank = {}
        for n, (r, c, i) in enumerate(wanted):
            by_rank.setdefault(r, []).append((n, c * self.ctxs[r]._n_alpha + i))
        for r, lst in by_rank.items():
            got = self.ctxs[r].fetch_rows([p for _, p in lst])
            for (n, _), row in zip(lst, got):
                out[n] = row
        return out
