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

import atexit
import threading
import weakref

import os

import operator

import numpy as np

from . import device


class Claim(object):
    """what ONE launch still holds in the result buffers of a device: all H, or all v, of its scans.  The lazy arrays of the
    single scans (:class:`LazyH`, :class:`LazyV`) and the result object of the launch (``elementwise_maxent.DeferredLaunch``)
    hold it strongly, the solver weakly: a result that was dropped takes its claim with it.  ``val``: the whole array
    [n_chain][n_alpha][...] once it has come to the host."""
    __slots__ = ('_owner', '_rank', '_what', 'val', 'shape', '__weakref__')

    def __init__(self, owner, rank, what, shape):
        self._owner, self._rank, self._what, self.val, self.shape = owner, rank, what, None, shape

    def materialize(self):
        if self.val is None:
            # (everything this launch still holds on the device comes with it -- v: 13 MB next to H's 100 --: what has been
            #  brought over has no claim on the buffers any more, a new object on the same grids may take the contexts over)
            self._owner._materialize_rank(self._rank)
        return self.val


class LazyH(object):
    """hidden images of one alpha scan, still on the device until somebody looks"""
    _what = 'H'
    __slots__ = ('_claim', '_chain', 'shape', '__weakref__')
    dtype = np.dtype(float)
    ndim = 2

    def __init__(self, claim, chain, n_alpha, n_omega):
        self._claim, self._chain = claim, chain
        self.shape = (n_alpha, n_omega)

    @property
    def _owner(self):
        return self._claim._owner

    @property
    def _rank(self):
        return self._claim._rank

    @property
    def _val(self):
        whole = self._claim.val
        return None if whole is None else whole[self._chain]

    def materialize(self):
        return self._claim.materialize()[self._chain]

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
    __slots__ = ()

    def __getitem__(self, item):
        return self.materialize()[item]


class LazyRows(object):
    """the H rows ONE of the three device analyzers chose for the scans of a launch ([n_chain][n_omega]), on the device until
    somebody looks (``mxe_select3_fetch_rows``): a result shows the rows of its default analyzer, which come with the solve;
    those of the other two are 1 MB each for a 16 x 16 job"""
    _what = 'rows'

    def __init__(self, owner, rank, which, n_chain, n_omega):
        self._owner, self._rank, self._which = owner, rank, which
        self.shape, self.dtype, self.ndim = (n_chain, n_omega), np.dtype(float), 2
        self._val = None

    def materialize(self):
        if self._val is None:
            self._owner._materialize_rows(self)
        return self._val

    def __array__(self, dtype=None, copy=None):
        val = self.materialize()
        return val if dtype is None else val.astype(dtype, copy=False)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, item):
        return self.materialize()[item]

    @property
    def on_host(self):
        return self._val is not None


class PickedRows(object):
    """``device_select['H']`` of one scan: [which] -> the H row analyzer ``which`` chose for it"""
    __slots__ = ('_rows', '_chain')

    def __init__(self, rows, chain):
        self._rows, self._chain = rows, chain

    def __getitem__(self, which):
        return self._rows[which][self._chain]

    def __len__(self):
        return len(self._rows)


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


class LazySols(object):
    """the per-scan result dicts of ONE launch on ONE device -- what :meth:`BatchSolver.solve` returns as a list --, each built
    when it is first asked for (256 dicts, lazy arrays and views cost 0.5 ms that a caller who reads ``result.A_out`` or
    ``result.chi2`` never needs).  ``arrays``: chi2 / S / Q / n_iter / n_evals [n][n_alpha], ``conv`` the flags as bool,
    ``picks`` = (indices [3][n], rows [LazyRows x 3], eager rows) or None"""

    def __init__(self, n, alpha_tab, alpha_sel, claim_H, claim_v, arrays, conv, picks, sel_params, n_omega, n_s):
        self.n, self.alpha_tab, self.alpha_sel = n, alpha_tab, alpha_sel
        self.claim_H, self.claim_v, self.arrays, self.conv, self.picks, self.sel_params = claim_H, claim_v, arrays, conv, picks, sel_params
        self.n_alpha, self.n_omega, self.n_s = alpha_tab.shape[1], n_omega, n_s
        self._made = {}

    def __len__(self):
        return self.n

    def __iter__(self):
        return (self[i] for i in range(self.n))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        i = int(i)
        if i < 0:
            i += self.n
        d = self._made.get(i)
        if d is None:
            if not 0 <= i < self.n:
                raise IndexError(i)
            o = self.arrays
            d = dict(alpha=self.alpha_tab[self.alpha_sel[i] if self.alpha_sel is not None else 0],
                     H=LazyH(self.claim_H, i, self.n_alpha, self.n_omega), A=None,
                     v=LazyV(self.claim_v, i, self.n_alpha, self.n_s),
                     chi2=o['chi2'][i], S=o['S'][i], Q=o['Q'][i], n_iter=o['n_iter'][i], converged=self.conv[i],
                     n_evals=o['n_evals'][i])
            if self.picks is not None:
                idx, rows = self.picks[0], self.picks[1]
                d['device_select'] = dict(params=self.sel_params, index=idx[:, i], H=PickedRows(rows, i), batch=self.picks, chain=i)
            self._made[i] = d
        return d


def _one_at_a_time(method):
    """the solver's lock around a method: solvers are shared between kernel objects with the same decomposition, and
    the library wants one thread per context at a time"""
    import functools

    @functools.wraps(method)
    def locked(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)
    return locked


DEVICE_DIRECTIONS = 64          # singular directions the lock-step kernels hold (mxe_kernel_mc.hip.h: NP = 64)
KEEP_BOUND = 1e-7               # largest |delta u| = |delta H / H| the dropped directions may cause (the parity gate is 1e-6; the bound overestimates tenfold)


def directions_to_keep(K, specs=None, arrays=None):
    """How many singular directions of ``K`` the device has to solve this job with: ``None`` (all), or 64 when there are more and
    the others cannot be told from zero in it.

    More than 64 singular values above the reference's absolute threshold (1e-14, maxent_loop.py:91,184) come with many data
    points -- 1 000 imaginary times: 79 -- and the device then has only its one-chain kernel with a 128 x 128 Newton matrix
    (15-80 x slower than the lock-step kernels).  But those directions sit at the rounding floor of the decomposition
    (S_64 / S_0 ~ 2e-16).  At the minimiser every direction obeys  alpha v_k = -c_k rho_k,  c_k = S_k / sigma,
    rho_k = c_k h_k - ghat_k  (grad Q = W g with W non-singular), so a dropped direction would have carried
    |v_k| <= c_k (|ghat_k| + c_k |h_k|) / alpha, and u = V v changes by at most the 2-norm of these (the rows of V are at most of
    unit length).  The job keeps 64 directions -- or 128, where 64 are too few and there are more: a mesh of 2 000 frequencies on
    1 000 data points leaves 991 'singular values' above 1e-14, the rounding floor of that decomposition -- when that norm, with
    the smallest alpha of the job, stays below ``KEEP_BOUND`` for every element; the v it returns are zero in the others.  Only for ONE error bar per element and an
    unrotated kernel: there M = S U^T diag(1 / err^2) U S is diagonal and rho_k is the direction's own."""
    S = np.asarray(K.S, dtype=float)
    n_s = S.shape[0]
    if n_s <= DEVICE_DIRECTIONS or K.rotation is not None or os.environ.get('MAXENT_AMD_ALL_DIRECTIONS'):
        return None
    try:
        if arrays is not None:
            G = np.asarray(arrays['G'], dtype=float)
            err = np.asarray(arrays['err'], dtype=float) * np.ones((1, G.shape[1]))
            if err.shape[0] not in (1, G.shape[0]):
                err = err[np.asarray(arrays['sel'])]                    # (rows per class of elements)
            amin = float(np.min(arrays['alpha']))                       # (the smallest alpha of the job, for every element)
            sumD = float(np.max(np.sum(np.abs(np.asarray(arrays['D'], dtype=float)), axis=-1)))
        else:
            if any(sp.get('U_rot') is not None for sp in specs):
                return None
            G = np.stack([np.asarray(sp['G'], dtype=float) for sp in specs])
            err = np.stack([np.asarray(sp['err'], dtype=float) * np.ones(G.shape[1]) for sp in specs])
            amin = float(min(np.min(sp['alpha']) for sp in specs))
            sumD = float(max(np.sum(np.abs(np.asarray(sp['D'], dtype=float))) for sp in specs))
        if np.any(np.max(err, axis=1) != np.min(err, axis=1)):
            return None         # (error bars that vary over the data points couple the directions through M = S U^T diag(1 / err^2) U S)
        ghat_all = np.abs((G / err) @ np.asarray(K.U)[:, DEVICE_DIRECTIONS:])          # [element][direction beyond the 64th]
        c_all = S[None, DEVICE_DIRECTIONS:] / err[:, :1]
        h1 = 1e3 * max(1.0, sumD)                                       # |h_k| = |V_k^T H| <= |H|_2: generous
        per = c_all * (ghat_all + c_all * h1)                           # |v_k| alpha of a dropped direction, at most
        for keep in (DEVICE_DIRECTIONS, 2 * DEVICE_DIRECTIONS):
            if keep >= n_s:
                break
            # u = V v and the rows of V are at most of unit length: |delta u_i| <= |v_dropped|_2
            bound = np.sqrt(np.sum(per[:, keep - DEVICE_DIRECTIONS:] ** 2, axis=1)) / amin
            if np.all(np.isfinite(bound)) and float(np.max(bound)) <= KEEP_BOUND:
                return keep
    except Exception:
        return None
    return None


class BatchSolver(object):
    def __init__(self, K, device_ids=(0,), keep=None):
        K.S                                         # decompose if needed
        self.device_ids = tuple(int(d) for d in device_ids)
        if not self.device_ids:
            raise ValueError('at least one device is needed')
        self._token = self._kernel_token(K)
        self._keep = keep
        U = None if K.rotation is not None else K.U
        if (len(K.S) if keep is None else keep) > 2 * DEVICE_DIRECTIONS:
            S = np.asarray(K.S)
            raise device.MaxEntDeviceError(
                'the kernel has %d singular values above its threshold and the device solves with at most %d; those beyond the '
                'first ~60 usually are the rounding floor of the decomposition (here S[%d] / S[0] = %.1e) and could not be '
                'dropped for this job (error bars far below the noise of the data, a rotated kernel or error bars that vary): '
                'K.reduce_singular_space(threshold) with a threshold above that floor' % (
                    len(S), 2 * DEVICE_DIRECTIONS, 2 * DEVICE_DIRECTIONS, S[2 * DEVICE_DIRECTIONS] / S[0]))
        self.ctxs = [device.DeviceContext(U, K.S, K.V, device=d, keep=keep) for d in self.device_ids]
        self.n_s, self.n_omega = self.ctxs[0].n_s, self.ctxs[0].n_omega
        if len(self.ctxs) > 1:
            device.comm_init_local(self.ctxs)
        self._pending = []                          # weak references to the LazyH whose data still live in a result buffer
        self._layout = None
        self.last_info = None
        self._busy = False                          # a batch is in flight (between solve_begin and the function it returned)
        self._lock = threading.RLock()              # (one job at a time on these contexts: include/maxent_hip.h)

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

    # Solvers outlive the kernel object they were made for: a new object with the SAME decomposition (the next
    # TauMaxEnt / ElementwiseMaxEnt on the same grids -- every iteration of a self-consistency loop) takes the contexts
    # and what is staged on them instead of creating and filling its own (2 ms) and destroying the old ones (2 ms).
    # The few most recently used are kept (each holds the result buffers of its last job on the device).
    _pool_lock = threading.Lock()
    _pooled = []                                    # most recently used first
    POOL_SIZE = 8                                   # (eight jobs in flight on one kernel: maxent_amd.run_many)

    def _same_contents(self, K, device_ids, keep=None):
        if self.device_ids != device_ids or not self.ctxs[0]._h or (K.rotation is None) != self._token[3] or self._keep != keep:
            return False
        for a, b in zip(self._token[:3], (K._U, K._S, K._V)):
            if a is b:
                continue
            if a is None or b is None or a.shape != b.shape or not np.array_equal(a, b):
                return False
        return True

    @classmethod
    def for_kernel(cls, K, device_ids=(0,), keep=None):
        K.S
        device_ids = tuple(int(d) for d in device_ids)
        held = K.__dict__.get('_batch_solvers')
        if held is None:
            held = K.__dict__['_batch_solvers'] = {}
        key = device_ids if keep is None else device_ids + ('keep', int(keep))
        s = held.get(key)
        if s is not None and cls._same_token(s._token, cls._kernel_token(K)) and s.ctxs[0]._h and not s._busy:
            return s           # (a solver with a batch in flight -- solve_begin without its end -- is nobody else's)
        if cls.POOL_SIZE <= 0:                       # no pool: every kernel object its own contexts
            if s is not None and not s._busy:
                s.close()
            s = held[key] = cls(K, device_ids, keep)
            return s
        with cls._pool_lock:
            found = None
            for cand in cls._pooled:
                # (not one whose last results somebody still holds unfetched: they would have to come to the host
                #  first -- 102 MB for a 16 x 16 x 100 job --, a context of its own is cheaper)
                if not cand._busy and cand._same_contents(K, device_ids, keep) and not cand._alive():
                    found = cand
                    break
            if found is None:
                found = cls(K, device_ids, keep)
            else:
                cls._pooled.remove(found)
                found._token = cls._kernel_token(K)        # (equal arrays: the staged basis is that of K)
            cls._pooled.insert(0, found)
            keep = [x for x in cls._pooled[cls.POOL_SIZE:] if x._busy]
            retired = [x for x in cls._pooled[cls.POOL_SIZE:] if not x._busy]
            del cls._pooled[cls.POOL_SIZE:]
            cls._pooled.extend(keep)
        for old in retired:
            if not old._alive():
                old.close()                         # (one with results out lives as long as they do: they hold it)
        held[key] = found
        return found

    @classmethod
    def close_pool(cls):
        """close every pooled solver (also registered for the end of the process: the contexts go before the HIP
        runtime does)"""
        with cls._pool_lock:
            retired, cls._pooled = cls._pooled, []
        for s in retired:
            try:
                s.close()
            except Exception:
                pass

    def close(self):
        with self._pool_lock:
            if self in self._pooled:
                self._pooled.remove(self)
        if not self.ctxs or not self.ctxs[0]._h:
            return
        self.materialize_pending()
        pool = self.__dict__.pop('_pool', None)
        if pool is not None:
            pool.shutdown(wait=True)
        for c in self.ctxs:
            c.close()

    # ---- one batch -----------------------------------------------------------
    def solve(self, K, specs, opts, want_logdet=False, want_H='lazy', output_map=None, select=(0, 0.2), while_waiting=None):
        """one batch, start to end: :meth:`solve_begin` and the function it returns"""
        return self.solve_begin(K, specs, opts, want_logdet=want_logdet, want_H=want_H, output_map=output_map, select=select,
                                while_waiting=while_waiting)()

    def solve_begin(self, K, specs, opts, want_logdet=False, want_H='lazy', output_map=None, select=(0, 0.2), while_waiting=None,
                    early_select=False, arrays=None):
        """Stage and LAUNCH one batch; returns the function that waits for it and returns what :meth:`solve` returns.  Between
        the two the solver is busy (``for_kernel`` hands it to nobody else) and the caller is free: it may begin the batches of
        OTHER solvers -- several jobs in flight on the GPU, launched from one thread without a wait in between
        (``ElementwiseMaxEnt.run_async``, ``maxent_amd.run_many``) -- and do its own host work.  ``early_select``: the selection
        kernel of the analyzers is enqueued right behind the solve instead of behind ``mxe_chains_finish`` (which waits for the
        device), and once more in the rare case that the finishing pass had something to do.  With several devices everything
        happens in this call and the returned function only hands the results over.
        ``arrays`` (instead of ``specs``, one device): the job as ARRAYS -- dict(G [n][n_tau], err [1 or n][n_tau], sel [n]: which
        row of the small tables D [m][n_omega], alpha [m][n_alpha], v0 [m][n_s], kinds [m] a scan takes) for an unrotated kernel --;
        the results then come as a :class:`LazySols` (per-scan dicts built when asked for), nothing per scan is done here.
        ``specs``: dicts with G, err, U_rot (or None), D, kind, v0, alpha (equal lengths).  Returns
        (list of per-spec result dicts in the order of ``specs``, info).  ``want_H``: 'lazy' (default),
        True (fetched now) or False.  ``select`` = (linefit_deg, gamma[, default]): the three default analyzers' alphas are
        picked on the device behind the solve (``mxe_select3_launch``) and come back as ``device_select`` of every result
        -- the indices of all three and the H rows of analyzer ``default`` (0 line fit, 1 chi2 curvature, 2 entropy: the one
        ``result.A_out`` shows) at once, the rows of the other two when somebody looks at them; None: not.
        ``while_waiting(results)``: called while the kernel runs, with the result dicts complete but for their VALUES (the
        arrays are there and are filled behind it) -- a caller builds its records from them then; only with one device and
        neither ``want_logdet`` nor ``output_map`` (which add keys later): who passes it checks that it was called."""
        self._lock.acquire()                        # (released by the returned function, or below if this call fails)
        try:
            if self._busy:
                raise RuntimeError('this solver has a batch in flight: end it first')
            if arrays is not None:
                rest = self._solve_begin_arrays(K, arrays, opts, select, early_select)
            else:
                rest = self._solve_begin(K, specs, opts, want_logdet, want_H, output_map, select, while_waiting, early_select)
            self._busy = True
        except BaseException:
            self._lock.release()
            raise

        def end():
            try:
                return rest()
            finally:
                self._busy = False
                self._lock.release()
        return end

    def _solve_begin(self, K, specs, opts, want_logdet, want_H, output_map, select, while_waiting, early_select):
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
        conv = {}
        eager = int(select[2]) if select is not None and len(select) > 2 else 0
        sel_params = (int(select[0]), float(select[1])) if select is not None else None

        def begin(r):
            # stage -> launch of ONE device (the launch returns at once)
            c = self.ctxs[r]
            self._stage(c, K, [specs[i] for i in per_rank[r]], opts)
            c.launch()
            if early_select and select is not None:
                c.select3_launch(select[0], select[1])

        def destinations(r):
            # the host arrays the results of device r come into -- made BEFORE the device is waited for, so that the per-scan views
            # of them can be cut while the kernel runs
            c = self.ctxs[r]
            if not gather:
                outs[r] = c.result_arrays()
                conv[r] = np.empty(outs[r]['converged'].shape, dtype=bool)
            if select is not None:
                idx, row = c.select3_arrays(1)
                rows = [LazyRows(self, r, w, c._n_chain, self.n_omega) for w in range(3)]
                rows[eager]._val = row[0]
                for w in range(3):
                    if w != eager:
                        self._pending.append(weakref.ref(rows[w]))
                picks[r] = (idx, rows, row)

        def end(r):
            # finish -> (fetch) of ONE device
            c = self.ctxs[r]
            left = c.finish()                       # (alphas the lock-step layout gave up on: one-chain layout)
            if select is not None:
                if not early_select or left:
                    c.select3_launch(select[0], select[1])
            elif gather:
                c.select_launch(0)
            if not gather:
                c.fetch(want_v=False, want_H=False, out=outs[r])
                np.not_equal(outs[r]['converged'], 0, out=conv[r])
            if select is not None:
                # (the indices of all three analyzers and the rows of the result's default analyzer; the rows of the other two when
                #  somebody looks at them)
                c.select3_fetch_rows(first=eager, count=1, idx=picks[r][0], rows=picks[r][2])

        def skeleton():
            # what of the results does not wait for the device: built while the kernel runs
            out = []
            claims = {}
            for r in active:
                nc = len(per_rank[r])
                claims[r] = (Claim(self, r, 'H', (nc, n_alpha, self.n_omega)), Claim(self, r, 'v', (nc, n_alpha, self.n_s)))
                self._pending.append(weakref.ref(claims[r][0]))
                self._pending.append(weakref.ref(claims[r][1]))
            for i, s in enumerate(specs):
                r, c = int(rank_of[i]), int(local_of[i])
                out.append(dict(alpha=np.asarray(s['alpha'], dtype=float), H=LazyH(claims[r][0], c, n_alpha, self.n_omega), A=None,
                                v=LazyV(claims[r][1], c, n_alpha, self.n_s)))
            return out

        def attach(res):
            # the per-scan views of the arrays of ``destinations`` (filled by ``end``: with one device while the kernel runs)
            for i, d in enumerate(res):
                r, c = int(rank_of[i]), int(local_of[i])
                o = outs[r]
                if o is not None:
                    d.update(chi2=o['chi2'][c], S=o['S'][c], Q=o['Q'][c],
                             n_iter=o['n_iter'][c], converged=conv[r][c], n_evals=o['n_evals'][c])
                if picks[r] is not None:
                    idx, rows = picks[r][0], picks[r][1]
                    d['device_select'] = dict(params=sel_params, index=idx[:, c], H=PickedRows(rows, c), batch=picks[r], chain=c)

        def tail(res):
            info = None
            if gather:
                # ONE gather of the per-alpha scalars (and of the analyzer's rows) to the first device
                counts = [self.ctxs[r].compact_count() for r in range(N)]
                recv = np.empty(int(np.sum(counts)))
                device.gather_local(self.ctxs, 0, counts, full=False, recv=recv)
                offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)

                def rest_of(r):
                    outs[r] = self._unpack_compact(recv[offs[r]:offs[r] + counts[r]], len(per_rank[r]), n_alpha, eta)
                    extra = self.ctxs[r].fetch(want_v=False, want_H=False)
                    for k in ('n_iter', 'converged', 'n_evals'):
                        outs[r][k] = extra[k]
                    conv[r] = outs[r]['converged'].astype(bool)
                self._on_devices(rest_of, list(range(N)))
                attach(res)
                for i, d in enumerate(res):
                    r, c = int(rank_of[i]), int(local_of[i])
                    d['device_linefit_index'] = int(outs[r]['linefit_index'][c])
                    d['device_linefit_H'] = outs[r]['linefit_H'][c]
            logdets = {r: self.ctxs[r].logdet() for r in active} if want_logdet else {}
            maps = {r: self.ctxs[r].apply_output_map(output_map) for r in active} if output_map is not None else {}
            ms = [self.ctxs[r].last_kernel_ms() for r in active]
            info = dict(kernel_ms=max(ms), kernel_ms_per_device=ms, devices=[self.device_ids[r] for r in active],
                        n_datasets=[self._n_datasets.get(id(self.ctxs[r]), 0) for r in active])
            info.update(self.ctxs[active[0]].last_launch_info())
            if os.environ.get('MAXENT_AMD_AUDIT'):
                # on request (the parity tests): mxe_audit over every problem of this launch -- the exact binary64 Newton correction
                # at the returned v, ||w * V delta|| / ||H||, to first order the distance of the returned H from the minimiser
                corr = np.concatenate([self.ctxs[r].audit()['corr'].ravel() for r in active])
                conv_all = np.concatenate([outs[r]['converged'].ravel() for r in active]).astype(bool)
                info['audit_max'] = float(np.nanmax(corr[conv_all])) if conv_all.any() else 0.0
                info['audit_problems'] = int(conv_all.sum())
            self.last_info = info
            if maps or logdets:
                for i, d in enumerate(res):
                    r, c = int(rank_of[i]), int(local_of[i])
                    if r in maps:
                        d['A'] = maps[r][c]
                    if r in logdets:
                        d['logdet'] = logdets[r][c]
            if want_H is True:
                self.materialize_pending()
            elif want_H is False:
                self._pending = [ref for ref in self._pending if getattr(ref(), '_what', 'H') != 'H']       # (claims on H given up)
                for d in res:
                    d['H'] = None
            return res, info

        # with several devices each runs on a thread of its own (ctypes releases the GIL, the library promises one
        # thread per context: include/maxent_hip.h)
        if len(active) == 1:
            begin(active[0])
            res = skeleton()
            destinations(active[0])
            attach(res)
            if while_waiting is not None and not want_logdet and output_map is None and want_H == 'lazy':
                while_waiting(res)

            def rest():
                end(active[0])
                return tail(res)
        else:
            def one(r):
                begin(r)
                destinations(r)         # (after the staging: the context knows its chains)
                end(r)
            waiting = self._on_devices(one, active, wait=False)
            res = skeleton()
            waiting()
            if not gather:
                attach(res)
            done = tail(res)

            def rest():
                return done
        return rest

    def _solve_begin_arrays(self, K, arrays, opts, select, early_select):
        """:meth:`solve_begin` for a job given as arrays, on the one device of this solver: stage (nothing when the job is what
        the context holds), launch, the selection kernel behind it; the returned function waits, brings the scalars and the
        analyzers' choice and returns (:class:`LazySols`, info)"""
        if len(self.ctxs) != 1:
            raise ValueError('a job given as arrays runs on one device')
        self.materialize_pending()                  # the result buffers are about to be overwritten
        c = self.ctxs[0]
        n, n_alpha = int(arrays['n']), arrays['alpha'].shape[1]
        self._stage_arrays(c, K, arrays, opts)
        c.launch()
        if select is not None and early_select:
            c.select3_launch(select[0], select[1])
        eager = int(select[2]) if select is not None and len(select) > 2 else 0
        sel_params = (int(select[0]), float(select[1])) if select is not None else None
        claim_H, claim_v = Claim(self, 0, 'H', (n, n_alpha, self.n_omega)), Claim(self, 0, 'v', (n, n_alpha, self.n_s))
        self._pending.append(weakref.ref(claim_H))
        self._pending.append(weakref.ref(claim_v))
        # (scalars, flags and the default analyzer's rows are copied out behind the kernels, into one page-locked block: with
        #  jobs in flight on several contexts they are in memory when rest() is called instead of being copied then)
        ahead = select is not None and early_select
        out = c.result_arrays(pinned_rows=1 if ahead else None)
        conv = np.empty(out['converged'].shape, dtype=bool)
        picks = None
        ahead = ahead and out.get('_pinned', False)     # (no page-locked block to be had: copied out when waited for, as before)
        if ahead:
            c.prefetch(out)
        if select is not None:
            if ahead:
                idx, row = np.empty((3, n), dtype=np.int32), out['_rows']
                c.select3_prefetch_rows(eager, 1, row)
            else:
                idx, row = c.select3_arrays(1)
            rows = [LazyRows(self, 0, w, n, self.n_omega) for w in range(3)]
            rows[eager]._val = row[0]
            for w in range(3):
                if w != eager:
                    self._pending.append(weakref.ref(rows[w]))
            picks = (idx, rows, row)
        sols = LazySols(n, arrays['alpha'], arrays['sel'] if arrays['alpha'].shape[0] > 1 else None, claim_H, claim_v, out, conv,
                        picks, sel_params, self.n_omega, self.n_s)

        def rest():
            if select is not None and early_select:
                # the scalars first (the fetch waits for the stream): when every alpha converged -- the rule -- the finishing pass
                # has nothing to do and is not called (it would copy the flags a second time and scan them: 0.05 ms per job)
                c.fetch(want_v=False, want_H=False, out=out)
                if not out['converged'].all():
                    if c.finish():                  # (alphas the lock-step layout gave up on: one-chain layout)
                        c.select3_launch(select[0], select[1])
                    c.fetch(want_v=False, want_H=False, out=out)
            else:
                c.finish()
                if select is not None:
                    c.select3_launch(select[0], select[1])
                c.fetch(want_v=False, want_H=False, out=out)
            np.not_equal(out['converged'], 0, out=conv)
            if select is not None:
                c.select3_fetch_rows(first=eager, count=1, idx=picks[0], rows=picks[2])
            ms = c.last_kernel_ms()
            info = dict(kernel_ms=ms, kernel_ms_per_device=[ms], devices=[self.device_ids[0]],
                        n_datasets=[self._n_datasets.get(id(c), 0)], eager=eager)
            info.update(c.last_launch_info())
            if os.environ.get('MAXENT_AMD_AUDIT'):
                corr = c.audit()['corr'].ravel()
                ok = conv.ravel()
                info['audit_max'] = float(np.nanmax(corr[ok])) if ok.any() else 0.0
                info['audit_problems'] = int(ok.sum())
            self.last_info = info
            return sols, info
        return rest

    def _stage_arrays(self, ctx, K, arrays, opts):
        """:meth:`_stage` for a job that comes as arrays (unrotated kernel, no per-element rotation): the same record of what is
        staged, so that a job given one way is recognised when it comes the other way"""
        n = int(arrays['n'])
        sel = arrays['sel']

        def rows(tab):
            tab = np.asarray(tab, dtype=float)
            if tab.shape[0] == 1 or all(np.array_equal(tab[0], t) for t in tab[1:]):
                return np.array(tab[:1])            # (a copy: what is staged must not follow an edit in place)
            return tab[sel]
        G = np.ascontiguousarray(arrays['G'], dtype=float)
        err = np.asarray(arrays['err'], dtype=float)
        staged = dict(n=n, G=G, err=(np.array(err[:1]) if err.shape[0] == 1 else err),
                      D=rows(arrays['D']), alpha=rows(arrays['alpha']), v0=rows(arrays['v0']),
                      kinds=np.asarray(arrays['kinds'])[sel], U_rot=[None] * n,
                      opts=(bytes(opts) if opts is not None else b''), rotated=K.rotation is not None)
        if staged['rotated']:
            raise ValueError('a job given as arrays needs an unrotated kernel')
        held = self.__dict__.setdefault('_staged', {})
        old = held.get(id(ctx))
        if old is not None and old['G'] is not None and old['n'] == n and ctx._n_chain == n and \
                old['opts'] == staged['opts'] and old['rotated'] == staged['rotated'] and \
                len(old['U_rot']) == n and all(u is None for u in old['U_rot']) and \
                all(old[k].shape == staged[k].shape and np.array_equal(old[k], staged[k])
                    for k in ('err', 'D', 'alpha', 'v0', 'kinds')) and old['G'].shape == G.shape:
            if not np.array_equal(old['G'], G):
                # new data on the same grids (every iteration of a self-consistency loop): only their projections change on the
                # device -- data sets, default models, the cut of the chains and their start states stay (mxe_elements_update_data)
                ctx.update_data(G)
                old['G'] = G
            return
        held.pop(id(ctx), None)
        ctx.clear_datasets()
        self.__dict__.setdefault('_n_datasets', {})
        full = lambda a: a if a.shape[0] == n else np.broadcast_to(a, (n, a.shape[1]))
        if staged['err'].shape[0] == 1:
            ds_ids = [ctx.add_dataset(staged['err'][0], None)] * n
            self._n_datasets[id(ctx)] = 1
        else:
            ds_ids, seen = [], []
            for e in staged['err']:
                found = None
                for (e0, i0) in seen:
                    if np.array_equal(e0, e):
                        found = i0
                        break
                if found is None:
                    found = ctx.add_dataset(e, None)
                    seen.append((e, found))
                ds_ids.append(found)
            self._n_datasets[id(ctx)] = len(seen)
        ctx.set_elements(ds_ids, G, full(staged['D']), staged['kinds'])
        ctx.upload_chains(np.arange(n, dtype=np.int32), np.ascontiguousarray(full(staged['alpha'])),
                          np.ascontiguousarray(full(staged['v0'])), opts)
        held[id(ctx)] = staged

    @staticmethod
    def _rows_of(specs, key):
        """the vectors ``key`` of the specs as rows -- ONE row when every spec holds the same array (the element-wise drivers
        hand one default model, alpha mesh, start vector and error array to every element of a worker: a handful of
        distinct objects per launch, compared by contents once each)"""
        vals = list(map(operator.itemgetter(key), specs))
        uniq = {}
        for i in map(id, vals):
            if i not in uniq:
                uniq[i] = len(uniq)
                if len(uniq) > 8:
                    break
        if len(uniq) <= 8 and all(isinstance(v, np.ndarray) for v in (vals[0], vals[-1])):
            firsts = {}
            for v in vals:                          # (the distinct objects, in the order of their first appearance)
                if len(firsts) == len(uniq):
                    break
                firsts.setdefault(id(v), v)
            rows = [np.asarray(v, dtype=float).ravel() for v in firsts.values()]
            if all(r.shape == rows[0].shape for r in rows):
                if all(np.array_equal(rows[0], r) for r in rows[1:]):
                    return np.array(rows[0], dtype=float).reshape(1, -1)       # (a copy: what is staged must not follow an edit in place)
                sel = np.fromiter(map(uniq.__getitem__, map(id, vals)), dtype=np.intp, count=len(vals))
                return np.stack(rows)[sel]
        rows = [np.asarray(v, dtype=float).ravel() for v in vals]
        return np.concatenate(rows).reshape(len(rows), -1)          # (np.stack costs 1 us per row)

    def _stage(self, ctx, K, specs, opts):
        """data sets, elements and chains of one device.  What is staged is remembered (contents, not identities): the
        same job again -- the same object run twice, a parameter of the analyzers changed -- uploads nothing"""
        n, n_tau = len(specs), len(specs[0]['G'])
        same_len = all(len(s['G']) == n_tau for s in specs)
        if same_len:
            # (the element-wise drivers hand ONE error array to all elements of a batch -- the first, which comes from the
            #  worker's own state, and the batches of two workers in one launch have arrays of their own: a handful of objects)
            uniq = {}
            for s in specs:
                uniq.setdefault(id(s['err']), s['err'])
            vals = list(uniq.values())
            if len(vals) <= 8 and all(isinstance(e, np.ndarray) and e.shape == (n_tau,) for e in vals):
                if all(np.array_equal(vals[0], e) for e in vals[1:]):
                    errs = np.array(vals[0], dtype=float).reshape(1, -1)
                else:
                    errs = np.stack([s['err'] for s in specs]).astype(float, copy=False)
            else:
                errs = np.stack([np.asarray(s['err'], dtype=float) * np.ones(n_tau) for s in specs])
        else:
            errs = None
        staged = dict(
            n=n,
            G=np.concatenate([s['G'] for s in specs]).astype(float, copy=False).reshape(n, n_tau) if same_len else None,
            err=errs,
            D=self._rows_of(specs, 'D'), alpha=self._rows_of(specs, 'alpha'), v0=self._rows_of(specs, 'v0'),
            kinds=np.array([s['kind'] for s in specs]),
            U_rot=[s.get('U_rot') for s in specs],
            opts=(bytes(opts) if opts is not None else b''), rotated=K.rotation is not None)
        held = self.__dict__.setdefault('_staged', {})
        old = held.get(id(ctx))
        if old is not None and same_len and old['G'] is not None and old['n'] == n and ctx._n_chain == n and \
                old['opts'] == staged['opts'] and old['rotated'] == staged['rotated'] and \
                len(old['U_rot']) == len(staged['U_rot']) and all(a is b for a, b in zip(old['U_rot'], staged['U_rot'])) and \
                all(old[k].shape == staged[k].shape and np.array_equal(old[k], staged[k])
                    for k in ('err', 'D', 'alpha', 'v0', 'kinds')) and old['G'].shape == staged['G'].shape:
            if not np.array_equal(old['G'], staged['G']):
                ctx.update_data(staged['G'])        # (new data, everything else as staged: mxe_elements_update_data)
                old['G'] = staged['G']
            return
        held.pop(id(ctx), None)
        ctx.clear_datasets()
        ds_ids, seen = [], []
        self.__dict__.setdefault('_n_datasets', {})
        full = lambda a: a if a.shape[0] == n else np.broadcast_to(a, (n, a.shape[1]))
        errs_n = full(errs) if errs is not None else None
        for i, s in enumerate(specs):
            err = errs_n[i] if errs_n is not None else np.asarray(s['err'], dtype=float) * np.ones(len(s['G']))
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
        ctx.set_elements(ds_ids, staged['G'] if same_len else [s['G'] for s in specs], full(staged['D']), staged['kinds'])
        ctx.upload_chains(np.arange(n, dtype=np.int32), np.ascontiguousarray(full(staged['alpha'])),
                          np.ascontiguousarray(full(staged['v0'])), opts)
        held[id(ctx)] = staged

    def _on_devices(self, fn, ranks, wait=True):
        """fn(rank) for every rank: in this thread for one device, one thread per device otherwise (``wait=False``:
        returns the function that joins them)"""
        if len(ranks) <= 1:
            for r in ranks:
                fn(r)
            return (lambda: None) if not wait else None
        pool = self.__dict__.get('_pool')
        if pool is None or pool._max_workers < len(ranks):
            from concurrent.futures import ThreadPoolExecutor
            pool = self.__dict__['_pool'] = ThreadPoolExecutor(max_workers=len(self.ctxs))
        futures = [pool.submit(fn, r) for r in ranks]

        def join():
            for f in futures:
                f.result()                          # (re-raises what a worker raised)
        if not wait:
            return join
        join()

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
        """the claims on the result buffers (:class:`Claim`: all H / all v of a launch; :class:`LazyRows`) that somebody still
        holds and that have not been fetched (a result that was dropped takes its claim on the device buffer with it)"""
        out = []
        for ref in self._pending:
            h = ref()
            if h is not None and (h.val if isinstance(h, Claim) else h._val) is None:
                out.append(h)
        return out

    @_one_at_a_time
    def _materialize_rank(self, rank, only=None):
        """everything of ``rank`` that is still claimed comes to the host (``only`` = 'H' / 'v': that array alone -- and the
        analyzers' rows with H)"""
        alive = self._alive()
        mine = [h for h in alive if h._rank == rank and h._what != 'rows' and (only is None or h._what == only)]
        if mine:
            got = self.ctxs[rank].fetch(want_v=any(h._what == 'v' for h in mine), want_H=any(h._what == 'H' for h in mine))
            for h in mine:
                h.val = got[h._what]
        if only != 'v':
            for h in alive:
                # (the rows of the analyzers go with the arrays: 1 MB each next to H's 100, and what has been brought over has no
                #  claim on the device buffers any more -- a new object on the same grids may take the contexts over)
                if h._rank == rank and h._what == 'rows':
                    h._val = self.ctxs[rank].select3_fetch_rows(first=h._which, count=1, want_index=False)[1][0]
        self._pending = [weakref.ref(h) for h in alive if (h.val if isinstance(h, Claim) else h._val) is None]

    @_one_at_a_time
    def _materialize_rows(self, lazy):
        """the rows of ONE analyzer of the last launch of its device (nothing else moves)"""
        if lazy._val is None:
            if not any(ref() is lazy for ref in self._pending):
                raise RuntimeError('the rows of this analyzer are no longer on the device')      # (cannot happen: materialize_pending runs before every launch)
            lazy._val = self.ctxs[lazy._rank].select3_fetch_rows(first=lazy._which, count=1, want_index=False)[1][0]

    @_one_at_a_time
    def materialize_pending(self):
        for r in sorted(set(h._rank for h in self._alive())):
            self._materialize_rank(r)
        self._pending = []

    @_one_at_a_time
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


atexit.register(BatchSolver.close_pool)
