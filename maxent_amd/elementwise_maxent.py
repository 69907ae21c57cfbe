"""Element-wise MaxEnt for matrix-valued G(tau): the batching boundary.

``ElementwiseMaxEnt`` / ``DiagonalMaxEnt`` / ``PoormanMaxEnt`` keep the
reference's surface and semantics (reference python/elementwise_maxent.py:
58-653): two workers -- diagonal elements with the normal entropy,
off-diagonal ones with the plus-minus entropy --, attribute shadowing onto
both, hermiticity shortcut (i > j skipped), real and imaginary parts as
separate real problems, per-element errors or covariances, and Poorman's
default model D_ij = sqrt(A_ii A_jj) + eps from the analyzed diagonals.

What is different is the execution: the reference runs the elements one after
the other, each with a fresh kernel fill and SVD.  Here
``run_diagonal`` and ``run_offdiagonal`` each collect the alpha scans of all
their elements and hand them to the device as ONE launch of the chain kernel
(one chain per element, :func:`maxent_amd.maxent_loop.solve_elements`); the
kernel matrix is filled and decomposed once.  Poorman's method keeps its
ordering constraint: diagonals (including their analyzers) finish before the
off-diagonal launch is assembled.
"""

from datetime import datetime

import numpy as np

from .default_models import DataDefaultModel
from .logtaker import VerbosityFlags
from .maxent_loop import solve_elements, select_params
from .maxent_result import MaxEntResult
from .tau_maxent import TauMaxEnt


class CallableMethodCheck(object):
    """call the same method on both workers; the results must agree."""

    def __init__(self, name, fun1, fun2):
        self.name, self.fun1, self.fun2 = name, fun1, fun2

    def __call__(self, *args, **kwargs):
        r1 = self.fun1(*args, **kwargs)
        r2 = self.fun2(*args, **kwargs)
        if np.all(r1 == r2):
            return r1
        raise Exception('Element {n} not uniquely defined. Use '
                        'self.maxent_diagonal.{n} or '
                        'self.maxent_offdiagonal.{n}!'.format(n=self.name))


class _LazySpecs(object):
    """the specs of the elements of a batch cut from ONE data array (``_prepare_batch``, direct input): the first element's
    spec as the worker made it, then copies of one template that differ in their data rows -- each made when it is asked for
    (a launch that is staged from the arrays and whose records nobody looks at never asks)"""

    def __init__(self, first_spec, template, g_rows, rest):
        self._first, self._template, self._g, self._rest = first_spec, template, g_rows, rest
        self._lead = 0 if first_spec is None else 1
        self._made = {}

    def __len__(self):
        return self._lead + len(self._rest)

    def __bool__(self):
        return len(self) > 0

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if i < self._lead:
            return self._first
        k = i - self._lead
        if k == 0:
            return self._template
        d = self._made.get(k)
        if d is None:
            g = self._g[self._rest[k]]
            d = self._made[k] = {**self._template, 'G': g, 'G_orig': g}
        return d


class _Picked(object):
    """``seq[i] for i in positions`` without making them"""

    def __init__(self, seq, positions):
        self._seq, self._pos = seq, positions

    def __len__(self):
        return len(self._pos)

    def __iter__(self):
        return (self._seq[i] for i in self._pos)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._seq[i] for i in self._pos[k]]
        return self._seq[self._pos[k]]


class _Chain(object):
    """seqs[n][k] as chain[(n, k)]"""

    def __init__(self, seqs):
        self._seqs = seqs

    def __getitem__(self, nk):
        return self._seqs[nk[0]][nk[1]]


class DeferredLaunch(object):
    """One launch of ``ElementwiseMaxEnt.run()`` on array input as the :class:`MaxEntResult` holds it until somebody looks at
    something per element (``MaxEntResult.add_deferred``): the launch's arrays as they came off the device, and ``settle``,
    which builds the records and analyses of its scans the way ``run()`` always did.  ``field`` / ``A_out`` serve the whole-matrix
    fields from the arrays (the same views and the same arithmetic as the assembly from records: the same bits)."""
    covers = None

    def __init__(self, keys, sols, alpha, omega, maps, which_of, settle):
        self.keys, self.sols, self.alpha, self.omega = keys, sols, alpha, omega
        self._maps, self._which_of, self._settle = maps, which_of, settle

    def settle(self, result):
        fn, self._settle = self._settle, None
        if fn is not None:
            fn(result)

    def _identity_delta(self):
        """delta of the output map when every scan's A is H / delta with the same delta, else None"""
        first = self._maps[0]
        if type(first).__name__ != 'IdentityA_of_H':
            return None
        d0 = getattr(getattr(first, '_omega', None), 'delta', None)
        for m in self._maps[1:]:
            if m is first:
                continue
            d = getattr(getattr(m, '_omega', None), 'delta', None)
            if type(m) is not type(first) or d is None or d0 is None or np.shape(d) != np.shape(d0) or not np.array_equal(d, d0):
                return None
        return d0

    def field(self, name, ems):
        sols = self.sols
        n, na = sols.n, sols.n_alpha
        if name in ('chi2', 'S', 'Q'):
            v = sols.arrays[name].reshape(tuple(ems) + (na,))
            v.setflags(write=False)
            return v
        if name == 'n_iter':
            return sols.arrays['n_iter'].astype(float).reshape(tuple(ems) + (na,))
        if name == 'converged':
            return sols.conv.astype(float).reshape(tuple(ems) + (na,))
        if name == 'H':
            whole = sols.claim_H.materialize()
            v = whole.reshape(tuple(ems) + (na, sols.n_omega))
            v.setflags(write=False)
            return v
        if name == 'A':
            delta = self._identity_delta()
            if delta is None:
                return None
            from . import device
            from .maxent_result import _by_rows
            H = self.field('H', ems)
            out = device.pinned_empty(H.shape)
            _by_rows(lambda a, b: np.divide(a, delta, out=b), H, out)
            return out
        return None

    def A_out(self, analyzer_name, ems):
        which = self._which_of.get(analyzer_name)
        picks = self.sols.picks
        if which is None or picks is None:
            return None
        idx = picks[0][which]
        delta = self._identity_delta()
        if delta is None or np.any(idx < 0):
            return None
        rows = np.asarray(picks[1][which])
        return (rows / delta).reshape(tuple(ems) + (rows.shape[-1],))


class PendingRun(object):
    """what :meth:`ElementwiseMaxEnt.run_async` returns: ``result()`` waits for the device and completes the run"""

    def __init__(self, owner, finish):
        self._owner, self._finish = owner, finish
        self._result = owner.maxent_result if finish is None else None
        self.done = finish is None

    def result(self):
        if not self.done:
            finish, self._finish = self._finish, None
            try:
                self._result = finish()
            finally:
                self.done = True
                if self._owner.__dict__.get('_pending_run') is self:
                    object.__setattr__(self._owner, '_pending_run', None)     # (the object does not hold on to its last result)
                self._owner = None
        return self._result


def run_many(objects, in_flight=None, same_cut=True):
    """``[ew.run() for ew in objects]`` with the jobs IN FLIGHT together: every object is prepared, staged and launched before
    the first is waited for (one thread, no wait between the launches), then each is completed in turn while the kernels of
    the others run.  ``in_flight``: jobs on the GPU at a time (default: all of them, at most eight).  Every job is cut into
    pieces as ``run()`` cuts it, so every field of every result is, bit for bit, what the sequential calls return.
    ``same_cut=False`` cuts a job for the company it has (``mxe_opts.in_flight``: fewer cold-started pieces, 0.65 instead of
    0.74 ms of GPU time per 16 x 16 x 100-alpha job with four in flight): other iterates, the same minimisers within the
    stopping tolerance (A_out to ~1e-10) -- through this API the host's work per job hides the difference (four jobs: 3.8 ms
    either way); a caller of the C interface with its own pipeline sees it (bench.py, value_in_flight)."""
    objects = list(objects)
    n = max(1, min(8, len(objects) if in_flight is None else int(in_flight)))
    cut = 1 if same_cut else n
    results, window = [None] * len(objects), []
    for k, ew in enumerate(objects):
        if len(window) >= n:
            j, h = window.pop(0)
            results[j] = h.result()
        window.append((k, ew.run_async(in_flight=cut) if hasattr(ew, 'run_async') else _Ran(ew.run())))
    for j, h in window:
        results[j] = h.result()
    return results


class _Ran(object):
    def __init__(self, result):
        self._r = result

    def result(self):
        return self._r


class ElementwiseMaxEnt(object):
    maxent_diagonal = None
    maxent_offdiagonal = None

    def __init__(self, use_hermiticity=True, use_complex=False, n_gpus=None, device_ids=None, **kwargs):
        """``n_gpus`` / ``device_ids``: shard the matrix elements over several GPUs of this process
        (element e of a phase on device e mod N, one gather).  Default: device 0."""
        if device_ids is None and n_gpus is not None:
            device_ids = tuple(range(int(n_gpus)))
        object.__setattr__(self, 'device_ids', None if device_ids is None else tuple(device_ids))
        self.maxent_diagonal = TauMaxEnt(**kwargs)
        self.maxent_offdiagonal = TauMaxEnt(cost_function='plusminus',
                                            **kwargs)
        self.set_G_element = None
        self.determine_shape = None
        self.G_mat = None
        self.maxent_result = None
        self.use_hermiticity = use_hermiticity
        self.use_complex = use_complex
        self.last_launches = []

    # ---- attribute shadowing onto both workers -------------------------
    def __getattr__(self, name):
        d = getattr(object.__getattribute__(self, 'maxent_diagonal'), name)
        o = getattr(object.__getattribute__(self, 'maxent_offdiagonal'), name)
        if callable(d) and callable(o):
            return CallableMethodCheck(name, d, o)
        if np.all(d == o):
            return d
        raise Exception('Element {n} not uniquely defined. Use '
                        'self.maxent_diagonal.{n} or '
                        'self.maxent_offdiagonal.{n}!'.format(n=name))

    def __setattr__(self, name, value):
        if hasattr(self.maxent_diagonal, name) and \
                hasattr(self.maxent_offdiagonal, name):
            setattr(self.maxent_offdiagonal, name, value)
            setattr(self.maxent_diagonal, name, value)
        else:
            object.__setattr__(self, name, value)

    # ---- result object ---------------------------------------------------
    def prepare_maxent_result(self, overwrite=False):
        if self.maxent_result is None or overwrite:
            self.maxent_result = MaxEntResult(
                matrix_structure=self.determine_shape(self.G_mat),
                element_wise=True,
                use_hermiticity=self.use_hermiticity,
                complex_elements=self.use_complex)

    # ---- single element (one chain) ---------------------------------------
    def _worker_for(self, element):
        return self.maxent_diagonal if element[0] == element[1] \
            else self.maxent_offdiagonal

    def _load_element(self, worker, element, re):
        i, j = element
        self.set_G_element(worker, self.G_mat, (i, j),
                           True if i == j else re)
        self.put_error(worker, self.get_error((i, j)))

    def run_element(self, element, re=True):
        """one matrix element, like the reference's ``run_element``
        (elementwise_maxent.py:170-221)."""
        self.prepare_maxent_result(overwrite=False)
        i, j = element
        worker = self._worker_for(element)
        if i != j and self.use_hermiticity and i > j:
            worker.logtaker.message(
                VerbosityFlags.ElementInfo,
                'Element {} {} not calculated, can be determined from '
                'hermiticity'.format(i, j))
            return self.maxent_result
        worker.logtaker.message(VerbosityFlags.ElementInfo,
                                'Calling MaxEnt for element {} {}'.format(i, j))
        self._load_element(worker, element, re)
        worker.run(result=self.maxent_result, matrix_element=(i, j),
                   complex_index=0 if re else 1)
        return self.maxent_result

    # ---- batched phases --------------------------------------------------
    def _run_batch(self, worker, jobs, per_job_D=None):
        """``jobs``: list of (element, re).  All scans in one launch."""
        batch = self._prepare_batch(worker, jobs, per_job_D)
        self._solve_batches([batch])
        return self._finish_batch(batch)

    def _prepare_batch(self, worker, jobs, per_job_D=None, defer_last_load=None):
        """the specs of the elements of ``jobs`` that are to be solved (the others go to the result's zero
        elements); leaves ``worker`` loaded with the last element, as the reference does"""
        self.prepare_maxent_result(overwrite=False)
        res = self.maxent_result
        loop = worker.maxent_loop
        specs, live = [], []
        self._share_decomposition()
        direct = self._direct_input(worker) and per_job_D is None
        template = None
        below = err_same = g_rows = first_spec = g_mat = arrays = None
        if direct and len(jobs) > 1:
            # (all elements at once: the data vectors as the rows of ONE array -- real part, or imaginary part of an off-diagonal
            #  element's second scan --, which of them are below the threshold; one error array when it is the same for all)
            Gm = self.G_mat[1]
            plan = self.__dict__.get('_plans', {}).get(id(jobs))
            if plan is not None and plan['jobs'] is jobs:
                ii, jj, real_part, all_real = plan['ii'], plan['jj'], plan['real_part'], plan['all_real']
            else:
                ii = np.fromiter((e[0] for e, _ in jobs), dtype=np.intp, count=len(jobs))
                jj = np.fromiter((e[1] for e, _ in jobs), dtype=np.intp, count=len(jobs))
                real_part = np.fromiter((bool(re) for _, re in jobs), dtype=bool, count=len(jobs)) | (ii == jj)
                all_real = bool(real_part.all())
            Gsel = Gm[ii, jj]
            if np.iscomplexobj(Gsel):
                g_rows = np.where(real_part[:, None], Gsel.real, Gsel.imag).astype(float, copy=False)
            else:
                g_rows = np.array(Gsel, dtype=float)
                if not all_real:
                    g_rows[~real_part] = 0.0         # (the imaginary part of real data)
            with np.errstate(all='ignore'):
                below = (np.max(np.abs(g_rows), axis=-1) < loop.G_threshold).tolist()
            g_mat = g_rows
            g_rows = list(g_rows)                    # (row views, made in one go)
            e0 = self.get_error(tuple(jobs[0][0]))
            if isinstance(self.error, float) or len(np.shape(self.error)) == self.error_dimension:
                err_same = np.asarray(e0, dtype=float) * np.ones(np.shape(Gm)[-1])
            # a worker that already holds this tau grid (an object that has run before) is not sent through its setters for the FIRST
            # element either: nothing of a spec but the data depends on the element that is loaded, and the state the phase leaves
            # behind is that of the last element, loaded below (two loads per run instead of four: 0.1 ms)
            try:
                skip_first = (err_same is not None and loop.G is not None and loop.err is not None and
                              np.array_equal(np.asarray(worker.tau), np.asarray(self.G_mat[0])))
            except Exception:
                skip_first = False
        else:
            skip_first = False
        for n, (element, re) in enumerate(jobs):
            cidx = 0 if re else 1
            if direct and (n > 0 or skip_first):
                break               # (array input, plain errors, unrotated kernel: every further element below, straight from the arrays)
            if per_job_D is not None:
                worker.set_D(per_job_D[n])
            self._load_element(worker, element, re)
            if n == 0:
                self._share_decomposition()      # (the worker has its tau grid now)
            if loop.below_threshold():
                key = tuple(element) + ((cidx,) if self.use_complex else ())
                res._zero_elements.append(key)
                worker.logtaker.error_message(
                    'G below threshold, not performing the calculation.')
                continue
            spec = loop.make_spec()
            spec['A_map'] = loop.A_of_H
            if n == 0:
                first_spec = spec
            specs.append(spec)
            live.append((element, cidx))
        if direct and len(jobs) > 1:
            # the specs of the further elements: copies of ONE template with their rows of the data array (the worker keeps the
            # first element's state until the last is loaded, the state the reference leaves behind)
            rest = []
            for n in range(0 if skip_first else 1, len(jobs)):
                if below[n]:
                    element, re = jobs[n]
                    res._zero_elements.append(tuple(element) + (((0 if re else 1),) if self.use_complex else ()))
                    worker.logtaker.error_message('G below threshold, not performing the calculation.')
                else:
                    rest.append(n)
            if rest:
                n0 = rest[0]
                g, e0 = g_rows[n0], self.get_error(tuple(jobs[n0][0]))
                # (everything but the data is what the first element's spec holds, when that was made a moment ago)
                template = loop.make_spec(G=g, err=e0) if first_spec is None else loop.spec_like(first_spec, g, e0)
                template['A_map'] = loop.A_of_H
                template['G'] = template['G_orig'] = g
                if err_same is not None:
                    # (ONE error array for every spec of the batch: BatchSolver._stage then stages one row instead of stacking
                    #  256 -- it was 1 ms of the 4.7 ms of a run on an object that has run before)
                    template['err'] = err_same
                    # the job of this batch as ARRAYS, and its specs made when somebody asks (round 5: 256 dict copies per run
                    # that a launch staged from the arrays never reads)
                    specs = _LazySpecs(first_spec, template, g_rows, rest)
                    lead = first_spec if first_spec is not None else template
                    arrays = dict(G=(g_mat if (first_spec is None and len(rest) == len(jobs)) else
                                     g_mat[([0] if first_spec is not None else []) + rest]), err=err_same, D=lead['D'],
                                  alpha=lead['alpha'], v0=lead['v0'], kind=lead['kind'])
                else:
                    specs.append(template)
                    specs.extend([loop.spec_like(template, g_rows[n], self.get_error(tuple(jobs[n][0]))) for n in rest[1:]])
                whole = len(live) + len(rest) == len(jobs) and plan is not None and plan['jobs'] is jobs
                if whole and 'live' in plan:
                    live = plan['live']              # (every element of the phase is solved: one list for every run)
                else:
                    live.extend([(jobs[n][0], 0 if jobs[n][1] else 1) for n in rest])
                    if whole:
                        plan['live'] = live
                if rest[-1] == len(jobs) - 1:
                    if defer_last_load is not None and arrays is not None:
                        # (the state the phase leaves behind -- the worker loaded with its last element, as in the reference -- does
                        #  not enter the launch: the caller makes it up behind the launch, beside the kernel)
                        defer_last_load.append((worker, jobs[-1][0], jobs[-1][1]))
                    else:
                        self._load_element(worker, jobs[-1][0], jobs[-1][1])
        # (the keys of the result's records, made once: they were made three times per element -- 0.2 ms of a 16 x 16 run)
        keys = None
        if direct and len(jobs) > 1:
            plan = self.__dict__.get('_plans', {}).get(id(jobs))
            if plan is not None and plan.get('live') is live:
                flags = (res.matrix_structure is None, bool(res.element_wise), bool(res.complex_elements))
                if plan.get('keys_flags') != flags:
                    plan['keys'], plan['keys_flags'] = res._keys(live), flags
                keys = plan['keys']
        if keys is None:
            keys = res._keys(live)
        return dict(worker=worker, specs=specs, live=live, keys=keys, arrays=arrays)

    def _solve_batches(self, batches, defer=False, in_flight=0):
        """(``defer``: launch only and return the function that waits for the device and completes the batches -- the caller
        launches other objects meanwhile, :meth:`run_async`)
        one launch for all batches whose workers share the decomposition of the kernel, the minimiser settings
        and the alpha count (plain ElementwiseMaxEnt: diagonal and off-diagonal elements together, as bench.py
        times them); otherwise one launch per batch"""
        batches = [b for b in batches if b['specs']]
        groups = []
        for b in batches:
            for g in groups:
                if self._same_launch(g[0], b):
                    g.append(b)
                    break
            else:
                groups.append([b])
        res = self.maxent_result
        ends = []
        for g in groups:
            loop = g[0]['worker'].maxent_loop
            # the scans of the launch in the order of the result's matrix (row major): what comes off the device in one
            # copy then IS the (M, N, n_alpha, n_omega) array of the result -- MaxEntResult._assemble takes it as a view
            # (the order of a launch and the index arrays that go with it depend on the key lists of its batches alone; the lists of
            #  the full phases are one object per object and phase -- _jobs --: their launch plan is made once)
            lplans = self.__dict__.setdefault('_launch_plans', {})
            lkey = tuple(id(b['keys']) for b in g)
            lp = lplans.get(lkey)
            if lp is None or not all(k0 is b['keys'] for k0, b in zip(lp['keys_of'], g)):
                where = [(key, n, k) for n, b in enumerate(g) for k, key in enumerate(b['keys'])]
                try:
                    where.sort(key=lambda t: t[0])
                except TypeError:
                    pass
                offs = np.cumsum([0] + [len(b['keys']) for b in g])
                src = np.fromiter((offs[n] + k for (_, n, k) in where), dtype=np.intp, count=len(where))
                inv = np.empty(len(where), dtype=np.intp)
                inv[src] = np.arange(len(where))            # position in the launch of scan k of batch n: inv[offs[n] + k]
                lp = dict(keys_of=[b['keys'] for b in g], where=where, offs=offs, src=src, inv=inv,
                          sel=np.fromiter((n for (_, n, k) in where), dtype=np.intp, count=len(where)),
                          nk=[(n, k) for (_, n, k) in where], launch_keys=[key for (key, _, _) in where])
                plans_held = self.__dict__.get('_plans', {})
                if all(any(p.get('keys') is b['keys'] for p in plans_held.values()) for b in g):
                    if len(lplans) > 8:
                        lplans.clear()
                    lplans[lkey] = lp                       # (only for key lists that live as long as this object does)
            where = lp['where']
            all_arrays = all(b.get('arrays') is not None for b in g)
            specs = _Picked(_Chain([b['specs'] for b in g]), lp['nk']) if all_arrays else \
                [g[n]['specs'][k] for (_, n, k) in where]
            t0 = datetime.now()
            for b in g:
                b['t_start'] = t0              # (into the result's table of start times with the records: _finish_batch)
            for b in g:
                b['sols'] = [None] * len(b['specs']) if not all_arrays else None

            def hand_out(sols, g=g, where=where):
                for b in g:
                    if b['sols'] is None:
                        b['sols'] = [None] * len(b['specs'])
                for sol, (_, n, k) in zip(sols, where):
                    g[n]['sols'][k] = sol

            def records_while_the_kernel_runs(sols, g=g):
                # (the result dicts hold their arrays already, the device fills them behind this: the records of the scans --
                #  0.3 ms of dictionaries for 256 elements -- cost nothing next to a kernel of 0.8 ms)
                hand_out(sols)
                for b in g:
                    b['records'] = b['worker'].maxent_loop.make_records(b['specs'], b['sols'])
            plain = all(b['worker'].maxent_loop.probability is None for b in g)
            select = select_params(loop.analyzers)
            lazy = (defer and plain and select is not None and all(b.get('arrays') is not None for b in g) and
                    (self.device_ids is None or len(self.device_ids) == 1) and hasattr(loop.minimizer, 'to_opts') and
                    not any(bool(b['worker'].logtaker.verbose & (VerbosityFlags.ElementInfo | VerbosityFlags.AlphaLoop)) for b in g))
            if lazy:
                # the launch from ARRAYS: the data rows of the batches in the order of the launch, one row of D / alpha / v0 /
                # kind per batch; nothing per scan is built here -- records and analyses when somebody looks (DeferredLaunch)
                offs, src, sel, inv = lp['offs'], lp['src'], lp['sel'], lp['inv']
                arrays = dict(n=len(where), G=np.concatenate([b['arrays']['G'] for b in g])[src],
                              err=np.asarray(g[0]['arrays']['err'], dtype=float).reshape(1, -1), sel=sel,
                              D=np.stack([b['arrays']['D'] for b in g]), alpha=np.stack([b['arrays']['alpha'] for b in g]),
                              v0=np.stack([b['arrays']['v0'] for b in g]), kinds=np.array([b['arrays']['kind'] for b in g]))
                if not all(np.array_equal(g[0]['arrays']['err'], b['arrays']['err']) for b in g[1:]):
                    arrays['err'] = np.stack([b['arrays']['err'] for b in g])[sel]
                wait = solve_elements(loop.K, None, loop.minimizer, device_id=loop.device_id, device_ids=self.device_ids,
                                      chi2_factor=loop.cost_function.chi2_factor, select=select, defer=True, in_flight=in_flight,
                                      arrays=arrays)
                for n, b in enumerate(g):
                    b['positions'] = inv[offs[n]:offs[n + 1]]
                    b['launch_keys'] = lp['launch_keys']
                    b['launch_select'] = select
                    b['launch_batches'] = len(g)          # (not g itself: a batch that holds the list it is in is a cycle, and its
                    #  claim on the device buffers would live until the collector runs)
            else:
                for b in g:
                    b['arrays'] = None
                wait = solve_elements(loop.K, specs, loop.minimizer,
                                      device_id=loop.device_id, device_ids=self.device_ids,
                                      want_logdet=loop.probability is not None,
                                      chi2_factor=loop.cost_function.chi2_factor,
                                      select=select,
                                      while_waiting=records_while_the_kernel_runs if plain else None,
                                      defer=True, in_flight=in_flight)

            def end(g=g, wait=wait, t0=t0, specs=specs, hand_out=hand_out, lazy=lazy):
                sols, info = wait()
                t1 = datetime.now()
                self.last_launches.append(info)
                for b in g:
                    b.update(info=info, t0=t0, t1=t1, per_alpha=(t1 - t0) / max(1, len(specs) * len(specs[0]['alpha'])))
                if lazy:
                    for b in g:
                        b['sols'] = _Picked(sols, b['positions'])
                        b['launch'] = sols
                elif not all('records' in b for b in g):
                    hand_out(sols)
            if defer:
                ends.append(end)
            else:
                end()

        def end_all():
            for e in ends:
                e()
        return end_all if defer else None

    def _same_launch(self, a, b):
        la, lb = a['worker'].maxent_loop, b['worker'].maxent_loop
        if la is lb:
            return True
        Ka, Kb = la.K, lb.K
        try:
            same = (Ka._U is not None and Ka._U is Kb._U and Ka._S is Kb._S and Ka._V is Kb._V and
                    Ka.rotation is None and Kb.rotation is None and
                    len(a['specs'][0]['alpha']) == len(b['specs'][0]['alpha']) and
                    (la.probability is None) == (lb.probability is None) and
                    la.device_id == lb.device_id and
                    la.cost_function.chi2_factor == lb.cost_function.chi2_factor and
                    bytes(la.minimizer.to_opts()) == bytes(lb.minimizer.to_opts()))
        except Exception:
            return False
        return bool(same)

    def _finish_batch(self, batch):
        """records and analyzers of a solved batch"""
        res = self.maxent_result
        if not batch['specs']:
            return res
        self._finish_now(batch)
        self._finish_records(batch, res)
        return res

    def _finish_now(self, batch):
        """what of the end of a batch is not a record: the log lines, the state the minimiser is left in, the timing line"""
        res = self.maxent_result
        worker, specs, live = batch['worker'], batch['specs'], batch['live']
        loop = worker.maxent_loop
        if res._default_analyzer_name is None and loop.analyzers:
            res._default_analyzer_name = loop.analyzers[0].name
        sols, info = batch['sols'], batch['info']
        talk = bool(worker.logtaker.verbose & (VerbosityFlags.ElementInfo | VerbosityFlags.AlphaLoop))
        if talk:
            for sol, (element, cidx) in zip(sols, live):
                worker.logtaker.message(
                    VerbosityFlags.ElementInfo,
                    'Element {} {}{}'.format(element[0], element[1],
                                             '' if cidx == 0 else ' (imaginary part)'))
                loop.log_alpha_lines(sol)
        elif len(sols):
            # (one reduction over the rows of the launch's count array, not one numpy call per element: 1.2 ms for 256 elements)
            if batch.get('launch') is not None:
                total = int(batch['launch'].arrays['n_iter'][batch['positions']].sum())
            else:
                total = int(np.asarray([x['n_iter'] for x in sols]).sum())
            loop.note_minimizer_state(sols[-1], total)
        worker.logtaker.message(
            VerbosityFlags.Timing,
            '{} alpha scans x {} alpha in one launch: kernel {:.3f} ms',
            len(specs), len(specs[0]['alpha']), info['kernel_ms'])

    def _finish_records(self, batch, res):
        """the records of a solved batch into ``res``, and its analyzers (for a launch that was handed to the result as ONE
        object -- DeferredLaunch -- this runs when somebody looks at something per element)"""
        worker, specs, live = batch['worker'], batch['specs'], batch['live']
        loop = worker.maxent_loop
        sols, t1, per_alpha = batch['sols'], batch['t1'], batch['per_alpha']
        records = batch.pop('records', None)
        if records is None:
            records = loop.make_records(specs, sols)
        times = {}
        for rec in records:
            X = len(rec['alpha'])
            if X not in times:
                times[X] = (per_alpha,) * X        # (one immutable tuple for the records of a launch; MaxEntResult.run_times hands out lists)
            rec['run_times'] = times[X]
        keys = res.add_batch_results(records, live, t_start=batch.get('t_start'), t_end=t1, keys=batch.get('keys'))
        # analyzers after every record of the batch is in (adding a record drops the assembled-array cache
        # of the result); the rows of A they select come off the device in one go
        res.analyze_batch(loop.analyzers, keys)

    def _finish_deferred(self, batches):
        """the end of ``run()`` for batches that were ONE launch from arrays: the result gets the launch as one object, the records
        of its scans when somebody looks.  False when that does not apply (then: :meth:`_finish_batch`)."""
        res = self.maxent_result
        live = [b for b in batches if b['specs']]
        if not live or any(b.get('launch') is None for b in live) or any(b['launch'] is not live[0]['launch'] for b in live) or \
                live[0]['launch_batches'] != len(live) or res.__dict__.get('_records_store') or res.__dict__.get('_deferred'):
            return False
        from .analyzers import LineFitAnalyzer, Chi2CurvatureAnalyzer, EntropyAnalyzer
        select = live[0]['launch_select']
        which_of = None
        for b in live:
            cur = {}
            for a in b['worker'].maxent_loop.analyzers:
                w = (0 if (type(a) is LineFitAnalyzer and a.linefit_deg == select[0]) else
                     1 if (type(a) is Chi2CurvatureAnalyzer and a.gamma == select[1]) else
                     2 if type(a) is EntropyAnalyzer else None)
                cur[a.name] = None if a.name in cur else w          # (two analyzers of one name: the general way)
            which_of = cur if which_of is None else {k: v for k, v in which_of.items() if cur.get(k, -1) == v}
        which_of = {k: v for k, v in (which_of or {}).items() if v is not None}
        for b in live:
            self._finish_now(b)
        lead = live[0]['worker'].maxent_loop
        first = live[0]['launch_keys'][0]
        first_batch = next(b for b in live if b['keys'] and b['keys'][0] == first)

        def settle(result, live=live):
            for b in live:
                self._finish_records(b, result)
        res.add_deferred(DeferredLaunch(live[0]['launch_keys'], live[0]['launch'], first_batch['arrays']['alpha'], lead.omega,
                                        [b['worker'].maxent_loop.A_of_H for b in live], which_of, settle))
        return True

    def _direct_input(self, worker):
        """G(tau) came as one array and the errors are plain (no covariance): specs can be cut from the
        arrays without sending every element through the worker's setters"""
        return (getattr(self, '_array_input', False) and self.error_dimension == 1 and
                worker.K.rotation is None and not isinstance(self.error, str))

    def _share_decomposition(self):
        """the two workers have kernels of their own; where these are the same matrix (same class, tau,
        omega, beta, no preblur on one side only) the second takes the SVD of the first instead of
        repeating it"""
        a, b = self.maxent_diagonal.K, self.maxent_offdiagonal.K
        if a is b or type(a) is not type(b) or not hasattr(a, 'tau') or a.rotation is not None or b.rotation is not None:
            return
        try:
            same = (np.array_equal(np.asarray(a.tau), np.asarray(b.tau)) and
                    np.array_equal(np.asarray(a.omega), np.asarray(b.omega)) and a.beta == b.beta and
                    a.svd_backend == b.svd_backend)
        except Exception:
            return
        if not same:
            return
        src, dst = (a, b) if b._U is None else ((b, a) if a._U is None else (None, None))
        if src is None:
            return
        src.S
        dst._U, dst._S, dst._V = src._U, src._S, src._V
        dst._last_threshold = src._last_threshold

    def _jobs(self, which):
        """the (element, re) list of a phase -- ONE list object per (phase, shape, hermiticity, complex) of this object, with the
        index arrays cut from it (``_plans``): the lists and arrays of a 16 x 16 matrix cost 0.15 ms per run to make again"""
        key = (which, tuple(self.shape), bool(self.use_hermiticity), bool(self.use_complex))
        plans = self.__dict__.setdefault('_plans', {})
        plan = plans.get(key)
        if plan is None:
            if which == 'diag':
                jobs = [((i, i), True) for i in range(self.shape[0])]
            else:
                jobs = []
                for i in range(self.shape[0]):
                    for j in range(self.shape[1]):
                        if i == j or (self.use_hermiticity and i > j):
                            continue
                        for re in ([True, False] if self.use_complex else [True]):
                            jobs.append(((i, j), re))
            ii = np.fromiter((e[0] for e, _ in jobs), dtype=np.intp, count=len(jobs))
            jj = np.fromiter((e[1] for e, _ in jobs), dtype=np.intp, count=len(jobs))
            real_part = np.fromiter((bool(re) for _, re in jobs), dtype=bool, count=len(jobs)) | (ii == jj)
            plan = plans[key] = dict(jobs=jobs, ii=ii, jj=jj, real_part=real_part, all_real=bool(real_part.all()))
            plans[id(jobs)] = plan
        return plan['jobs']

    def _diag_jobs(self):
        return self._jobs('diag')

    def _offdiag_jobs(self):
        return self._jobs('off')

    def run_diagonal(self):
        """all diagonal elements (reference elementwise_maxent.py:223-242)."""
        self.maxent_diagonal.logtaker.message(
            VerbosityFlags.ElementInfo, 'Calculating diagonal elements.')
        res = self._run_batch(self.maxent_diagonal, self._diag_jobs())
        self._mark_imaginary_diagonal(res)
        return res

    def run_offdiagonal(self):
        """all off-diagonal elements (reference elementwise_maxent.py:244-268)."""
        self.maxent_offdiagonal.logtaker.message(
            VerbosityFlags.ElementInfo, 'Calculating off-diagonal elements.')
        return self._run_batch(self.maxent_offdiagonal, self._offdiag_jobs())

    def run(self):
        """all elements (reference elementwise_maxent.py:270-285)"""
        return self.run_async().result()

    def run_async(self, in_flight=1):
        """Start :meth:`run` and return at once: the data are staged and the kernel is LAUNCHED, nothing is waited for.
        ``handle.result()`` waits, builds records and analyzers and returns the :class:`MaxEntResult` -- the one ``run()``
        returns, field for field.  In between the caller is free: a self-consistency loop that continues several Green
        functions per iteration (reference: ``ElementwiseMaxEnt.run()`` once per job, elementwise_maxent.py:270-285) starts
        them all and collects them afterwards,

            handles = [ew.run_async(in_flight=len(jobs)) for ew in jobs]
            results = [h.result() for h in handles]          # = maxent_amd.run_many(jobs)

        and the GPU works on all of them side by side (every object gets device contexts -- a stream -- of its own) while
        the host prepares the next and finishes the previous one.  ``in_flight``: how many jobs the caller keeps in flight
        (``mxe_opts.in_flight``): each is then cut into fewer cold-started pieces, n of them fill the GPU together (four 16 x 16
        x 100-alpha jobs: 0.65 instead of 0.83 ms of GPU time each).  The answers do not depend on it beyond the stopping
        tolerance.  A subclass with phases of its own (PoormanMaxEnt: the off-diagonal elements need the finished diagonal
        ones) runs to the end in this call."""
        cls = type(self)
        if (cls.run_diagonal is not ElementwiseMaxEnt.run_diagonal or
                cls.run_offdiagonal is not ElementwiseMaxEnt.run_offdiagonal):
            self.run_diagonal()              # (a subclass with phases of its own: PoormanMaxEnt needs the
            self.run_offdiagonal()           #  diagonal results before the off-diagonal elements start)
            return PendingRun(self, None)
        pending = self.__dict__.get('_pending_run')
        if pending is not None and not pending.done:
            raise RuntimeError('this object has a run in flight: take its result() first')
        # both phases in ONE launch where the two workers share the kernel's decomposition
        self.maxent_diagonal.logtaker.message(VerbosityFlags.ElementInfo, 'Calculating diagonal elements.')
        later = []
        diag = self._prepare_batch(self.maxent_diagonal, self._diag_jobs(), defer_last_load=later)
        self.maxent_offdiagonal.logtaker.message(VerbosityFlags.ElementInfo, 'Calculating off-diagonal elements.')
        off = self._prepare_batch(self.maxent_offdiagonal, self._offdiag_jobs(), defer_last_load=later)
        def load_last():                              # (the workers as the reference leaves them: loaded with their last element)
            while later:
                worker, element, re = later.pop(0)
                self._load_element(worker, element, re)
        try:
            wait = self._solve_batches([diag, off], defer=True, in_flight=in_flight)
        except BaseException:
            load_last()
            raise

        def finish():
            # (before the wait, beside the kernel -- and not between this job's launch and the next job's: with four jobs in flight
            #  the 0.08 ms per job delayed every later launch)
            load_last()
            wait()
            if self._finish_deferred([diag, off]):
                self._mark_imaginary_diagonal(self.maxent_result)
                return self.maxent_result
            res = self._finish_batch(diag)
            self._mark_imaginary_diagonal(res)
            return self._finish_batch(off)
        handle = PendingRun(self, finish)
        object.__setattr__(self, '_pending_run', handle)
        return handle

    def _mark_imaginary_diagonal(self, res):
        if self.use_complex:
            for i in range(self.shape[0]):
                if (i, i, 1) not in res._zero_elements:
                    res._zero_elements.append((i, i, 1))

    # ---- input ----------------------------------------------------------------
    def set_G(self, G_mat, set_G_element, determine_shape):
        """generic entry: ``set_G_element(maxent, G_mat, elem, re)`` feeds one
        element to a worker (reference elementwise_maxent.py:287-315)."""
        self.G_mat = G_mat
        self.set_G_element = set_G_element
        self.determine_shape = determine_shape
        self.maxent_result = None
        object.__setattr__(self, '_array_input', False)

    def set_G_tau(self, *args, **kwargs):
        raise NotImplementedError('set_G_tau needs TRIQS Green functions; '
                                  'use set_G_tau_data')

    set_G_iw = set_G_tau

    def set_G_tau_data(self, tau, G_tau, *args, **kwargs):
        """``G_tau``: (M, N, T) array (reference elementwise_maxent.py:373-395)."""
        def feed(maxent, G_mat, elem, re):
            g = G_mat[1][elem]
            maxent.set_G_tau_data(G_mat[0], np.real(g) if re else np.imag(g),
                                  *args, **kwargs)
        self.set_G((tau, G_tau), feed, lambda G_mat: G_mat[1].shape[:2])
        object.__setattr__(self, '_array_input', not args and not kwargs)

    def set_G_tau_filename_pattern(self, filename, dimension, tau_col=0,
                                   G_col_re=1, G_col_im=2, *args, **kwargs):
        """one file per element, name with ``{i}`` and ``{j}``
        (reference elementwise_maxent.py:397-436)."""
        def feed(maxent, G_mat, elem, re):
            maxent.set_G_tau_file(G_mat.format(i=elem[0], j=elem[1]), tau_col,
                                  G_col_re if re else G_col_im, *args, **kwargs)
        self.set_G(filename, feed, lambda G_mat: dimension)

    def set_G_tau_filenames(self, filenames, tau_col=0, G_col_re=1,
                            G_col_im=2, *args, **kwargs):
        """2-d array of file names (reference elementwise_maxent.py:438-470)."""
        def feed(maxent, G_mat, elem, re):
            maxent.set_G_tau_file(G_mat[elem[0]][elem[1]], tau_col,
                                  G_col_re if re else G_col_im, *args, **kwargs)
        self.set_G(filenames, feed, lambda G_mat: np.shape(G_mat))

    def set_error(self, error):
        """float, (T,) or (M, N, T) (reference elementwise_maxent.py:472-487)."""
        self.error = error
        self.error_dimension = 1
        self.put_error = lambda maxent, err: maxent.set_error(err)

    def get_error(self, elem):
        if isinstance(self.error, float):
            return self.error
        if len(np.shape(self.error)) == self.error_dimension:
            return self.error
        return self.error[elem]

    def set_cov(self, cov):
        """(T, T) or (M, N, T, T) (reference elementwise_maxent.py:502-515)."""
        self.error_dimension = 2
        self.error = cov
        self.put_error = lambda maxent, err: maxent.set_cov(err)

    def get_tau(self):
        d = self.maxent_diagonal.get_data_variable()
        o = self.maxent_offdiagonal.get_data_variable()
        if np.all(d == o):
            return d
        raise Exception('tau not uniquely defined. Use self.maxent_diagonal.tau '
                        'or self.maxent_offdiagonal.tau!')

    def set_tau(self, tau, **kwargs):
        self.maxent_diagonal.set_tau(tau, **kwargs)
        self.maxent_offdiagonal.set_tau(tau, **kwargs)

    tau = property(get_tau, set_tau)

    @property
    def shape(self):
        try:
            return self.determine_shape(self.G_mat)
        except Exception as e:
            raise Exception('Cannot determine shape. ({})'.format(e))


class DiagonalMaxEnt(ElementwiseMaxEnt):
    """diagonal elements only (reference elementwise_maxent.py:549-559)."""

    def run(self):
        self.run_diagonal()
        return self.maxent_result

    def run_offdiagonal(self):
        raise TypeError('DiagonalMaxEnt cannot run for off-diagonals.')


class PoormanMaxEnt(ElementwiseMaxEnt):
    r"""off-diagonals with :math:`D_{ij} = \sqrt{A_{ii}A_{jj}} + \epsilon`
    from the analyzed diagonals (reference elementwise_maxent.py:562-653)."""

    def __init__(self, analyzer_offdiag_D='LineFitAnalyzer',
                 D_add_constant=1.e-6, *args, **kwargs):
        super(PoormanMaxEnt, self).__init__(*args, **kwargs)
        self.analyzer_offdiag_D = analyzer_offdiag_D
        self.D_add_constant = D_add_constant

    def run_offdiagonal(self):
        self.prepare_maxent_result(overwrite=False)
        self.maxent_offdiagonal.logtaker.message(
            VerbosityFlags.ElementInfo,
            'Calculating off-diagonal elements using default model from '
            'diagonal solution')
        ar = self.maxent_result.analyzer_results
        jobs = self._offdiag_jobs()
        models = []
        for (i, j), re in jobs:
            if self.use_complex:
                A1 = ar[i][i][0][self.analyzer_offdiag_D]['A_out']
                A2 = ar[j][j][0][self.analyzer_offdiag_D]['A_out']
            else:
                A1 = ar[i][i][self.analyzer_offdiag_D]['A_out']
                A2 = ar[j][j][self.analyzer_offdiag_D]['A_out']
            models.append(DataDefaultModel(
                np.sqrt(A1 * A2) + self.D_add_constant, self.omega))
        return self._run_batch(self.maxent_offdiagonal, jobs,
                               per_job_D=models)
