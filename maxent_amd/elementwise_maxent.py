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

    def _prepare_batch(self, worker, jobs, per_job_D=None):
        """the specs of the elements of ``jobs`` that are to be solved (the others go to the result's zero
        elements); leaves ``worker`` loaded with the last element, as the reference does"""
        self.prepare_maxent_result(overwrite=False)
        res = self.maxent_result
        loop = worker.maxent_loop
        specs, live = [], []
        self._share_decomposition()
        direct = self._direct_input(worker) and per_job_D is None
        template = None
        below = err_same = g_rows = first_spec = None
        if direct and len(jobs) > 1:
            # (all elements at once: the data vectors as the rows of ONE array -- real part, or imaginary part of an off-diagonal
            #  element's second scan --, which of them are below the threshold; one error array when it is the same for all)
            Gm = self.G_mat[1]
            ii = np.fromiter((e[0] for e, _ in jobs), dtype=np.intp, count=len(jobs))
            jj = np.fromiter((e[1] for e, _ in jobs), dtype=np.intp, count=len(jobs))
            Gsel = Gm[ii, jj]
            real_part = np.fromiter((bool(re) for _, re in jobs), dtype=bool, count=len(jobs)) | (ii == jj)
            if np.iscomplexobj(Gsel):
                g_rows = np.where(real_part[:, None], Gsel.real, Gsel.imag).astype(float, copy=False)
            else:
                g_rows = np.array(Gsel, dtype=float)
                g_rows[~real_part] = 0.0             # (the imaginary part of real data)
            with np.errstate(all='ignore'):
                below = (np.max(np.abs(g_rows), axis=-1) < loop.G_threshold).tolist()
            g_rows = list(g_rows)                    # (row views, made in one go)
            e0 = self.get_error(tuple(jobs[0][0]))
            if isinstance(self.error, float) or len(np.shape(self.error)) == self.error_dimension:
                err_same = np.asarray(e0, dtype=float) * np.ones(np.shape(Gm)[-1])
        for n, (element, re) in enumerate(jobs):
            cidx = 0 if re else 1
            if direct and n > 0:
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
            for n in range(1, len(jobs)):
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
                    specs.append(template)
                    specs.extend([{**template, 'G': g_rows[n], 'G_orig': g_rows[n]} for n in rest[1:]])
                else:
                    specs.append(template)
                    specs.extend([loop.spec_like(template, g_rows[n], self.get_error(tuple(jobs[n][0]))) for n in rest[1:]])
                live.extend([(jobs[n][0], 0 if jobs[n][1] else 1) for n in rest])
                if rest[-1] == len(jobs) - 1:
                    self._load_element(worker, jobs[-1][0], jobs[-1][1])
        # (the keys of the result's records, made once: they were made three times per element -- 0.2 ms of a 16 x 16 run)
        return dict(worker=worker, specs=specs, live=live, keys=res._keys(live))

    def _solve_batches(self, batches):
        """one launch for all batches whose workers share the decomposition of the kernel, the minimiser settings
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
        for g in groups:
            loop = g[0]['worker'].maxent_loop
            # the scans of the launch in the order of the result's matrix (row major): what comes off the device in one
            # copy then IS the (M, N, n_alpha, n_omega) array of the result -- MaxEntResult._assemble takes it as a view
            where = [(key, n, k) for n, b in enumerate(g) for k, key in enumerate(b['keys'])]
            try:
                where.sort(key=lambda t: t[0])
            except TypeError:
                pass
            specs = [g[n]['specs'][k] for (_, n, k) in where]
            t0 = datetime.now()
            for b in g:
                res._start.update(dict.fromkeys(b['keys'], t0))
            for b in g:
                b['sols'] = [None] * len(b['specs'])

            def hand_out(sols, g=g, where=where):
                for sol, (_, n, k) in zip(sols, where):
                    g[n]['sols'][k] = sol

            def records_while_the_kernel_runs(sols, g=g):
                # (the result dicts hold their arrays already, the device fills them behind this: the records of the scans --
                #  0.3 ms of dictionaries for 256 elements -- cost nothing next to a kernel of 0.8 ms)
                hand_out(sols)
                for b in g:
                    b['records'] = b['worker'].maxent_loop.make_records(b['specs'], b['sols'])
            plain = all(b['worker'].maxent_loop.probability is None for b in g)
            sols, info = solve_elements(loop.K, specs, loop.minimizer,
                                        device_id=loop.device_id, device_ids=self.device_ids,
                                        want_logdet=loop.probability is not None,
                                        chi2_factor=loop.cost_function.chi2_factor,
                                        select=select_params(loop.analyzers),
                                        while_waiting=records_while_the_kernel_runs if plain else None)
            t1 = datetime.now()
            self.last_launches.append(info)
            for b in g:
                b.update(info=info, t0=t0, t1=t1, per_alpha=(t1 - t0) / max(1, len(specs) * len(specs[0]['alpha'])))
            if not all('records' in b for b in g):
                hand_out(sols)

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
        worker, specs, live = batch['worker'], batch['specs'], batch['live']
        if not specs:
            return res
        loop = worker.maxent_loop
        if res._default_analyzer_name is None and loop.analyzers:
            res._default_analyzer_name = loop.analyzers[0].name
        sols, info, t1, per_alpha = batch['sols'], batch['info'], batch['t1'], batch['per_alpha']
        talk = bool(worker.logtaker.verbose & (VerbosityFlags.ElementInfo | VerbosityFlags.AlphaLoop))
        if talk:
            for sol, (element, cidx) in zip(sols, live):
                worker.logtaker.message(
                    VerbosityFlags.ElementInfo,
                    'Element {} {}{}'.format(element[0], element[1],
                                             '' if cidx == 0 else ' (imaginary part)'))
                loop.log_alpha_lines(sol)
        records = batch.pop('records', None)
        if records is None:
            records = loop.make_records(specs, sols)
        times = {}
        for rec in records:
            X = len(rec['alpha'])
            if X not in times:
                times[X] = (per_alpha,) * X        # (one immutable tuple for the records of a launch; MaxEntResult.run_times hands out lists)
            rec['run_times'] = times[X]
        keys = res.add_batch_results(records, live, t_end=t1, keys=batch.get('keys'))
        if not talk and sols:
            # (one reduction over the rows of the launch's count array, not one numpy call per element: 1.2 ms for 256 elements)
            loop.note_minimizer_state(sols[-1], int(np.asarray([x['n_iter'] for x in sols]).sum()))
        # analyzers after every record of the batch is in (adding a record drops the assembled-array cache
        # of the result); the rows of A they select come off the device in one go
        res.analyze_batch(loop.analyzers, keys)
        worker.logtaker.message(
            VerbosityFlags.Timing,
            '{} alpha scans x {} alpha in one launch: kernel {:.3f} ms',
            len(specs), len(specs[0]['alpha']), info['kernel_ms'])
        return res

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

    def _diag_jobs(self):
        return [((i, i), True) for i in range(self.shape[0])]

    def _offdiag_jobs(self):
        jobs = []
        for i in range(self.shape[0]):
            for j in range(self.shape[1]):
                if i == j or (self.use_hermiticity and i > j):
                    continue
                for re in ([True, False] if self.use_complex else [True]):
                    jobs.append(((i, j), re))
        return jobs

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
        cls = type(self)
        if (cls.run_diagonal is not ElementwiseMaxEnt.run_diagonal or
                cls.run_offdiagonal is not ElementwiseMaxEnt.run_offdiagonal):
            self.run_diagonal()              # (a subclass with phases of its own: PoormanMaxEnt needs the
            self.run_offdiagonal()           #  diagonal results before the off-diagonal elements start)
            return self.maxent_result
        # both phases in ONE launch where the two workers share the kernel's decomposition
        self.maxent_diagonal.logtaker.message(VerbosityFlags.ElementInfo, 'Calculating diagonal elements.')
        diag = self._prepare_batch(self.maxent_diagonal, self._diag_jobs())
        self.maxent_offdiagonal.logtaker.message(VerbosityFlags.ElementInfo, 'Calculating off-diagonal elements.')
        off = self._prepare_batch(self.maxent_offdiagonal, self._offdiag_jobs())
        self._solve_batches([diag, off])
        res = self._finish_batch(diag)
        self._mark_imaginary_diagonal(res)
        return self._finish_batch(off)

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
