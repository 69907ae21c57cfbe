"""Result container with the reference's field layout.

``MaxEntResult`` exposes the same fields, shapes and NaN conventions as the
reference's (reference python/maxent_result.py:157-188, 688-1081):

scalar run          alpha (X,)  v (X,n_s)  A,H (X,n_omega)  chi2,S,Q,probability
                    (X,)  G,G_orig,data_variable (n_tau,)  G_rec (X,n_tau)
element-wise run    the same with a prefix (M,N) or (M,N,2); NaN where an
                    element was not calculated; H and A of a missing (i,j)
                    are mirrored from (j,i) when ``use_hermiticity`` is set
                    (imaginary part negated) -- maxent_result.py:733-747;
``A_out``           from the default analyzer, zeros for ``zero_elements``,
                    hermitian mirror -- maxent_result.py:314-366.

The reference stores one cost-function object per (element, alpha) and
reduces lazily; here the device hands back whole arrays per element, which are
stored as they are (the reference's own on-disk form ``MaxEntResultData`` is
exactly that).  ``v`` holds the correct per-alpha vectors (the reference's
``v`` has every row equal to the last alpha's).
"""

import copy
from collections import OrderedDict
from datetime import datetime, timedelta
from itertools import product

import numpy as np

from .alpha_meshes import DataAlphaMesh
from .analyzers import Deferred, Picks
from .omega_meshes import DataOmegaMesh

_ALL_FIELDS = ['alpha', 'v', 'chi2', 'S', 'A', 'Q', 'omega', 'probability',
               'analyzer_results', 'run_times', 'run_time_total',
               'matrix_structure', 'effective_matrix_structure',
               'element_wise', 'complex_elements', 'use_hermiticity', 'G',
               'data_variable', 'G_rec', 'H', 'default_analyzer_name',
               'zero_elements', 'G_orig']


def _nested(shape, fill):
    if not shape:
        return fill()
    return [_nested(shape[1:], fill) for _ in range(shape[0])]


class MaxEntResultData(object):
    """plain-array form of a result; picklable (reference
    maxent_result.py:157-685)."""

    def __init__(self, matrix_structure=None, element_wise=True,
                 complex_elements=False, use_hermiticity=True):
        self._all_fields = list(_ALL_FIELDS)
        self._matrix_structure = None if matrix_structure is None \
            else tuple(matrix_structure)
        self._complex_elements = complex_elements
        self._element_wise = element_wise
        self._use_hermiticity = use_hermiticity
        self._default_analyzer_name = None
        self._zero_elements = []
        self._saved = dict()

    def __getattr__(self, name):
        if name == '_saved':
            raise AttributeError(name)
        saved = self.__dict__.get('_saved', {})
        if name in saved:
            return saved[name]
        raise AttributeError("'{}' object has no attribute '{}'".format(
            type(self).__name__, name))

    # ---- element access ---------------------------------------------
    def _get_element(self, array, matrix_element):
        if self.matrix_structure is None:
            assert matrix_element is None, \
                'Cannot give matrix_element when matrix_structure is None'
            return array
        if not self.element_wise:
            assert matrix_element is None, \
                'Cannot give matrix_element when element_wise is False'
            return array
        assert matrix_element is not None, 'matrix_element must be given'
        ret = array
        for i in matrix_element:
            ret = ret[i]
        return ret

    def element_array(self, name, matrix_element=None):
        """field ``name`` of one element (all of it for a scalar result)."""
        return self._get_element(getattr(self, name), matrix_element)

    # ---- field bookkeeping (h5 / pickle subset) -----------------------
    def include_only(self, fields):
        self._all_fields = []
        self.include(fields)

    def include(self, fields):
        for field in fields:
            if field not in _ALL_FIELDS:
                raise AttributeError('Unknown field: {}'.format(field))
            if field not in self._all_fields:
                self._all_fields.append(field)

    def exclude(self, fields):
        for field in fields:
            if field not in _ALL_FIELDS:
                raise AttributeError('Unknown field: {}'.format(field))
            if field in self._all_fields:
                self._all_fields.remove(field)

    # ---- analyzers ----------------------------------------------------
    def get_default_analyzer(self, analyzer=None):
        if analyzer is None:
            analyzer = self.default_analyzer_name
        if analyzer is None:
            analyzer = 'LineFitAnalyzer'
        if self.matrix_structure is None or not self.element_wise:
            return self.analyzer_results[analyzer]
        ret = np.empty(self.effective_matrix_structure, dtype=object)
        for elem in product(*map(range, self.effective_matrix_structure)):
            try:
                ret[elem] = self._get_element(self.analyzer_results,
                                              elem)[analyzer]
            except KeyError:
                ret[elem] = None
        return ret

    default_analyzer = property(get_default_analyzer)

    def get_A_out(self, analyzer=None):
        da = self.get_default_analyzer(analyzer)
        if self.matrix_structure is None or not self.element_wise:
            return da['A_out']
        n_omega = len(self.omega)
        ems = self.effective_matrix_structure
        A_out = np.full(tuple(ems) + (n_omega,), np.nan)
        for elem in self.zero_elements:
            A_out[elem] = 0.0
        for elem in product(*map(range, ems)):
            src, sign = elem, 1.0
            if da[elem] is None and self.use_hermiticity:
                t = list(elem)
                t[0], t[1] = t[1], t[0]
                if self.complex_elements and t[-1] == 1:
                    sign = -1.0
                src = tuple(t)
            if da[src] is not None and not isinstance(da[src], str):
                A_out[elem] = sign * da[src]['A_out']
        if self.complex_elements:
            return A_out[..., 0, :] + 1.0j * A_out[..., 1, :]
        return A_out

    A_out = property(get_A_out)

    # ---- persistence ----------------------------------------------------
    def __reduce_to_dict__(self):
        ret = dict(all_fields=list(self._all_fields))
        for key in self._all_fields:
            val = getattr(self, key)
            ret[key] = 'None' if val is None else val

        def conv(t):
            if isinstance(t, timedelta):
                return dict(days=t.days, seconds=t.seconds,
                            microseconds=t.microseconds)
            if isinstance(t, float):
                return t
            if isinstance(t, np.ndarray):
                return [conv(x) for x in t.tolist()] if t.ndim else conv(t.item())
            return [conv(x) for x in t]
        if 'run_times' in ret:
            ret['run_times'] = conv(ret['run_times'])
        if 'run_time_total' in ret:
            ret['run_time_total'] = conv(ret['run_time_total'])
        return ret

    @classmethod
    def __factory_from_dict__(cls, name, D):
        self = cls()
        D = dict(D)

        def unconv(t):
            if isinstance(t, dict):
                return timedelta(**t)
            if isinstance(t, float):
                return t
            return [unconv(x) for x in t]
        if 'run_times' in D:
            D['run_times'] = unconv(D['run_times'])
        if 'run_time_total' in D:
            D['run_time_total'] = unconv(D['run_time_total'])
        if 'omega' in D and not isinstance(D['omega'], str):
            D['omega'] = DataOmegaMesh(np.asarray(D['omega']))
        if 'alpha' in D and not isinstance(D['alpha'], str):
            D['alpha'] = DataAlphaMesh(np.asarray(D['alpha']))
        if 'all_fields' in D:
            self._all_fields = D.pop('all_fields')
        for key, val in D.items():
            self._saved[key] = None if (isinstance(val, str) and val == 'None') else val
        for key in ('matrix_structure', 'element_wise', 'complex_elements',
                    'use_hermiticity', 'default_analyzer_name',
                    'zero_elements'):
            if key in self._saved:
                setattr(self, '_' + key, self._saved[key])
        return self

    # fields that live in ``_saved`` for a pure data object
    @property
    def matrix_structure(self):
        return self._matrix_structure

    @property
    def element_wise(self):
        return self._element_wise

    @property
    def complex_elements(self):
        return self._complex_elements

    @property
    def use_hermiticity(self):
        return self._use_hermiticity

    @property
    def default_analyzer_name(self):
        return self._default_analyzer_name

    @property
    def zero_elements(self):
        return self._zero_elements

    @property
    def effective_matrix_structure(self):
        if self._matrix_structure is None:
            return None
        if self._element_wise and self._complex_elements:
            return tuple(self._matrix_structure) + (2,)
        return tuple(self._matrix_structure)


_row_pool = None


def _by_rows(fn, src, dst, n_threads=4, min_bytes=32 << 20):
    """``fn(src[i0:i1], dst[i0:i1])`` over blocks of the first axis, on a few threads for arrays of tens of MB (one thread
    moves ~15 GB/s on the hosts of the GPU boxes; numpy releases the GIL inside its loops)"""
    global _row_pool
    n = src.shape[0] if src.ndim else 0
    if src.nbytes < min_bytes or n < 2:
        fn(src, dst)
        return
    if _row_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _row_pool = ThreadPoolExecutor(max_workers=n_threads)
    step = -(-n // n_threads)
    for f in [_row_pool.submit(fn, src[i:i + step], dst[i:i + step]) for i in range(0, n, step)]:
        f.result()


class ElementAnalysis(OrderedDict):
    """{analyzer name: AnalyzerResult} of one element.  An entry may still be a :class:`analyzers.Deferred` -- the
    device picked the alpha, nobody has looked at the result object yet --: it is built when it is first read;
    ``raw(name)`` reads without building (``MaxEntResult.A_out`` takes the A row straight from the batch)."""

    _n = None        # position of this element in the batches whose Picks stand for its entries

    def raw(self, name, default=None):
        val = OrderedDict.get(self, name, default)
        return Deferred(val, self._n) if isinstance(val, Picks) else val

    def __getitem__(self, name):
        val = OrderedDict.__getitem__(self, name)
        if isinstance(val, Picks):                 # (the whole batch's choice of one analyzer: this element's result, built now)
            val = val.build(self._n)
            OrderedDict.__setitem__(self, name, val)
        elif isinstance(val, Deferred):
            val = val.build()
            OrderedDict.__setitem__(self, name, val)
        return val

    def get(self, name, default=None):
        return self[name] if OrderedDict.__contains__(self, name) else default

    def settle(self):
        for name in list(OrderedDict.keys(self)):
            self[name]
        return self

    def values(self):
        return OrderedDict.values(self.settle())

    def items(self):
        return OrderedDict.items(self.settle())

    def copy(self):
        return OrderedDict(OrderedDict.items(self.settle()))

    def __eq__(self, other):
        return OrderedDict.__eq__(self.settle(), other)

    def __reduce__(self):
        return (OrderedDict, (list(OrderedDict.items(self.settle())),))

    def __repr__(self):
        return OrderedDict.__repr__(self.settle())


class MaxEntResult(MaxEntResultData):
    """live result that the solver fills element by element."""

    def __init__(self, matrix_structure=None, element_wise=True,
                 complex_elements=False, use_hermiticity=True):
        super(MaxEntResult, self).__init__(matrix_structure, element_wise,
                                           complex_elements, use_hermiticity)
        self._deferred = []                 # launches whose per-element records have not been built (add_deferred)
        self._records = OrderedDict()       # key (tuple or None) -> dict
        self._analysis = OrderedDict()      # key -> {name: AnalyzerResult}
        self._start = dict()
        self._end = dict()
        self._cache = dict()

    # ---- launches whose records are built when somebody looks -----------------
    # ``ElementwiseMaxEnt.run()`` on array input hands the result ONE object per launch (``elementwise_maxent.DeferredLaunch``)
    # instead of 256 records and 256 analyses: the fields a caller usually reads -- A_out, chi2, S, Q, alpha, omega, H, A,
    # n_iter, converged -- are served from the launch's arrays as they came off the device; anything per element (a record,
    # an analyzer result, G_rec, run_times, a pickle ...) first SETTLES the launch: it then builds its records and analyses
    # exactly as run() used to, and everything goes the general way.  _records / _analysis / _start / _end are therefore
    # properties that settle before they hand the tables out.
    def _table(name):                                           # noqa: N805
        store = '_' + name + '_store'

        def get(self):
            if self.__dict__.get('_deferred'):
                self._settle()
            return self.__dict__[store]

        def put(self, value):
            self.__dict__[store] = value
        return property(get, put)
    _records, _analysis, _start, _end = _table('records'), _table('analysis'), _table('start'), _table('end')
    del _table

    def add_deferred(self, launch):
        """``launch``: keys (in launch order), ``settle(result)``, and the arrays the fast accessors read"""
        if self.__dict__['_deferred'] or self.__dict__['_records_store']:
            self._settle()                  # (something is there already: everything the general way)
        self._deferred.append(launch)
        self._cache = dict()
        if self.__dict__['_records_store']:
            self._settle()

    def _settle(self):
        pending, self.__dict__['_deferred'] = self.__dict__['_deferred'], []
        for launch in pending:
            launch.settle(self)
        if pending:
            self._cache = dict()

    def _whole(self):
        """the one unsettled launch that covers every element of the matrix in the order of the matrix, or None"""
        d = self.__dict__.get('_deferred')
        if not d or len(d) != 1:
            return None
        launch = d[0]
        ok = launch.covers
        if ok is None:
            ok = launch.covers = bool(
                self._matrix_structure is not None and self.element_wise and not self.complex_elements and
                not self._zero_elements and launch.keys == list(product(*map(range, self._matrix_structure))))
        return launch if ok else None

    def _whole_field(self, name):
        launch = self._whole()
        if launch is None:
            return None
        val = self._cache.get(name)
        if val is None:
            val = launch.field(name, tuple(self._matrix_structure))
            if val is None:
                return None
            self._cache[name] = val
        return val

    # ---- filling -------------------------------------------------------
    def _key(self, matrix_element, complex_index):
        if self.matrix_structure is None or not self.element_wise:
            return None
        assert matrix_element is not None, 'matrix_element must be given'
        key = tuple(matrix_element)
        if self.complex_elements and complex_index is not None \
                and len(key) == 2:
            key = key + (complex_index,)
        return key

    def _keys(self, elements):
        """:meth:`_key` for many (matrix_element, complex_index) pairs"""
        if self.matrix_structure is None or not self.element_wise:
            return [None] * len(elements)
        if self.complex_elements:
            return [self._key(element, cidx) for (element, cidx) in elements]
        return [tuple(element) for (element, _) in elements]

    def add_element_results(self, record, matrix_element=None,
                            complex_index=None):
        """Store the arrays of one finished alpha scan.  ``record`` needs the
        keys alpha, v, H, A, chi2, S, Q, G, G_orig, data_variable, G_rec,
        omega and optionally probability, n_iter, converged, run_times."""
        key = self._key(matrix_element, complex_index)
        if self.matrix_structure is not None and self.element_wise:
            assert key is not None, 'matrix_element must be given'
        self._records[key] = record
        self._cache = dict()

    def add_batch_results(self, records, elements, t_start=None, t_end=None, keys=None):
        """:meth:`add_element_results` (+ :meth:`start_timing` / :meth:`end_timing` when the times are given) for the
        scans of one launch; ``elements``: (matrix_element, complex_index) pairs (``keys``: their keys, where the caller
        has made them already).  Returns their keys"""
        if keys is None:
            keys = [self._key(element, cidx) for (element, cidx) in elements]
        self._records.update(zip(keys, records))
        if t_start is not None:
            self._start.update(dict.fromkeys(keys, t_start))
        if t_end is not None:
            self._end.update(dict.fromkeys(keys, t_end))
        self._cache = dict()
        return keys

    def add_result(self, cost_function, log_probability=None, matrix_element=None, complex_index=None):
        """Append the result of ONE alpha (reference maxent_result.py:751-791): ``cost_function`` is a cost
        function evaluated at its optimum, ``Q(v)`` -- chi2, S, Q, H, A are read off it (device evaluation,
        cost_functions.py) and become one more row of the element's arrays.  A result that a launch filled
        scan by scan (``add_element_results``) can be extended the same way; its H then comes to the host."""
        if self.matrix_structure is None or not self.element_wise:
            assert matrix_element is None, 'this result has no matrix structure: matrix_element must not be given'
        key = self._key(matrix_element, complex_index)
        q = cost_function
        H = np.array(q.H_of_v.f(), dtype=float)
        A = np.asarray(q.A_of_H.f(H))
        K = q.K
        now = datetime.now()
        last = self.__dict__.get('_last_add', now)
        self._last_add = now
        row = dict(alpha=float(q._alpha), v=np.array(q._x, dtype=float), H=H, A=A,
                   chi2=float(q.chi2.f()), S=float(q.S.f()), Q=float(q.f()),
                   G_rec=np.dot(K.K_delta, A),
                   probability=np.nan if log_probability is None else log_probability,
                   n_iter=0, converged=True, n_evals=0, run_times=now - last)
        rec = self._records.get(key)
        if rec is None:
            rec = dict((k, np.array([val])) for k, val in row.items())
            rec.update(G=np.array(q.G), G_orig=np.array(q.G_orig), data_variable=np.array(K.data_variable), omega=q.omega)
        else:
            rec = dict(rec)
            for k, val in row.items():
                if k in rec and rec[k] is not None:
                    rec[k] = np.concatenate([np.asarray(rec[k]), np.array([val])])
        self._records[key] = rec
        self._cache = dict()

    def _get_empty(self, fill_with=list):
        """a nested list in the shape of the matrix structure, every entry a fresh ``fill_with()``
        (reference maxent_result.py:268-290)"""
        shape = self.effective_matrix_structure
        if shape is None:
            return fill_with()

        def level(dims):
            return [fill_with() if len(dims) == 1 else level(dims[1:]) for _ in range(dims[0])]
        return level(list(shape))

    def start_timing(self, matrix_element=None, complex_index=None, time=None):
        self._start[self._key(matrix_element, complex_index)] = \
            time or datetime.now()

    def end_timing(self, matrix_element=None, complex_index=None, time=None):
        key = self._key(matrix_element, complex_index)
        self._end[key] = time or datetime.now()
        return self._end[key] - self._start.get(key, self._end[key])

    def analyze(self, analyzers, matrix_element=None, complex_index=None):
        """run the analyzers on one element; a ``ValueError`` of an analyzer
        is stored as its message (reference maxent_result.py:793-822)."""
        key = self._key(matrix_element, complex_index)
        out = OrderedDict()
        for analyzer in analyzers:
            try:
                res = analyzer.analyze(self, key)
                res.maxent_result = self
                out[res['name']] = res
            except ValueError as e:
                out[analyzer.name] = str(e)
        self._analysis[key] = out
        self._cache.pop('analyzer_results', None)

    # ---- single rows of H / A without moving the rest off the device -----------
    class _RowPlaceholder(object):
        """stands for ``record[name][idx]`` until :meth:`analyze_batch` has fetched all rows at once"""

        def __init__(self, lazy, idx):
            self.lazy, self.idx, self.value = lazy, int(idx), None

    def element_row(self, name, matrix_element, idx):
        """``element_array(name, matrix_element)[idx]``; when the element's H still lives on the device
        only that row is fetched (inside :meth:`analyze_batch`: later, together with everybody else's)"""
        key = None if (self.matrix_structure is None or not self.element_wise) else tuple(matrix_element)
        rec = self._records.get(key)
        val = None if rec is None else rec.get(name)
        if val is None or getattr(val, 'on_host', True) or not hasattr(val, 'row_request'):
            return np.asarray(self.element_array(name, matrix_element))[idx]
        deferred = self.__dict__.get('_deferred_rows')
        if deferred is None:
            return val[int(idx)]
        ph = MaxEntResult._RowPlaceholder(val, idx)
        deferred.append(ph)
        return ph

    def get_A_out(self, analyzer=None):
        if self.matrix_structure is None or not self.element_wise:
            return self.get_default_analyzer(analyzer)['A_out']
        if analyzer is None:
            analyzer = self.default_analyzer_name
        if analyzer is None:
            analyzer = 'LineFitAnalyzer'
        launch = self._whole()
        if launch is not None:
            fast = launch.A_out(analyzer, tuple(self._matrix_structure))       # (the device's rows of that analyzer, one division)
            if fast is not None:
                return fast

        def chosen(elem):
            """A_out of the analyzer for this element, None where it has none (the rows of a batch the device picked
            are read as they are: no result object is built for this)"""
            out = self._analysis.get(tuple(elem))
            if out is None:
                return None
            val = out.raw(analyzer) if hasattr(out, 'raw') else out.get(analyzer)
            if isinstance(val, Deferred):
                return val.A_out()
            return None if (val is None or isinstance(val, str)) else val['A_out']
        n_omega = len(self.omega)
        ems = self.effective_matrix_structure
        A_out = np.full(tuple(ems) + (n_omega,), np.nan)
        for elem in self.zero_elements:
            A_out[elem] = 0.0
        for elem in product(*map(range, ems)):
            sign, row = 1.0, chosen(elem)
            if row is None and self.use_hermiticity and tuple(elem) not in self._analysis:
                t = list(elem)
                t[0], t[1] = t[1], t[0]
                if self.complex_elements and t[-1] == 1:
                    sign = -1.0
                row = chosen(t)
            if row is not None:
                A_out[elem] = sign * row
        if self.complex_elements:
            return A_out[..., 0, :] + 1.0j * A_out[..., 1, :]
        return A_out

    A_out = property(get_A_out)

    def analyze_batch(self, analyzers, keys, picks_for_one=False):
        """:meth:`analyze` for many elements (``keys``: tuples as :meth:`_key` makes them); the rows of A
        the analyzers pick are fetched from the device in ONE go at the end"""
        self._deferred_rows = []
        self.__dict__.pop('_picks_scan', None)
        try:
            many = {}
            for analyzer in analyzers:
                if hasattr(analyzer, 'pick_many') and (len(keys) > 1 or picks_for_one):
                    picks = analyzer.pick_many(self, keys)        # the device chose: result objects when somebody looks
                    if picks is not None:
                        many[analyzer.name] = picks               # (ONE object for the batch; an element finds its own through its position)
                        continue
                if hasattr(analyzer, 'analyze_many') and len(keys) > 1:
                    many[analyzer.name] = analyzer.analyze_many(self, keys)
            for n, key in enumerate(keys):
                out = ElementAnalysis()
                out._n = n
                for analyzer in analyzers:
                    try:
                        got = many.get(analyzer.name)
                        if isinstance(got, Picks):
                            OrderedDict.__setitem__(out, analyzer.name, got)
                            continue
                        res = got[n] if got is not None else analyzer.analyze(self, key)
                        if isinstance(res, Deferred):
                            OrderedDict.__setitem__(out, analyzer.name, res)
                            continue
                        if isinstance(res, str):
                            out[analyzer.name] = res
                        else:
                            res.maxent_result = self
                            out[res['name']] = res
                    except ValueError as e:
                        out[analyzer.name] = str(e)
                self._analysis[key] = out
            self._cache.pop('analyzer_results', None)
            pending = self._deferred_rows
        finally:
            self._deferred_rows = None
            self.__dict__.pop('_picks_scan', None)
        if pending:
            owner = None
            for ph in pending:
                H = getattr(ph.lazy, '_H', ph.lazy)
                owner = H._owner
                break
            rows = owner.rows([ph.lazy.row_request(ph.idx) for ph in pending])
            for ph, row in zip(pending, rows):
                ph.value = ph.lazy.from_H_row(row) if hasattr(ph.lazy, 'from_H_row') else row
            for key in keys:
                for res in OrderedDict.values(self._analysis.get(key, {})):
                    if isinstance(res, dict):
                        for k, v in list(res.items()):
                            if isinstance(v, MaxEntResult._RowPlaceholder):
                                res[k] = v.value

    def element_array(self, name, matrix_element=None):
        """field ``name`` of one element straight from its record (no assembly of the
        array over all elements); mirrored / missing elements go the general way."""
        if self.matrix_structure is None or not self.element_wise:
            return np.asarray(self._records[None][name])
        rec = self._records.get(tuple(matrix_element))
        if rec is None or name not in rec:
            return self._get_element(getattr(self, name), matrix_element)
        return np.asarray(rec[name])

    # ---- assembling ------------------------------------------------------
    def _reference_record(self):
        if not self._records:
            raise AttributeError('no results have been added yet')
        best = self._cache.get(('reference record',))       # (the cache is dropped whenever a record is added)
        if best is None:
            for rec in self._records.values():
                if best is None or len(rec['alpha']) > len(best['alpha']):
                    best = rec
            self._cache[('reference record',)] = best
        return best

    @property
    def _n_alphas(self):
        if self._matrix_structure is None:
            rec = self._records.get(None)
            return 0 if rec is None else len(rec['alpha'])
        ret = np.zeros(self.effective_matrix_structure, dtype=int)
        for key, rec in self._records.items():
            ret[key] = len(rec['alpha'])
        return ret

    def _assemble(self, name, mirror=False, per_alpha=True):
        if name in self._cache:
            return self._cache[name]
        if self.matrix_structure is None or not self.element_wise:
            arr = np.asarray(self._records[None][name])
        else:
            ref = self._reference_record()
            X = len(ref['alpha'])
            full = getattr(ref[name], 'shape', None)            # (a lazy array knows its shape without being formed)
            if full is None:
                full = np.asarray(ref[name]).shape
            tail = tuple(full)[1:] if per_alpha else tuple(full)
            ems = tuple(self.effective_matrix_structure)
            shape = ems + ((X,) if per_alpha else ()) + tuple(tail)
            fast = self._assemble_whole(name, ems, shape) if per_alpha else None
            if fast is not None:
                self._cache[name] = fast
                return fast
            arr = np.full(shape, np.nan)
            for key, rec in self._records.items():
                val = np.asarray(rec[name], dtype=float)
                if per_alpha:
                    arr[key][:len(val)] = val
                else:
                    arr[key] = val
            if mirror and self.use_hermiticity:
                for elem in product(*map(range, self._matrix_structure)):
                    if elem == elem[::-1]:
                        continue
                    if np.all(np.isnan(arr[elem])):
                        arr[elem] = arr[elem[::-1]]
                        if self.complex_elements:
                            arr[elem + (1,)] = -arr[elem + (1,)]
        self._cache[name] = arr
        return arr

    def _assemble_whole(self, name, ems, shape):
        """the array of ``name`` over all elements when EVERY element of the matrix has a full-length record (nothing to
        fill with NaN, nothing to mirror), or None.  The records of one launch are consecutive rows of ONE fetched array
        in the order of the matrix (ElementwiseMaxEnt launches them that way): the result then is a view of that array --
        no second copy of 100 MB; otherwise one copy into an uninitialised array.  A = H / delta of all elements is one
        division."""
        keys = list(product(*map(range, ems)))
        recs = [self._records.get(k) for k in keys]
        if not recs or any(r is None or name not in r for r in recs):
            return None
        if name == 'A':
            maps = [getattr(r['A'], '_map', None) for r in recs]
            first = maps[0]

            def same_map(m):
                # (the diagonal and the off-diagonal worker have maps of their own: equal when both divide by the same delta)
                if m is first:
                    return True
                if m is None or type(m) is not type(first) or m.matrix() is not None:
                    return False
                a, b = getattr(getattr(m, '_omega', None), 'delta', None), getattr(getattr(first, '_omega', None), 'delta', None)
                return a is not None and b is not None and np.shape(a) == np.shape(b) and np.array_equal(a, b)
            distinct = {id(m): m for m in maps}
            if first is not None and first.matrix() is None and all(same_map(m) for m in distinct.values()) and \
                    not any(getattr(r['A'], '_val', None) is not None for r in recs):
                H = self._assemble('H', mirror=True)
                if H.shape == shape:
                    delta = getattr(getattr(first, '_omega', None), 'delta', None)
                    if type(first).__name__ == 'IdentityA_of_H' and delta is not None:
                        # (into a block of the library's page-locked pool: a fresh pageable array of 100 MB costs more in
                        #  page faults than the division itself)
                        from . import device
                        out = device.pinned_empty(shape)
                        _by_rows(lambda a, b: np.divide(a, delta, out=b), H, out)
                        return out
                    return np.asarray(first.f(H))
        vals = [np.asarray(r[name]) for r in recs]
        v0 = vals[0]
        if v0.dtype != np.float64 or any(v.shape != v0.shape or v.dtype != v0.dtype for v in vals) or \
                tuple(ems) + v0.shape != tuple(shape):
            return None
        nb = v0.nbytes
        if nb and v0.flags.c_contiguous:
            p0 = v0.__array_interface__['data'][0]
            if all(v.flags.c_contiguous and v.__array_interface__['data'][0] == p0 + i * nb for i, v in enumerate(vals)):
                # (a VIEW of what came off the device, shared with the records and with every field derived later: read-only, so
                #  that an in-place edit of result.H fails loudly instead of silently changing them -- ``np.array(result.H)``
                #  gives a private copy, as every field of the reference is)
                whole = np.lib.stride_tricks.as_strided(v0, shape=(len(vals),) + v0.shape, strides=(nb,) + v0.strides, writeable=False)
                return whole.reshape(shape)
        arr = np.empty(shape)
        flat = arr.reshape((len(vals),) + v0.shape)
        for i, v in enumerate(vals):
            flat[i] = v
        return arr

    @property
    def alpha(self):
        launch = self._whole()
        if launch is not None:
            return np.asarray(launch.alpha)
        return np.asarray(self._reference_record()['alpha'])

    @property
    def omega(self):
        launch = self._whole()
        if launch is not None:
            return launch.omega
        return self._reference_record()['omega']

    @property
    def v(self):
        return self._assemble('v')

    @property
    def chi2(self):
        fast = self._whole_field('chi2')
        return fast if fast is not None else self._assemble('chi2')

    @property
    def S(self):
        fast = self._whole_field('S')
        return fast if fast is not None else self._assemble('S')

    @property
    def Q(self):
        fast = self._whole_field('Q')
        return fast if fast is not None else self._assemble('Q')

    @property
    def H(self):
        fast = self._whole_field('H')
        return fast if fast is not None else self._assemble('H', mirror=True)

    @property
    def A(self):
        fast = self._whole_field('A')
        return fast if fast is not None else self._assemble('A', mirror=True)

    @property
    def G_rec(self):
        return self._assemble('G_rec')

    @property
    def G(self):
        return self._assemble('G', per_alpha=False)

    @property
    def G_orig(self):
        return self._assemble('G_orig', per_alpha=False)

    @property
    def data_variable(self):
        return self._assemble('data_variable', per_alpha=False)

    @property
    def probability(self):
        return self._assemble('probability')

    @property
    def n_iter(self):
        """Newton iterations per (element, alpha) (not in the reference)."""
        fast = self._whole_field('n_iter')
        return fast if fast is not None else self._assemble('n_iter')

    @property
    def converged(self):
        fast = self._whole_field('converged')
        return fast if fast is not None else self._assemble('converged')

    def _nested_from(self, table, fill):
        if self.matrix_structure is None or not self.element_wise:
            return table.get(None, fill())
        ems = self.effective_matrix_structure
        out = _nested(ems, fill)
        for key, val in table.items():
            node = out
            for i in key[:-1]:
                node = node[i]
            node[key[-1]] = val
        return out

    @property
    def analyzer_results(self):
        if 'analyzer_results' not in self._cache:
            self._cache['analyzer_results'] = self._nested_from(self._analysis, dict)
        return self._cache['analyzer_results']

    def _object_array(self, table, per_alpha):
        """object array in the shape of the matrix structure (+ the alpha axis), NaN where an element has
        nothing (reference maxent_result.py:720-731 with dtype=object)"""
        if self.matrix_structure is None or not self.element_wise:
            val = table.get(None, [] if per_alpha else np.nan)
            if not per_alpha:
                return val
            out = np.empty(len(val), dtype=object)
            out[:] = list(val)
            return out
        ems = tuple(self.effective_matrix_structure)
        n = max([len(v) for v in table.values()] + [0]) if per_alpha else 0
        out = np.empty(ems + ((n,) if per_alpha else ()), dtype=object)
        out[...] = np.nan
        for key, val in table.items():
            if per_alpha:
                for i, t in enumerate(val):
                    out[key + (i,)] = t
            else:
                out[key] = val
        return out

    @property
    def run_times(self):
        return self._object_array({k: list(r.get('run_times', [])) for k, r in self._records.items()}, True)

    @property
    def run_time_total(self):
        tot = {k: self._end[k] - self._start[k] for k in self._end if k in self._start}
        if not tot and self._records:
            # filled alpha by alpha (add_result): the sum of the single times
            tot = {k: sum(list(r.get('run_times', [])), timedelta(0)) for k, r in self._records.items()}
        return self._object_array(tot, False)

    @property
    def data(self):
        """detached :class:`MaxEntResultData` with plain arrays (for pickle /
        h5; reference maxent_result.py:1069-1081)."""
        d = self.__reduce_to_dict__()
        d = copy.deepcopy({k: v for k, v in d.items() if k != 'analyzer_results'})
        ar = self.analyzer_results

        def strip(x):
            if isinstance(x, dict):
                if 'name' in x or not x:
                    y = type(x)(x)
                    return y
                return {k: strip(v) for k, v in x.items()}
            if isinstance(x, list):
                return [strip(y) for y in x]
            return x
        d['analyzer_results'] = strip(ar)
        return MaxEntResultData.__factory_from_dict__('MaxEntResultData', d)
