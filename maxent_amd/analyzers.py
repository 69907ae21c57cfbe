"""alpha selection: pick one A(omega) from the alpha-dependent family.

Same classes, result keys and selection rules as the reference's
``analyzers`` package (reference python/analyzers/): ``LineFitAnalyzer``
(linefit_analyzer.py:28-87,151-183), ``Chi2CurvatureAnalyzer``
(chi2_curvature_analyzer.py:25-49,101-131), ``EntropyAnalyzer``
(entropy_analyzer.py:72-103), ``BryanAnalyzer`` (bryan_analyzer.py:106-154),
``ClassicAnalyzer`` (classic_analyzer.py:50-82).  They run on the host on the
arrays the device returned; the piecewise line fit is done with running sums
(O(n_alpha)) instead of 2 n_alpha calls of ``np.polyfit``, because it runs
once per matrix element and once between the two phases of PoormanMaxEnt.
"""

import weakref

import numpy as np


class AnalyzerResult(dict):
    """dict with the keys ``A_out``, ``name``, ``info`` and, where it applies,
    ``alpha_index`` (reference analyzers/analyzer.py:25-46)."""

    # ``maxent_result`` (reference analyzers/analyzer.py:52-64: the result the analysis belongs to) is held
    # weakly: the result owns its analyses, and a strong reference back would keep a dropped result -- with
    # its claim on 100 MB of H on the device -- alive until the cycle collector gets to it
    @property
    def maxent_result(self):
        ref = self.__dict__.get('_result_ref')
        return None if ref is None else ref()

    @maxent_result.setter
    def maxent_result(self, value):
        self.__dict__['_result_ref'] = None if value is None else weakref.ref(value)

    def __getstate__(self):
        self._settle()
        return dict((k, v) for k, v in self.__dict__.items() if k not in ('_result_ref', '_lazy'))

    def __reduce_to_dict__(self):
        self._settle()
        return self

    # Entries that cost something and that hardly anybody reads (the fit parameters of the line fit, the curvature
    # and dS / dlog alpha curves when the device has already picked the alpha) are computed when first looked at:
    # ``later(key, thunk)``.  ``res[key]`` computes that one; anything that looks at the dict as a whole settles all.
    def later(self, key, thunk):
        self.__dict__.setdefault('_lazy', {})[key] = thunk

    def __missing__(self, key):
        lazy = self.__dict__.get('_lazy')
        if lazy and key in lazy:
            val = self[key] = lazy.pop(key)()
            return val
        raise KeyError(key)

    def _settle(self):
        lazy = self.__dict__.get('_lazy')
        while lazy:
            key, thunk = lazy.popitem()
            dict.__setitem__(self, key, thunk())

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.__dict__.get('_lazy', ())

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def _settled(name):
        def method(self, *a, **k):
            self._settle()
            return getattr(dict, name)(self, *a, **k)
        method.__name__ = name
        return method
    for _n in ('keys', 'values', 'items', '__iter__', '__len__', '__repr__', '__eq__', 'copy', 'pop', 'popitem', '__reduce_ex__'):
        locals()[_n] = _settled(_n)
    del _n, _settled

    @classmethod
    def __factory_from_dict__(cls, name, D):
        self = cls()
        self.update(D)
        return self


class Analyzer(object):
    def __init__(self, name=None, **kwargs):
        self.name = self.__class__.__name__ if name is None else name

    def analyze(self, maxent_result, matrix_element=None):
        raise NotImplementedError('Please use a subclass of Analyzer.')


def _linfit_sse(x, y):
    """least-squares line through (x, y): (slope, intercept, SSE)."""
    n = len(x)
    xm, ym = np.mean(x), np.mean(y)
    sxx = np.sum((x - xm) ** 2)
    sxy = np.sum((x - xm) * (y - ym))
    if n < 2 or sxx == 0.0:
        return 0.0, ym, float(np.sum((y - ym) ** 2))
    slope = sxy / sxx
    icpt = ym - slope * xm
    return slope, icpt, float(np.sum((y - (slope * x + icpt)) ** 2))


def fit_piecewise(logx, logy, p2_deg=0):
    """Two-piece linear fit of ``logy(logx)``: a general line for the first
    ``i`` points and a constant (``p2_deg=0``) or a line (``p2_deg=1``) for
    the rest; ``i`` minimises the summed squared misfit.  Returns the index of
    the x closest to the intersection and the two polynomials (highest power
    first) -- semantics of reference linefit_analyzer.py:28-87, NaNs in
    ``logy`` dropped like its ``denan``."""
    logx = np.asarray(logx, dtype=float)
    logy = np.asarray(logy, dtype=float)
    n = len(logx)
    misfit = np.full(n, np.nan)
    p1 = [None] * n
    p2 = [None] * n
    ok = np.logical_not(np.isnan(logy))

    def exact(i):
        a, b = ok[:i], ok[i:]
        x1, y1, x2, y2 = logx[:i][a], logy[:i][a], logx[i:][b], logy[i:][b]
        if len(x1) < 1 or len(x2) < 1:
            return
        s1, c1, e1 = _linfit_sse(x1, y1)
        if p2_deg == 1:
            s2, c2, e2 = _linfit_sse(x2, y2)
            p2[i] = np.array([s2, c2])
        else:
            c2 = float(np.mean(y2))
            e2 = float(np.sum((y2 - c2) ** 2))
            p2[i] = np.array([c2])
        p1[i] = np.array([s1, c1])
        misfit[i] = e1 + e2

    # All break points at once from prefix sums (O(n) instead of O(n^2)); these values only
    # select the candidates, which are then evaluated exactly as above, so the chosen index is
    # that of the plain loop (a 16x16 run calls this 256 times with n = 100).
    cands = range(2, n - 2)
    if n > 8:
        w = ok.astype(float)
        x0 = logx - np.mean(logx)                      # centred: less cancellation
        y0 = np.where(ok, logy, 0.0)
        y0 = y0 - (np.sum(y0) / max(np.sum(w), 1.0)) * w
        cs = [np.concatenate(([0.0], np.cumsum(v))) for v in
              (w, w * x0, y0, w * x0 * x0, x0 * y0, y0 * y0)]

        def sse(lo, hi, line):
            m, sx, sy, sxx, sxy, syy = [c[hi] - c[lo] for c in cs]
            with np.errstate(all='ignore'):
                vyy = syy - sy * sy / m
                if not line:
                    return np.where(m >= 1, vyy, np.nan)
                vxx = sxx - sx * sx / m
                vxy = sxy - sx * sy / m
                fit = np.where(vxx > 0, vyy - vxy * vxy / vxx, vyy)
                return np.where(m >= 1, np.where(m >= 2, fit, vyy), np.nan)
        idx = np.arange(2, n - 2)
        approx = sse(np.zeros_like(idx), idx, True) + sse(idx, np.full_like(idx, n), p2_deg == 1)
        if np.any(np.isfinite(approx)):
            scale = np.nanmax(np.abs(cs[5][-1])) + 1e-300
            best = np.nanmin(approx)
            cands = idx[approx <= best + 1e-9 * scale + 1e-6 * abs(best)]
    for i in cands:
        exact(int(i))
    if np.all(np.isnan(misfit)):
        raise ValueError('chi2 is all NaN')
    i = int(np.nanargmin(misfit))
    slope2 = p2[i][0] if p2_deg == 1 else 0.0
    icpt2 = p2[i][1] if p2_deg == 1 else p2[i][0]
    with np.errstate(all='ignore'):
        x_cross = (icpt2 - p1[i][1]) / (p1[i][0] - slope2)
        dist = np.abs(logx - x_cross)
    if np.all(np.isnan(dist)):
        raise ValueError('abs(logx - X_x) is all NaN')
    return int(np.nanargmin(dist)), (p1[i], p2[i])


def fit_piecewise_many(logx, logY, p2_deg=0, max_candidates=6):
    """:func:`fit_piecewise` for the rows of ``logY`` (m x n) at once: the break points of all rows from
    prefix sums, then the exact two-pass misfit of every row's candidate break points, row-parallel.  A row
    whose candidates cannot be settled that way (more than ``max_candidates`` near-ties, nothing finite) is
    handed to the scalar routine.  Returns (index array, -1 where the fit fails; list of (p1, p2) or None)."""
    logx = np.asarray(logx, dtype=float)
    Y = np.asarray(logY, dtype=float)
    m, n = Y.shape
    out_idx = np.full(m, -1, dtype=int)
    out_par = [None] * m
    todo_scalar = np.zeros(m, dtype=bool)
    if n <= 8:
        todo_scalar[:] = True
    else:
        ok = np.logical_not(np.isnan(Y))
        w = ok.astype(float)
        Y0 = np.where(ok, Y, 0.0)
        x0 = logx - np.mean(logx)
        yc = Y0 - (np.sum(Y0, axis=1) / np.maximum(np.sum(w, axis=1), 1.0))[:, None] * w
        zero = np.zeros((m, 1))
        cs = [np.concatenate((zero, np.cumsum(v, axis=1)), axis=1) for v in
              (w, w * x0, yc, w * x0 * x0, x0 * yc, yc * yc)]
        brk = np.arange(2, n - 2)

        def sse(lo, hi, line):
            mm, sx, sy, sxx, sxy, syy = [c[:, hi] - c[:, lo] for c in cs]
            with np.errstate(all='ignore'):
                vyy = syy - sy * sy / mm
                if not line:
                    return np.where(mm >= 1, vyy, np.nan)
                vxx = sxx - sx * sx / mm
                vxy = sxy - sx * sy / mm
                fit = np.where(vxx > 0, vyy - vxy * vxy / vxx, vyy)
                return np.where(mm >= 1, np.where(mm >= 2, fit, vyy), np.nan)
        approx = sse(np.zeros_like(brk), brk, True) + sse(brk, np.full_like(brk, n), p2_deg == 1)
        finite = np.isfinite(approx)
        todo_scalar |= ~finite.any(axis=1)
        with np.errstate(all='ignore'):
            best = np.nanmin(np.where(finite, approx, np.inf), axis=1)
        scale = np.abs(cs[5][:, -1]) + 1e-300
        cand = finite & (approx <= (best + 1e-9 * scale + 1e-6 * np.abs(best))[:, None])
        ncand = cand.sum(axis=1)
        todo_scalar |= ncand > max_candidates
        rows = np.where(~todo_scalar)[0]
        if len(rows):
            ar = np.arange(n)[None, :]
            Yr, okr = Y0[rows], ok[rows]
            xb = np.broadcast_to(logx, Yr.shape)

            def part(mask, line):
                cnt = mask.sum(axis=1)
                safe = np.maximum(cnt, 1)
                xm = (xb * mask).sum(axis=1) / safe
                ym = (Yr * mask).sum(axis=1) / safe
                dx = (xb - xm[:, None]) * mask
                sxx = (dx * dx).sum(axis=1)
                sxy = (dx * (Yr - ym[:, None])).sum(axis=1)
                with np.errstate(all='ignore'):
                    slope = np.where(line & (cnt >= 2) & (sxx != 0.0), sxy / np.where(sxx != 0.0, sxx, 1.0), 0.0)
                icpt = ym - slope * xm
                r = (Yr - (slope[:, None] * xb + icpt[:, None])) * mask
                return slope, icpt, (r * r).sum(axis=1), cnt
            kmax = int(ncand[rows].max())
            order = np.argsort(~cand[rows], axis=1, kind='stable')[:, :kmax]      # candidate positions, ascending
            mis = np.full((len(rows), kmax), np.nan)
            keep = []
            for k in range(kmax):
                bi = brk[order[:, k]]
                valid = k < ncand[rows]
                m1 = (ar < bi[:, None]) & okr
                m2 = (ar >= bi[:, None]) & okr
                s1, c1, e1, n1 = part(m1, np.ones(len(rows), dtype=bool))
                s2, c2, e2, n2 = part(m2, np.full(len(rows), p2_deg == 1))
                good = valid & (n1 >= 1) & (n2 >= 1)
                mis[:, k] = np.where(good, e1 + e2, np.nan)
                keep.append((bi, s1, c1, s2, c2))
            # every row's best candidate and the alpha nearest to where its two lines cross, for all rows at once
            has = ~np.all(np.isnan(mis), axis=1)
            kbest = np.argmin(np.where(np.isnan(mis), np.inf, mis), axis=1)           # (first minimum, like nanargmin)
            pick = [np.stack([kp[i] for kp in keep], axis=1)[np.arange(len(rows)), kbest] for i in range(5)]
            bi_b, s1_b, c1_b, s2_b, c2_b = pick
            with np.errstate(all='ignore'):
                x_cross = (c2_b - c1_b) / (s1_b - (s2_b if p2_deg == 1 else 0.0))
                dist = np.abs(logx[None, :] - x_cross[:, None])
            has &= ~np.all(np.isnan(dist), axis=1)
            nearest = np.argmin(np.where(np.isnan(dist), np.inf, dist), axis=1)
            for q in np.where(has)[0]:
                r = rows[q]
                out_idx[r] = int(nearest[q])
                out_par[r] = (np.array([s1_b[q], c1_b[q]]),
                              np.array([s2_b[q], c2_b[q]]) if p2_deg == 1 else np.array([c2_b[q]]))
    for r in np.where(todo_scalar)[0]:
        try:
            out_idx[r], out_par[r] = fit_piecewise(logx, Y[r], p2_deg)
        except ValueError:
            pass
    return out_idx, out_par


def curv(x, y):
    """curvature y'' / (1 + y'^2)^(3/2) from second-order central
    differences; NaN at both ends (chi2_curvature_analyzer.py:25-49)."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    n = len(x)
    der1 = np.full(n, np.nan)
    der2 = np.full(n, np.nan)
    if n > 2:
        hp = x[2:] - x[1:-1]
        hm = x[1:-1] - x[:-2]
        der2[1:-1] = (y[2:] - 2 * y[1:-1] + y[:-2]) / (hp * hm)
        der1[1:-1] = ((y[2:] - y[1:-1]) / hp + (y[1:-1] - y[:-2]) / hm) / 2
    return der2 / (1 + der1 * der1) ** 1.5, der1, der2


def _element(maxent_result, name, matrix_element):
    """the arrays of ONE element, without assembling the (M, N, ...) array of all
    elements first (``maxent_result.A`` of a 16x16 run is 100 MB)."""
    return maxent_result.element_array(name, matrix_element)


class Picks(object):
    """the alphas one analyzer takes for the elements ``keys`` of a batch, as the device chose them: indices and A rows.
    ``build(n)`` makes the :class:`AnalyzerResult` of element n; :class:`maxent_result.MaxEntResult` keeps the batch
    and builds single results when somebody looks at them (``result.A_out`` reads the rows directly)."""

    def __init__(self, analyzer, maxent_result, keys, idx, rows, fixed, extras, info):
        self.analyzer, self.keys, self.idx, self.rows = analyzer, keys, idx, rows
        self.owner = weakref.ref(maxent_result)      # (the result owns its analyses: no strong reference back)
        self.alpha = np.asarray(maxent_result.alpha)
        self.fixed, self.extras, self.info = fixed, extras, info

    def build(self, n):
        res = AnalyzerResult()
        i, k = self.idx[n], self.keys[n]
        dict.update(res, alpha_index=i, A_out=self.rows[n], name=self.analyzer.name, **self.fixed)
        for name, fn in self.extras.items():
            res.later(name, lambda fn=fn, k=k: fn(self.owner(), k, self.alpha))
        res.later('info', lambda: self.info.format(self.alpha[i], i))
        res.maxent_result = self.owner()
        return res


class Deferred(object):
    """element n of a :class:`Picks` batch, not yet an AnalyzerResult"""
    __slots__ = ('picks', 'n')

    def __init__(self, picks, n):
        self.picks, self.n = picks, n

    def build(self):
        return self.picks.build(self.n)

    def A_out(self):
        return self.picks.rows[self.n]


def _scan_picks(maxent_result, keys):
    """one pass over the records of ``keys`` for all three analyzers of a batch (it was three): the parameters the device used,
    the launch's arrays and the chains of the elements in them, the A maps -- or None when some element has no choice of the
    device (no fit, a record made some other way).  Kept on the result for the ``keys`` list object at hand."""
    held = maxent_result.__dict__.get('_picks_scan')
    if held is not None and held[0] is keys and held[1] == len(keys):
        return held[2]
    recs = getattr(maxent_result, '_records', {})
    sels, maps = [], []
    params, uniform_params = None, True
    out = None
    for k in keys:
        rec = recs.get(k)
        sel = None if rec is None else rec.get('device_select')
        A = None if rec is None else rec.get('A')
        if sel is None or not hasattr(A, 'from_H_row'):
            break
        if params is None:
            params = sel['params']
        elif sel['params'] is not params and sel['params'] != params:
            uniform_params = False
        maps.append(A)
        sels.append(sel)
    else:
        batch = sels[0].get('batch') if sels else None
        if batch is not None and not all(s.get('batch') is batch for s in sels):
            batch = None
        cs = np.fromiter((s['chain'] for s in sels), dtype=np.intp, count=len(sels)) if batch is not None else None
        first = maps[0]._map if maps else None
        same_map = bool(maps) and first.matrix() is None and all(m._map is first for m in maps)
        out = dict(sels=sels, maps=maps, params=params if uniform_params else None, batch=batch, cs=cs,
                   first=first, same_map=same_map)
    maxent_result.__dict__['_picks_scan'] = (keys, len(keys), out)
    return out


def _device_picks(maxent_result, keys, which, matches):
    """What the device chose for these elements (``mxe_select3_launch``, one launch behind the solve; the indices and
    the three H rows of every scan came back in one copy): (indices, A rows), or None when any element has no such
    choice -- another parameter than the device used (``matches(params)``), no fit, a record made some other way.
    ``which``: 0 line fit, 1 chi2 curvature, 2 entropy."""
    scan = _scan_picks(maxent_result, keys)
    if scan is None:
        return None
    sels, maps = scan['sels'], scan['maps']
    if not sels:
        return [], []
    if scan['params'] is not None:
        if not matches(scan['params']):
            return None
    elif not all(matches(s['params']) for s in sels):
        return None
    batch = scan['batch']
    if batch is not None:
        # the launch's arrays as they came off the device: [3][n_chain] indices, [3][n_chain][n_omega] rows
        cs = scan['cs']
        idx = batch[0][which][cs]
        if np.any(idx < 0):
            return None
        src = batch[1][which]
        if not getattr(src, 'on_host', True):
            # (the rows of an analyzer that is not the result's default are still on the device: 1 MB that comes when somebody
            #  looks at this analyzer's A_out)
            return idx.tolist(), DeferredRows(src, cs, scan['first'] if scan['same_map'] else None, maps)
        H = src[cs]
    else:
        idx = np.array([s['index'][which] for s in sels])
        if np.any(idx < 0):
            return None
        H = np.array([s['H'][which] for s in sels])
    return idx.tolist(), _rows_of_H(H, scan['first'] if scan['same_map'] else None, maps)


def _rows_of_H(H, shared_map, maps):
    if shared_map is not None:
        # A = H / delta for everybody: one division for all rows (elementwise: the same bits as row by row)
        return list(shared_map.f(H))
    return [m.from_H_row(r) for m, r in zip(maps, H)]


class DeferredRows(object):
    """the A rows of one analyzer for the elements of a batch, formed from the device's H rows when first looked at"""

    def __init__(self, src, cs, shared_map, maps):
        self._src, self._cs, self._shared, self._maps = src, cs, shared_map, maps
        self._rows = None

    def _build(self):
        if self._rows is None:
            self._rows = _rows_of_H(np.asarray(self._src)[self._cs], self._shared, self._maps)
            self._src = self._maps = None
        return self._rows

    def __len__(self):
        return len(self._cs)

    def __getitem__(self, n):
        return self._build()[n]

    def __iter__(self):
        return iter(self._build())


class LineFitAnalyzer(Analyzer):
    """kink of log chi2 (log alpha)."""

    def __init__(self, linefit_deg=0, name=None):
        self.linefit_deg = linefit_deg
        super(LineFitAnalyzer, self).__init__(name=name)

    def analyze(self, maxent_result, matrix_element=None):
        res = AnalyzerResult()
        alpha = np.asarray(maxent_result.alpha)
        chi2 = np.asarray(_element(maxent_result, 'chi2', matrix_element), dtype=float)
        with np.errstate(all='ignore'):
            idx, params = fit_piecewise(np.log(alpha), np.log(chi2),
                                        self.linefit_deg)
        return self._result(maxent_result, matrix_element, alpha, idx, params)

    def pick_many(self, maxent_result, keys):
        """the device's line fit for these elements (the same two-stage fit: linefit_kernel), or None; the parameters
        of the two lines, which nothing but a plot reads, are fitted when somebody asks for them"""
        dev = _device_picks(maxent_result, keys, 0, lambda p: p[0] == self.linefit_deg)
        if dev is None:
            return None
        return Picks(self, maxent_result, keys, dev[0], dev[1], dict(linefit_deg=self.linefit_deg),
                     dict(linefit_params=self._params), 'Ideal alpha (linefit): {} (= index {} zero-based)')

    def _params(self, maxent_result, matrix_element, alpha):
        chi2 = np.asarray(_element(maxent_result, 'chi2', matrix_element), dtype=float)
        with np.errstate(all='ignore'):
            return fit_piecewise(np.log(alpha), np.log(chi2), self.linefit_deg)[1]

    def _result(self, maxent_result, matrix_element, alpha, idx, params):
        res = AnalyzerResult()
        res['alpha_index'] = idx
        res['linefit_params'] = params
        res['A_out'] = maxent_result.element_row('A', matrix_element, idx)
        res['linefit_deg'] = self.linefit_deg
        res['name'] = self.name
        res['info'] = 'Ideal alpha (linefit): {} (= index {} zero-based)' \
            .format(alpha[idx], idx)
        return res

    def analyze_many(self, maxent_result, keys):
        """one result (or the error message) per key; all break points in one vectorised pass"""
        alpha = np.asarray(maxent_result.alpha)
        picks = self.pick_many(maxent_result, keys)
        if picks is not None:
            return [picks.build(n) for n in range(len(keys))]
        chi2 = np.array([np.asarray(_element(maxent_result, 'chi2', k), dtype=float) for k in keys])
        with np.errstate(all='ignore'):
            idx, params = fit_piecewise_many(np.log(alpha), np.log(chi2), self.linefit_deg)
        return [self._result(maxent_result, k, alpha, int(i), p) if i >= 0 else 'chi2 is all NaN'
                for k, i, p in zip(keys, idx, params)]


class Chi2CurvatureAnalyzer(Analyzer):
    """maximum curvature of log10 chi2 (gamma log10 alpha)."""

    def __init__(self, gamma=0.2, name=None):
        self.gamma = gamma
        super(Chi2CurvatureAnalyzer, self).__init__(name=name)

    def analyze(self, maxent_result, matrix_element=None):
        res = AnalyzerResult()
        alpha = np.asarray(maxent_result.alpha)
        chi2 = np.asarray(_element(maxent_result, 'chi2', matrix_element), dtype=float)
        with np.errstate(all='ignore'):
            res['curvature'], _, _ = curv(self.gamma * np.log10(alpha),
                                          np.log10(chi2))
        return self._finish(res, maxent_result, matrix_element, alpha)

    def analyze_many(self, maxent_result, keys):
        alpha = np.asarray(maxent_result.alpha)
        picks = self.pick_many(maxent_result, keys)
        if picks is not None:
            return [picks.build(n) for n in range(len(keys))]
        x = self.gamma * np.log10(alpha)
        out = []
        with np.errstate(all='ignore'):
            Y = np.log10(np.array([np.asarray(_element(maxent_result, 'chi2', k), dtype=float) for k in keys]))
            hp, hm = x[2:] - x[1:-1], x[1:-1] - x[:-2]
            der2 = (Y[:, 2:] - 2 * Y[:, 1:-1] + Y[:, :-2]) / (hp * hm)
            der1 = ((Y[:, 2:] - Y[:, 1:-1]) / hp + (Y[:, 1:-1] - Y[:, :-2]) / hm) / 2
            c = np.full(Y.shape, np.nan)
            c[:, 1:-1] = der2 / (1 + der1 * der1) ** 1.5
        best = np.argmax(np.where(np.isnan(c), -np.inf, c), axis=1)        # (first maximum, like nanargmax)
        empty = np.all(np.isnan(c), axis=1)
        for k, row, i, e in zip(keys, c, best, empty):
            res = AnalyzerResult()
            res['curvature'] = row
            try:
                out.append(self._finish(res, maxent_result, k, alpha, None if e else int(i)))
            except ValueError as err:
                out.append(str(err))
        return out

    def pick_many(self, maxent_result, keys):
        dev = _device_picks(maxent_result, keys, 1, lambda p: p[1] == self.gamma)
        if dev is None:
            return None
        return Picks(self, maxent_result, keys, dev[0], dev[1], dict(gamma=self.gamma), dict(curvature=self._curve),
                     'Ideal alpha (curvature): {} (= index {} zero-based)')

    def _curve(self, maxent_result, matrix_element, alpha):
        chi2 = np.asarray(_element(maxent_result, 'chi2', matrix_element), dtype=float)
        with np.errstate(all='ignore'):
            return curv(self.gamma * np.log10(alpha), np.log10(chi2))[0]

    def _finish(self, res, maxent_result, matrix_element, alpha, idx=None):
        if idx is None:
            if np.all(np.isnan(res['curvature'])):
                raise ValueError('curvature is all NaN')
            idx = int(np.nanargmax(res['curvature']))
        res['alpha_index'] = idx
        res['A_out'] = maxent_result.element_row('A', matrix_element, idx)
        res['gamma'] = self.gamma
        res['name'] = self.name
        res['info'] = 'Ideal alpha (curvature): {} (= index {} zero-based)' \
            .format(alpha[idx], idx)
        return res


class EntropyAnalyzer(Analyzer):
    """flattest point of S(log alpha)."""

    def analyze(self, maxent_result, matrix_element=None):
        res = AnalyzerResult()
        alpha = np.asarray(maxent_result.alpha)
        S = np.asarray(_element(maxent_result, 'S', matrix_element), dtype=float)
        d = np.full(len(alpha), np.nan)
        d[1:-1] = (S[2:] - S[:-2]) / (np.log(alpha[2:]) - np.log(alpha[:-2]))
        res['dS_dalpha'] = d
        return self._finish(res, maxent_result, matrix_element, alpha)

    def analyze_many(self, maxent_result, keys):
        alpha = np.asarray(maxent_result.alpha)
        picks = self.pick_many(maxent_result, keys)
        if picks is not None:
            return [picks.build(n) for n in range(len(keys))]
        S = np.array([np.asarray(_element(maxent_result, 'S', k), dtype=float) for k in keys])
        D = np.full(S.shape, np.nan)
        D[:, 1:-1] = (S[:, 2:] - S[:, :-2]) / (np.log(alpha[2:]) - np.log(alpha[:-2]))
        out = []
        best = np.argmin(np.where(np.isnan(D), np.inf, D ** 2), axis=1)       # (first minimum, like nanargmin)
        empty = np.all(np.isnan(D), axis=1)
        for k, row, i, e in zip(keys, D, best, empty):
            res = AnalyzerResult()
            res['dS_dalpha'] = row
            try:
                out.append(self._finish(res, maxent_result, k, alpha, None if e else int(i)))
            except ValueError as err:
                out.append(str(err))
        return out

    def pick_many(self, maxent_result, keys):
        dev = _device_picks(maxent_result, keys, 2, lambda p: True)
        if dev is None:
            return None
        return Picks(self, maxent_result, keys, dev[0], dev[1], {}, dict(dS_dalpha=self._slope),
                     'Ideal alpha (entropy): {} (= index {} zero-based)')

    def _slope(self, maxent_result, matrix_element, alpha):
        S = np.asarray(_element(maxent_result, 'S', matrix_element), dtype=float)
        d = np.full(len(alpha), np.nan)
        d[1:-1] = (S[2:] - S[:-2]) / (np.log(alpha[2:]) - np.log(alpha[:-2]))
        return d

    def _finish(self, res, maxent_result, matrix_element, alpha, idx=None):
        d = res['dS_dalpha']
        if idx is None:
            if np.all(np.isnan(d)):
                raise ValueError('dS_dalpha is all NaN')
            idx = int(np.nanargmin(d ** 2))
        res['alpha_index'] = idx
        res['A_out'] = maxent_result.element_row('A', matrix_element, idx)
        res['name'] = self.name
        res['info'] = 'Ideal alpha (entropy): {} (= index {} zero-based)' \
            .format(alpha[idx], idx)
        return res


def get_delta(v):
    d = np.empty(len(v))
    d[1:-1] = (v[2:] - v[:-2]) / 2.0
    d[0] = (v[1] - v[0]) / 2.0
    d[-1] = (v[-1] - v[-2]) / 2.0
    return d


class BryanAnalyzer(Analyzer):
    """average of A_alpha weighted by p(alpha)."""

    def __init__(self, average_by_integration=False, name=None):
        self.average_by_integration = average_by_integration
        super(BryanAnalyzer, self).__init__(name=name)

    def analyze(self, maxent_result, matrix_element=None):
        res = AnalyzerResult()
        res['name'] = self.name
        logp = np.asarray(_element(maxent_result, 'probability', matrix_element), dtype=float)
        if np.all(np.isnan(logp)):
            res['info'] = 'Probability not calculated. Cannot use BryanAnalyzer.'
            return res
        alpha = np.asarray(maxent_result.alpha)
        A = _element(maxent_result, 'A', matrix_element)
        good = np.logical_not(np.isnan(logp))
        p = np.exp(logp[good] - np.nanmax(logp))
        if self.average_by_integration:
            p = p / np.trapezoid(p, alpha[good])
            p = p * get_delta(alpha[good])
        else:
            p = p / np.sum(p)
        res['A_out'] = np.dot(p, np.asarray(A)[good])
        res['info'] = 'Bryan analyzer: average of A weighted by probability calculated.'
        return res


class ClassicAnalyzer(Analyzer):
    """A at the maximum of p(alpha)."""

    def analyze(self, maxent_result, matrix_element=None):
        res = AnalyzerResult()
        res['name'] = self.name
        logp = np.asarray(_element(maxent_result, 'probability', matrix_element), dtype=float)
        if np.all(np.isnan(logp)):
            res['info'] = 'Probability not calculated. Cannot use ClassicAnalyzer.'
            return res
        idx = int(np.nanargmax(logp))
        res['alpha_index'] = idx
        res['A_out'] = maxent_result.element_row('A', matrix_element, idx)
        res['info'] = 'Ideal alpha (classic): {} (= index {} zero-based)' \
            .format(np.asarray(maxent_result.alpha)[idx], idx)
        return res
