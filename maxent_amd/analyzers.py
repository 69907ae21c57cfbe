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

import numpy as np


class AnalyzerResult(dict):
    """dict with the keys ``A_out``, ``name``, ``info`` and, where it applies,
    ``alpha_index`` (reference analyzers/analyzer.py:25-46)."""

    def __reduce_to_dict__(self):
        return self

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
        res['alpha_index'] = idx
        res['linefit_params'] = params
        res['A_out'] = _element(maxent_result, 'A', matrix_element)[idx]
        res['linefit_deg'] = self.linefit_deg
        res['name'] = self.name
        res['info'] = 'Ideal alpha (linefit): {} (= index {} zero-based)' \
            .format(alpha[idx], idx)
        return res


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
        if np.all(np.isnan(res['curvature'])):
            raise ValueError('curvature is all NaN')
        idx = int(np.nanargmax(res['curvature']))
        res['alpha_index'] = idx
        res['A_out'] = _element(maxent_result, 'A', matrix_element)[idx]
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
        if np.all(np.isnan(d)):
            raise ValueError('dS_dalpha is all NaN')
        idx = int(np.nanargmin(d ** 2))
        res['alpha_index'] = idx
        res['A_out'] = _element(maxent_result, 'A', matrix_element)[idx]
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
        res['A_out'] = _element(maxent_result, 'A', matrix_element)[idx]
        res['info'] = 'Ideal alpha (classic): {} (= index {} zero-based)' \
            .format(np.asarray(maxent_result.alpha)[idx], idx)
        return res
