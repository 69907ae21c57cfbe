"""MaxEntResultData's dict form (the h5 layout of the reference, python/maxent_result.py:181-188, 616-685):
key set, timedelta encoding, round trip, hermitian mirror -- host only."""
import pickle
from datetime import timedelta

import numpy as np

import maxent_amd as mx
from maxent_amd.maxent_result import MaxEntResult, MaxEntResultData

REFERENCE_FIELDS = ['alpha', 'v', 'chi2', 'S', 'A', 'Q', 'omega', 'probability', 'analyzer_results', 'run_times',
                    'run_time_total', 'matrix_structure', 'effective_matrix_structure', 'element_wise',
                    'complex_elements', 'use_hermiticity', 'G', 'data_variable', 'G_rec', 'H',
                    'default_analyzer_name', 'zero_elements', 'G_orig']        # maxent_result.py:181-188


def filled_result():
    rng = np.random.RandomState(5)
    n_alpha, n_omega, n_tau, n_s = 9, 12, 7, 4
    omega = mx.LinearOmegaMesh(-2, 2, n_omega)
    alpha = np.logspace(2, -1, n_alpha)
    res = MaxEntResult(matrix_structure=(2, 2), element_wise=True, use_hermiticity=True)
    res._default_analyzer_name = 'LineFitAnalyzer'
    for key in ((0, 0), (0, 1), (1, 1)):
        chi2 = 10.0 + 1e3 * alpha ** 1.5 * (1 + 0.01 * rng.rand(n_alpha))
        H = rng.rand(n_alpha, n_omega)
        rec = dict(alpha=alpha, v=rng.randn(n_alpha, n_s), H=H, A=H / omega.delta, chi2=chi2, S=-rng.rand(n_alpha),
                   Q=rng.rand(n_alpha), G=rng.randn(n_tau), G_orig=rng.randn(n_tau), data_variable=np.arange(n_tau) * 1.0,
                   G_rec=rng.randn(n_alpha, n_tau), omega=omega, probability=np.full(n_alpha, np.nan),
                   run_times=[timedelta(microseconds=3)] * n_alpha)
        res.start_timing(key)
        res.add_element_results(rec, key)
        res.end_timing(key)
    res.analyze_batch([mx.LineFitAnalyzer(), mx.Chi2CurvatureAnalyzer()], [(0, 0), (0, 1), (1, 1)])
    return res


def test_dict_has_the_reference_key_set_and_round_trips():
    res = filled_result()
    d = res.__reduce_to_dict__()
    assert set(d) == set(REFERENCE_FIELDS) | {'all_fields'}
    assert d['all_fields'] == REFERENCE_FIELDS
    # timedeltas travel as dicts of days / seconds / microseconds, nested like the matrix
    assert d['run_times'][0][0][0] == dict(days=0, seconds=0, microseconds=3)
    assert set(d['run_time_total'][1][1]) == {'days', 'seconds', 'microseconds'}
    assert d['matrix_structure'] == (2, 2) and d['effective_matrix_structure'] == (2, 2)
    assert d['A'].shape == (2, 2, 9, 12) and d['chi2'].shape == (2, 2, 9)
    back = MaxEntResultData.__factory_from_dict__('MaxEntResultData', dict(d))
    for name in ('alpha', 'v', 'chi2', 'S', 'A', 'Q', 'H', 'G', 'G_orig', 'G_rec', 'data_variable'):
        np.testing.assert_array_equal(np.asarray(getattr(back, name)), np.asarray(getattr(res, name)))
    assert isinstance(back.omega, mx.DataOmegaMesh) and isinstance(back.alpha, mx.DataAlphaMesh)
    assert back.run_times[0][1][2] == timedelta(microseconds=3)
    assert back.matrix_structure == (2, 2) and back.use_hermiticity and back.element_wise
    np.testing.assert_array_equal(back.A_out, res.A_out)
    assert back.analyzer_results[0][1]['LineFitAnalyzer']['alpha_index'] == \
        res.analyzer_results[0][1]['LineFitAnalyzer']['alpha_index']


def test_missing_element_is_the_hermitian_mirror_and_fields_can_be_excluded():
    res = filled_result()
    assert np.array_equal(res.A[1, 0], res.A[0, 1]) and np.array_equal(res.A_out[1, 0], res.A_out[0, 1])
    assert np.all(np.isnan(res.chi2[1, 0]))
    res.exclude(['G_rec', 'H'])
    d = res.__reduce_to_dict__()
    assert 'H' not in d and 'G_rec' not in d and 'A' in d
    data = pickle.loads(pickle.dumps(res.data))
    np.testing.assert_array_equal(data.A, res.A)
    assert data.default_analyzer_name == 'LineFitAnalyzer'
