"""The reference's sub-module paths.

The reference spreads analyzers, cost functions and minimisers over packages
(``triqs_maxent.analyzers.linefit_analyzer``, ``.cost_functions.maxent_cost_function``,
``.minimizers.levenberg_minimizer`` ...; reference python/analyzers/, cost_functions/, minimizers/);
here each family is one module.  Scripts that import from the long paths keep working: the long
names are registered as modules that hold the same objects.
"""

import sys
import types

from . import analyzers, cost_functions, minimizers

_TABLE = {
    'analyzers.analyzer': (analyzers, ('Analyzer', 'AnalyzerResult')),
    'analyzers.linefit_analyzer': (analyzers, ('LineFitAnalyzer', 'fit_piecewise', 'Analyzer', 'AnalyzerResult')),
    'analyzers.chi2_curvature_analyzer': (analyzers, ('Chi2CurvatureAnalyzer', 'curv', 'Analyzer', 'AnalyzerResult')),
    'analyzers.entropy_analyzer': (analyzers, ('EntropyAnalyzer', 'Analyzer', 'AnalyzerResult')),
    'analyzers.bryan_analyzer': (analyzers, ('BryanAnalyzer', 'Analyzer', 'AnalyzerResult')),
    'analyzers.classic_analyzer': (analyzers, ('ClassicAnalyzer', 'Analyzer', 'AnalyzerResult')),
    'cost_functions.cost_function': (cost_functions, ('CostFunction',)),
    'cost_functions.maxent_cost_function': (cost_functions, ('MaxEntCostFunction', 'CostFunction')),
    'cost_functions.bryan_cost_function': (cost_functions, ('BryanCostFunction', 'CostFunction')),
    'minimizers.minimizer': (minimizers, ('Minimizer',)),
    'minimizers.levenberg_minimizer': (minimizers, ('LevenbergMinimizer', 'Minimizer')),
    'minimizers.convergence_methods': (minimizers, ('ConvergenceMethod', 'AndConvergenceMethod', 'OrConvergenceMethod',
                                                    'MaxDerivativeConvergenceMethod', 'FunctionChangeConvergenceMethod',
                                                    'RelativeFunctionChangeConvergenceMethod',
                                                    'NullConvergenceMethod', 'NewtonStepConvergenceMethod')),
}


def register(package):
    for tail, (source, names) in _TABLE.items():
        full = package + '.' + tail
        mod = types.ModuleType(full, 'alias of %s (reference layout)' % source.__name__)
        for n in names:
            setattr(mod, n, getattr(source, n))
        mod.__all__ = list(names)
        sys.modules[full] = mod
        setattr(source, tail.split('.')[1], mod)
