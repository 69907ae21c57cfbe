"""maxent_amd -- MI355X-native alpha-scan solver behind the TRIQS/maxent API.

The public names are the reference's (reference python/__init__.py:21-38):
``TauMaxEnt``, ``ElementwiseMaxEnt``, ``DiagonalMaxEnt``, ``PoormanMaxEnt``,
``MaxEntLoop``, ``MaxEntResult``, the omega / alpha meshes, default models,
kernels, cost functions, minimiser, analyzers and the probability.  The inner
solver runs in ``libmaxent_hip.so`` (hand-written HIP for gfx950) reached
through ctypes (``maxent_amd.device``); there is no CPU fallback.
"""

from .omega_meshes import *            # noqa: F401,F403
from .alpha_meshes import *            # noqa: F401,F403
from .default_models import *          # noqa: F401,F403
from .preblur import get_preblur       # noqa: F401
from .kernels import (KernelSVD, Kernel, DataKernel, TauKernel,   # noqa: F401
                      PreblurKernel)
from .functions import (GenericFunction, DoublyDerivableFunction, cached,   # noqa: F401
                        NormalChi2, NormalEntropy, PlusMinusEntropy,
                        NormalH_of_v, PlusMinusH_of_v, IdentityA_of_H,
                        PreblurA_of_H)
from .cost_functions import (CostFunction, MaxEntCostFunction,    # noqa: F401
                             BryanCostFunction)
from .minimizers import (Minimizer, LevenbergMinimizer, ConvergenceMethod,   # noqa: F401
                         AndConvergenceMethod, OrConvergenceMethod,
                         MaxDerivativeConvergenceMethod, FunctionChangeConvergenceMethod,
                         RelativeFunctionChangeConvergenceMethod,
                         NewtonStepConvergenceMethod, NullConvergenceMethod)
from .analyzers import (Analyzer, AnalyzerResult, LineFitAnalyzer,   # noqa: F401
                        Chi2CurvatureAnalyzer, EntropyAnalyzer, BryanAnalyzer,
                        ClassicAnalyzer)
from .probabilities import NormalLogProbability     # noqa: F401
from .logtaker import Logtaker, VerbosityFlags      # noqa: F401
from .maxent_result import MaxEntResult, MaxEntResultData   # noqa: F401
from .maxent_loop import MaxEntLoop                 # noqa: F401
from .tau_maxent import TauMaxEnt                   # noqa: F401
from .elementwise_maxent import (ElementwiseMaxEnt, DiagonalMaxEnt,   # noqa: F401
                                 PoormanMaxEnt, PendingRun, run_many)
from .device import MaxEntDeviceError, device_count  # noqa: F401
from . import _layout as _layout        # the reference's sub-module paths (analyzers.linefit_analyzer, ...)
_layout.register(__name__)

__version__ = '0.1.0'
