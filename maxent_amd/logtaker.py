"""Verbosity flags and message sink (reference python/logtaker.py:25-230).

Only what the alpha-scan path prints is kept: the header, the per-alpha line
``alpha[i] = ..., chi2 = ..., n_iter=...`` (reference maxent_loop.py:248-255),
timing and error messages, each behind the same bit flags.
"""

import sys
from datetime import datetime


class VerbosityFlags(object):
    Quiet = 0
    Header = 1
    ElementInfo = 2
    Timing = 4
    AlphaLoop = 8
    SolverDetails = 16
    Errors = 32
    Default = Header | ElementInfo | Timing | AlphaLoop | Errors


class Logtaker(object):
    """print and/or append to a log file, filtered by ``verbose``."""

    def __init__(self, verbose=VerbosityFlags.Default, logfile=None):
        self.verbose = verbose
        self.logfile = logfile
        self._welcomed = False

    def _emit(self, text, stream=None):
        print(text, file=stream or sys.stdout)
        if self.logfile is not None:
            with open(self.logfile, 'a') as f:
                f.write(text + '\n')

    def message(self, flag, fmt, *args):
        if self.verbose & flag:
            self._emit(fmt.format(*args) if args else fmt)

    def error_message(self, fmt, *args):
        if self.verbose & VerbosityFlags.Errors:
            self._emit(fmt.format(*args) if args else fmt, sys.stderr)

    def welcome_message(self):
        if self.verbose & VerbosityFlags.Header:
            self._emit('{}\nMaxEnt alpha scan on MI355X (maxent_amd)'.format(
                datetime.now().strftime('%Y-%m-%d %H:%M:%S')))

    def solver_verbose_callback(self, msg):
        self.message(VerbosityFlags.SolverDetails, '{}', msg)
