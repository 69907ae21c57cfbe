"""Verbosity flags and message sink (reference python/logtaker.py:25-230).

Same contract as the reference's ``Logtaker``: a message carries a set of flags and is shown when ALL of
them are switched on in ``verbose`` (so flag 0, ``Quiet``, always shows); errors are kept in a list and
shown with an ``ERROR: `` prefix under the ``Errors`` flag; a log file, opened with ``open_logfile``, gets
every message its own level (``logfile_verbose``, by default the terminal's) lets through, one per line;
messages of the kinds in ``one_line`` (the per-iteration solver details) overwrite each other on the
terminal instead of scrolling.  What the alpha-scan path prints through it: the header, the per-alpha line
``alpha[i] = ..., chi2 = ..., n_iter=...`` (reference maxent_loop.py:248-255), timing and errors.
"""

from __future__ import print_function

import sys
from datetime import datetime


class VerbosityFlags(object):
    """bit flags of the message kinds (reference logtaker.py:64-79)"""
    Quiet, Header, ElementInfo, Timing, AlphaLoop, SolverDetails, Errors = 0, 1, 2, 4, 8, 16, 32
    Default = Header | ElementInfo | Timing | AlphaLoop | Errors


class _Terminal(object):
    """stdout with the memory of whether the last thing written left the cursor inside a line"""

    def __init__(self):
        self.inside_line = False

    def show(self, text, overwrite):
        out = sys.stdout
        if overwrite:
            out.write('\r' + text)
            self.inside_line = True
        else:
            if self.inside_line:
                out.write('\n')
                self.inside_line = False
            out.write(text + '\n')


class Logtaker(object):
    def __init__(self, verbose=VerbosityFlags.Default, logfile=None):
        self.verbose = verbose
        self.one_line = VerbosityFlags.SolverDetails
        self.logfile = None
        self.logfile_verbose = None          # None: the terminal's level
        self._errors = []
        self._terminal = _Terminal()
        self._welcomed = False
        if logfile is not None:
            self.open_logfile(logfile)

    # ---- where messages go ---------------------------------------------------
    def open_logfile(self, name, append=True):
        self.close_logfile()
        self.logfile = open(name, {True: 'a', False: 'w'}[bool(append)])

    def close_logfile(self):
        handle, self.logfile = self.logfile, None
        if handle is not None:
            handle.close()

    @staticmethod
    def _passes(level, flags):
        return (level & flags) == flags

    def wants(self, message_verbosity):
        """would a message of this verbosity go anywhere (terminal or log file)?  Callers with many lines to format ask first"""
        if self._passes(self.verbose, message_verbosity):
            return True
        if self.logfile is None:
            return False
        return self._passes(self.verbose if self.logfile_verbose is None else self.logfile_verbose, message_verbosity)

    def message(self, message_verbosity, msg, *args, **kwargs):
        to_terminal = self._passes(self.verbose, message_verbosity)
        to_file = self.logfile is not None and self._passes(
            self.verbose if self.logfile_verbose is None else self.logfile_verbose, message_verbosity)
        if not (to_terminal or to_file):
            return                               # (nothing shows it: it is not formatted either)
        text = msg.format(*args, **kwargs) if (args or kwargs) else msg
        if to_terminal:
            self._terminal.show(text, overwrite=bool(message_verbosity & self.one_line))
        if to_file:
            self.logfile.write(text + '\n')

    def logged_message(self, msg, *args, **kwargs):
        """(deprecated in the reference: a message that always shows)"""
        self.message(VerbosityFlags.Quiet, msg, *args, **kwargs)

    # ---- errors -----------------------------------------------------------------
    def error_message(self, msg, *args, **kwargs):
        self._errors.append(msg.format(*args, **kwargs) if (args or kwargs) else msg)
        self.message(VerbosityFlags.Errors, 'ERROR: ' + msg, *args, **kwargs)

    def get_error_messages(self):
        return self._errors

    def clear_error_messages(self):
        del self._errors[:]

    # ---- fixed texts ------------------------------------------------------------
    def log_time(self, message_verbosity=VerbosityFlags.Header):
        self.message(message_verbosity, str(datetime.now()))

    def welcome_message(self, always=False, message_verbosity=VerbosityFlags.Header):
        if self._welcomed and not always:
            return
        self.log_time(message_verbosity)
        self.message(message_verbosity, 'MaxEnt alpha scan on MI355X (maxent_amd)')
        self._welcomed = True

    def solver_verbose_callback(self, *args, **kwargs):
        self.message(VerbosityFlags.SolverDetails, *args, **kwargs)
