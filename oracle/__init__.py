"""CPU oracle for the alpha-scan inner solver of TRIQS/maxent.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  ``maxent_amd`` never
imports this package and fails loudly when its HIP library is missing.

Modules
-------
ref_numpy    step-faithful numpy restatement of the reference's Python path
             (kernel fill, SVD truncation, chi2 / entropy / H(v) functions,
             MaxEntCostFunction / BryanCostFunction, LevenbergMinimizer and
             the warm-started alpha loop).  Every function cites the reference
             file:line it follows.  Pinned against the imported reference by
             ``tests/golden/make_golden.py`` (identical per-alpha iteration
             counts, fields to <=1e-12) and against the reference's own
             known-answer tests (``tests/test_oracle_golden.py``).
sform        the same mathematics in the whitened singular-space form the HIP
             kernel computes in (chi2 as a sum of squares in the rotated data
             space, diagonal M), plus a numpy model of the kernel's damped
             Newton iteration, used to localise GPU/CPU differences.

Parity status: PINNED (golden vectors generated from the imported reference
are committed under tests/golden/, with the generating script).
"""
