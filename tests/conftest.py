import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # a fresh checkout has no built library (it is git-ignored): build it once, like
    # __graft_entry__.build() does (hipcc cross-compiles for gfx950 without a GPU)
    lib = os.path.join(ROOT, 'maxent_amd', 'lib', 'libmaxent_hip.so')
    if not os.path.exists(lib):
        import subprocess
        try:
            subprocess.call(['make', '-C', os.path.join(ROOT, 'maxent_amd', 'csrc')],
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except OSError:
            pass
