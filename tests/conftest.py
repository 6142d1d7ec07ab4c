import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def oracle_lib():
    """oracle/liboracle.so, built on demand (gcc only; no GPU)."""
    so = os.path.join(ROOT, 'oracle', 'liboracle.so')
    src = os.path.join(ROOT, 'oracle', 'ssn_oracle.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'liboracle.so'])
    from oracle import ssn_numpy
    return ssn_numpy.load_oracle_lib()


@pytest.fixture(scope='session')
def reference_lib():
    """oracle/_ref/libssnode.so (reference C file compiled unmodified); None if absent."""
    from oracle import ssn_numpy
    return ssn_numpy.load_reference_lib()


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
