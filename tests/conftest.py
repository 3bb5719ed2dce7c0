import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """Both shared libraries are git-ignored build products: (re)build them when sources are newer (make decides),
    so that a fresh checkout can run the suite. hipcc cross-compiles for gfx950 without a GPU."""
    import subprocess
    for d in (os.path.join(ROOT, 'axtrack_amd', 'csrc'), os.path.join(ROOT, 'oracle')):
        r = subprocess.run(['make', '-C', d, '-s', '-j8'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:               # never run the suite against a stale library
            pytest.exit(f'building {d} failed:\n{r.stdout[-4000:]}', returncode=2)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


@pytest.fixture(scope='session')
def weights():
    from axtrack_amd import synth
    return synth.synth_state_dict(42)
