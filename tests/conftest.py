import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def golden_forest(z):
    from pastml_amd.tree import FlatForest
    return FlatForest(z['parent'], z['n_children'], z['first_child'], z['dist'], np.arange(int(z['n_roots'])))


def golden_spec(z, prefix=''):
    """Model description dict (oracle / kernel_spec form) + (sf, tau, tau_factor) from a golden file."""
    name = str(z[prefix + 'model_name'])
    rates = (float(z[prefix + 'sf']), float(z[prefix + 'tau']), float(z[prefix + 'tau_factor']))
    pi = z[prefix + 'frequencies']
    if name in ('F81', 'JC', 'EFT'):
        spec = dict(kind=0, pi=pi)
    elif name == 'HKY':
        spec = dict(kind=1, pi=pi, kappa=float(z[prefix + 'kappa']))
    else:
        spec = dict(kind=2, pi=pi, d=z[prefix + 'eig_d'], A=z[prefix + 'eig_A'], Ainv=z[prefix + 'eig_Ainv'])
    return spec, rates


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
