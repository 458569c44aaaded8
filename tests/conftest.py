import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle(threads=min(8, os.cpu_count() or 1))


@pytest.fixture(scope='session')
def host():
    from spacedust_amd.api import Host
    return Host()


@pytest.fixture(scope='session')
def gpu():
    from spacedust_amd.api import Context
    return Context(0)


@pytest.fixture(scope='session')
def small_proteomes():
    from spacedust_amd.synth import make_proteomes
    return make_proteomes(3, genes_per_proteome=120, n_families=200, seed=11)
