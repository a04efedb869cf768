import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope='session')
def engine0():
    """The HIP engine on cuda:0; GPU tests fail (not skip) when the native library is missing."""
    from simplyp_amd import engine
    return engine.get_engine(0)
