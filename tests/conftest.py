import os
import sys

import pytest

# harness side: one BLAS thread under the cgroup CPU quota of the GPU boxes; importing the library itself has no side effects
os.environ.setdefault("STOCS_PIN_BLAS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def tiny(oracle_lib):
    from model_matching_amd import synth
    m, s, k = synth.workload("tiny")
    o = oracle_lib.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm)
    return m, s, k, o
