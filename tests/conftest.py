import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "drl-on-robot-arm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def golden_npz(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def kuka(O):
    return O.make_chain("kuka")


@pytest.fixture(scope="session")
def diana(O):
    return O.make_chain("diana")
