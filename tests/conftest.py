import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import bp_oracle

    return bp_oracle


@pytest.fixture(scope="session")
def coracle():
    import c_oracle

    return c_oracle.COracle()


@pytest.fixture(scope="session")
def ctx():
    import bulletproofs_amd as bp

    c = bp.Context(0)
    yield c
    c.close()
