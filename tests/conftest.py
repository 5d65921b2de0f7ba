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


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "oracle_vectors.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def rocrand_vectors():
    with open(os.path.join(GOLDEN, "rocrand_vectors.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle_mod():
    import hw_oracle
    hw_oracle.build()
    return hw_oracle


@pytest.fixture(scope="session")
def hwmc_mod():
    import hwmc
    hwmc.build_library()
    return hwmc
