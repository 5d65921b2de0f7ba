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
    config.addinivalue_line("markers", "perf: absolute-throughput claims (README/DESIGN: >= 10x the V100 figure); needs an "
                                       "MI355X, run with -m perf -- deliberately NOT part of -m gpu, whose tests assert "
                                       "results and orderings only, so a slow box cannot fail the correctness suite")


# Collection order of the GPU suite: the parity tests proper first, the lifecycle / failure-injection cases last (each
# of those runs in a child process of its own, tests/test_gpu_hazards.py, so a fault there is ONE failed test and cannot
# erase the results of the tests before it).
_ORDER = ["test_gpu_parity", "test_gpu_readme_pin", "test_gpu_streams", "test_gpu_pipeline", "test_gpu_executables",
          "test_gpu_bench", "test_gpu_hazards"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else -1          # everything else (the CPU suite) keeps its place in front
    items.sort(key=rank)                                              # stable: order inside a file is unchanged
    if "perf" not in (config.getoption("-m") or ""):                  # perf claims run only when asked for by name (-m perf)
        skip = pytest.mark.skip(reason="absolute-throughput claim: run with -m perf on an MI355X")
        for item in items:
            if "perf" in item.keywords:
                item.add_marker(skip)


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
