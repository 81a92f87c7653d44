import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pt():
    import pathtracer_amd
    pathtracer_amd.lib()
    return pathtracer_amd


@pytest.fixture(scope="session")
def ol():
    import oracle_lib
    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False))
        return cache[name]

    return load
