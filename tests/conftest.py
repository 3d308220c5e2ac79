import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def gate_cases():
    z = load_golden("gate_cases.npz")
    names = [str(n) for n in z["__cases__"]]
    return {n: {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


@pytest.fixture(scope="session")
def layer_small():
    return load_golden("layer_small.npz")


@pytest.fixture(scope="session")
def layer_cfg1():
    return load_golden("layer_cfg1.npz")


@pytest.fixture(scope="session")
def ep_small():
    return load_golden("ep_small.npz")
