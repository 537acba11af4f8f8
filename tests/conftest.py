import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_kats():
    with open(os.path.join(GOLDEN, "triangle_hit_kats.json")) as f:
        cases = json.load(f)["cases"]
    eps = float(np.finfo(np.float32).eps)

    def val(x):
        return -eps if x == "-eps" else float(x)
    for c in cases:
        for k in ("origin", "dir", "a", "b", "c"):
            c[k] = np.array([val(x) for x in c[k]], np.float32)
    return cases


@pytest.fixture(scope="session")
def kats():
    return load_kats()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py
