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


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the library (hipcc
    cross-compiles gfx950 without a GPU, ~35 s) and the test oracle once, as __graft_entry__.build() does."""
    lib = os.environ.get("RT_MI355X_LIB") or os.path.join(ROOT, "raytracertest_amd", "lib", "librt_mi355x.so")
    orc_lib = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(lib) or not os.path.exists(orc_lib):
        import __graft_entry__
        __graft_entry__.build()


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
