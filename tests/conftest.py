import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_artefacts_built():
    """The in-tree .so files are build products (git-ignored): compile them when a fresh checkout runs the suite.
    build() is a no-op when they are newer than their sources (hipcc cross-compiles gfx950 without a GPU, about a minute)."""
    import __graft_entry__ as g

    g.build()


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        d = json.load(f)
    rings = dict(d["rings"])
    rings["decomposition"] = d["decomposition"]   # balanced-decomposition KATs ("next" row 2) ride along under their own key
    rings["monomial"] = d["monomial"]             # monomial.rs tests ("next" row 4)
    return rings
