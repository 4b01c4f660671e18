import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # tools/asan_host.sh points the CPU suite at a sanitizer build of the host code (test infrastructure only)
    alt = os.environ.get("FLEX_TEST_LIB")
    if alt:
        from flex_amd import binding
        binding._SO = alt


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "golden.npz"))


@pytest.fixture(scope="session")
def pubmed():
    import oracle
    return oracle.csv_load(os.path.join(GOLDEN, "pubmed.csv"))


@pytest.fixture(scope="session")
def a_mat():
    import oracle
    return oracle.csv_load(os.path.join(GOLDEN, "a_mat.csv"))


class _Knobs:
    """Plan-time knobs (fields of flex_plan_tuning) for every Plan() created while a test runs, also inside helpers:
    knobs.set(two_d=1, panel_kb=32); knobs.clear("two_d").  Reset when the test ends."""

    def set(self, **kw):
        from flex_amd import binding
        binding.DEFAULT_TUNING.update({k: int(v) for k, v in kw.items()})

    def clear(self, *names):
        from flex_amd import binding
        for n in names or list(binding.DEFAULT_TUNING):
            binding.DEFAULT_TUNING.pop(n, None)


@pytest.fixture
def knobs():
    k = _Knobs()
    k.clear()
    yield k
    k.clear()
