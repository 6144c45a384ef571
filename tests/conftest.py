import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    # /dev/kfd is how ROCm sees a GPU; do not initialise HIP just to decide a skip
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":  # dry-run of the GPU tests' logic on the CPU test double
        return True
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def product_libraries_present():
    """A checkout without the built (git-ignored) libraries: compile them ONCE, before any test or worker process
    loads them.  Not a fallback -- the same hipcc / g++ build as __graft_entry__.build(); stale-but-present libraries are
    left alone here so that concurrent worker processes can never race a rebuild."""
    from rust_lbfgs_amd import _build

    if not (os.path.exists(_build.HIP_LIB) and os.path.exists(_build.SOLVER_LIB)):
        _build.build_all()
    yield
