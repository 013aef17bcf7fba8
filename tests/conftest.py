import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--no-caching-allocator", action="store_true", default=False,
                     help="out-of-bounds probe (GPU box): PYTORCH_NO_CUDA_MEMORY_CACHING=1 -- every tensor its own hipMalloc, so a kernel that reads "
                          "or writes behind a tensor lands on an unmapped page far more often than inside the caching allocator's arenas; "
                          "hipGraph tests are deselected (capture cannot allocate).  tools/profile_round.sh runs the suite once in this mode.")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--no-caching-allocator"):
        os.environ["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"      # read when torch initialises its device allocator (first use, after this hook)


def pytest_collection_modifyitems(config, items):
    if not config.getoption("--no-caching-allocator"):
        return
    skip = pytest.mark.skip(reason="--no-caching-allocator: graph capture cannot hipMalloc")
    for it in items:
        if "graph" in it.name.lower():
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
