import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """When GPU tests are selected, initialise torch's HIP context once, before the C-ABI library
    creates its own (both share the primary context; doing it up front keeps the order fixed)."""
    if any("gpu" in it.keywords for it in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
