import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "sanitizer: emulation tests re-run under AddressSanitizer+UBSan / ThreadSanitizer "
                                       "(minutes; opt-in: -m sanitizer, or RMH_RUN_SANITIZERS=1)")


def pytest_collection_modifyitems(config, items):
    """the sanitizer runs are opt-in: they take minutes and are for the CPU box only"""
    if "sanitizer" in (config.getoption("-m") or "") or os.environ.get("RMH_RUN_SANITIZERS") == "1":
        return
    skip = pytest.mark.skip(reason="opt-in: python -m pytest tests -m sanitizer (or RMH_RUN_SANITIZERS=1)")
    for item in items:
        if "sanitizer" in item.keywords:
            item.add_marker(skip)
