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


def pytest_sessionfinish(session, exitstatus):
    """worst stage-vector error per order of this session (tests/helpers.py::check_rel) -> gpurun_out/rel_measured.json"""
    import json

    from tests.helpers import REL, REL_MEASURED

    if not REL_MEASURED:
        return
    out = {str(p): {"worst": v[0], "where": v[1], "tolerance": REL[p]} for p, v in sorted(REL_MEASURED.items())}
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "rel_measured.json"), "w") as f:
            json.dump(out, f, indent=1)
    print("\nworst stage-vector error per order:", {k: f"{v['worst']:.2e}" for k, v in out.items()})
