import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["small_ra_slam_problem", "single_rpm", "single_range"]
# Known-answer costs: reference tests/test_utils.cpp:210-222
EXPECTED_COST = {
    "small_ra_slam_problem": 1.063888372855624e03,
    "single_rpm": 0.809173848024762,
    "single_range": 4.718031199983851,
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=CASES)
def case(request):
    return request.param
