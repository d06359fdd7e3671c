import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


# The longest validation sweeps of plans that are no longer the default (rounds 4-5: their numbers are in DESIGN section 2 and in
# tools/x2_plan_validate.py) run only with RESR_TEST_SLOW=1: the -m gpu suite has a time box.
SLOW = os.environ.get("RESR_TEST_SLOW") == "1"
slow = pytest.mark.skipif(not SLOW, reason="validation sweep of a non-default plan: RESR_TEST_SLOW=1 runs it")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must never silently pass without a GPU
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def diag_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d
