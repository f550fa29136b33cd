import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__)), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    """TACO_POISON=nan|big (opt-in): before every GPU test, fill a few GB of device memory with NaN / 3e38 and hand them back to
    the caching allocator, so that a kernel reading workspace it never wrote sees garbage instead of a fresh process's zeros."""
    mode = os.environ.get("TACO_POISON")
    if mode and "gpu" in request.keywords:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
            x = torch.empty(int(os.environ.get("TACO_POISON_GB", "4")) << 30, dtype=torch.uint8, device="cuda").view(torch.float32)
            x.fill_(float("nan") if mode == "nan" else 3.0e38)
            torch.cuda.synchronize()
            del x
    yield


def pytest_terminal_summary(terminalreporter):
    """How much the alignment-argmax criterion ("bit-identical in argmax", BASELINE.json) masked and excused over the whole run."""
    try:
        from util import ARGMAX_STATS as A
    except Exception:
        return
    if A["calls"]:
        terminalreporter.write_line("argmax_match over the session: %d calls, %d steps, %d masked by the floor, %d excused as fp32 ties, "
                                    "%d mismatches" % (A["calls"], A["steps"], A["masked_by_floor"], A["excused_as_ties"], A["mismatches"]))
