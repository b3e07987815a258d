import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


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
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_gpu_objects_between_tests(request):
    """GPU tests: drop what a test leaves behind -- models, their captured HIP graphs and graph memory pools -- at a QUIET point, with the
    device idle, instead of whenever the cyclic garbage collector gets to them in the middle of a later test.  (Round 6: with the MViT
    tests in front, the train-loop tests aborted or hung inside the HIP runtime in 3 of 4 runs -- the MViT engines' ~20 captured graphs
    being destroyed while the next test's step was in flight; tools/runs/r6_flake4.sh reproduces it with PVRL_TEST_NO_GC_FIXTURE=1.)"""
    yield
    if "gpu" not in request.keywords or os.environ.get("PVRL_TEST_NO_GC_FIXTURE"):
        return
    try:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:
        pass
