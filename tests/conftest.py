import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    have_gpu = torch.cuda.is_available()
    have_ref = os.path.isdir("/root/reference/bnn_priors")
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))
        if "reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """the CPU sides of the parity tests (oracle samplers, eager torch-CPU forward / backward of the nets) are many small
    operators: on a 256-core GPU host torch's default of one thread per core makes them several times SLOWER"""
    import torch
    if (os.cpu_count() or 1) > 16:
        torch.set_num_threads(16)
    yield
