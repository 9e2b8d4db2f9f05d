import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init_torch_hip_first():
    """torch bundles its own HIP runtime: when a GPU is present initialise it BEFORE libfaqcs_mi.so pulls
    in /opt/rocm's libamdhip64, otherwise torch.cuda later reports "No HIP GPUs are available"."""
    try:
        import torch

        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:
        pass


def pytest_configure(config):
    _init_torch_hip_first()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fixture_cache():
    d = os.path.join(ROOT, "tests", "golden", "_cache")
    os.makedirs(d, exist_ok=True)
    return d
